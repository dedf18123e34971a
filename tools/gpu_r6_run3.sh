#!/bin/bash
# round 6, call 4: the new tests (lazy automatic episodes, full-size configs, wrapped clip), ABAB of the overlapped optimizer update, the bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
(timeout 1500 python -m pytest tests/test_auto_episode_gpu.py tests/test_full_size_configs_gpu.py "tests/test_parity_gpu.py::test_g12_episode_accumulated_gradients_vs_reference" tests/test_episode_isolation_gpu.py tests/test_episode_gpu.py -q -s -m gpu --durations=12 > $O/r6_new_tests2.log 2>&1; echo "rc=$?" >> $O/r6_new_tests2.log)
grep -n "passed\|failed\|^FAILED\|^ERROR\|\[config\|\[lazy\|rc=" $O/r6_new_tests2.log | tail -30
for rep in 1 2; do for ov in 1 0; do
  NAVILLM_ADAMW_OVERLAP=$ov timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --infer-steps 0 > $O/r6_ab_overlap${ov}_$rep.json 2> $O/r6_ab_overlap${ov}_$rep.err
  python - <<PY
import json
d = json.load(open("$O/r6_ab_overlap${ov}_$rep.json"))
print("OVERLAP=$ov rep $rep", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "whole", (d.get("whole_episodes") or {}).get("nav_steps_per_s"), "psf", (d.get("whole_episodes") or {}).get("per_step_forward_nav_steps_per_s"))
PY
done; done
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6_bench_v2.json 2> $O/r6_bench_v2.err; echo "rc=$?" >> $O/r6_bench_v2.err)
tail -3 $O/r6_bench_v2.err
python - <<PY
import json
d = json.load(open("$O/r6_bench_v2.json"))
print("HEADLINE", d["value"], d["roofline"]["frac"], (d.get("whole_episodes") or {}).get("nav_steps_per_s"))
u = d.get("unmodified_rollout", {})
for k in ("B8", "B1x8"):
    print("UNMOD", k, {f: (v.get("nav_steps_per_s_per_gpu"), v.get("gemm_frac_of_mfma_peak"), v.get("closed_by"), v.get("error")) for f, v in (u.get(k) or {}).items() if isinstance(v, dict)})
PY
