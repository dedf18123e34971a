#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
(timeout 1500 python -m pytest tests/test_auto_episode_gpu.py tests/test_full_size_configs_gpu.py "tests/test_parity_gpu.py::test_g12_episode_accumulated_gradients_vs_reference" tests/test_episode_isolation_gpu.py tests/test_episode_gpu.py tests/test_parity_r4_gpu.py -q -s -m gpu --durations=12 > $O/r6_new_tests3.log 2>&1; echo "rc=$?" >> $O/r6_new_tests3.log)
grep -n "passed\|failed\|^FAILED\|^ERROR\|\[config\|\[lazy\|rc=\|s call" $O/r6_new_tests3.log | tail -40
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6_bench_v3.json 2> $O/r6_bench_v3.err; echo "rc=$?" >> $O/r6_bench_v3.err)
tail -3 $O/r6_bench_v3.err
python - <<PY
import json
d = json.load(open("$O/r6_bench_v3.json"))
print("HEADLINE", d["value"], d["roofline"]["frac"], (d.get("whole_episodes") or {}))
u = d.get("unmodified_rollout", {})
for k in ("B8", "B1x8"):
    print("UNMOD", k, {f: (v.get("nav_steps_per_s_per_gpu"), v.get("gemm_frac_of_mfma_peak"), v.get("closed_by"), v.get("error")) for f, v in (u.get(k) or {}).items() if isinstance(v, dict)})
PY
