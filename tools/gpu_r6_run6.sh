#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_auto_episode_gpu.py -q -m gpu > $O/r6_dp_tests.log 2>&1; echo "rc=$?" >> $O/r6_dp_tests.log); tail -5 $O/r6_dp_tests.log
for tag in v5 v6; do
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_$tag.json 2> $O/r06_bench_$tag.err; echo "rc=$?" >> $O/r06_bench_$tag.err)
tail -2 $O/r06_bench_$tag.err
python - <<PY
import json
d = json.load(open("$O/r06_bench_$tag.json"))
print("HEADLINE", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("episodes", "")[:40], (d.get("whole_episodes") or {}).get("nav_steps_per_s"), (d.get("whole_episodes") or {}).get("per_step_forward_nav_steps_per_s"), "other", (d.get("other_mode") or {}).get("nav_steps_per_s_per_gpu"))
u = d.get("unmodified_rollout", {})
for k in ("B8", "B1x8"):
    print("UNMOD", k, {f: (v.get("nav_steps_per_s_per_gpu"), v.get("error")) for f, v in (u.get(k) or {}).items() if isinstance(v, dict)})
c3 = d.get("mixed_task_training_config3", {}); print("C3", c3.get("nav_steps_per_s_per_gpu"), (c3.get("navigation_over_cached_prefix") or {}).get("nav_steps_per_s_per_gpu"), (c3.get("unmodified_rollout_automatic_episodes") or {}))
PY
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --explicit-episodes --no-extras --no-cpu-baseline --infer-steps 0 > $O/r06_bench_explicit.json 2> $O/r06_bench_explicit.err
python - <<PY
import json
d = json.load(open("$O/r06_bench_explicit.json"))
print("EXPLICIT", d["value"], d["ms_per_step"], d["roofline"]["frac"], (d.get("whole_episodes") or {}).get("nav_steps_per_s"))
PY
