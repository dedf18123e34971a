#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
(timeout 1500 python -m pytest tests/test_dp_gpu.py tests/test_auto_episode_gpu.py tests/test_episode_gpu.py -q -m gpu > $O/r6_tests_dp2.log 2>&1; echo "rc=$?" >> $O/r6_tests_dp2.log); tail -4 $O/r6_tests_dp2.log
NAVILLM_DP_FORCE=1 timeout 600 python bench.py --gpus 1 --steps 12 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode > $O/r6_bench_dpforce.json 2> $O/r6_bench_dpforce.err; echo "dpforce rc=$?"; tail -2 $O/r6_bench_dpforce.err
python - <<PY
import json
d = json.load(open("$O/r6_bench_dpforce.json"))
print("DPFORCE", d["value"], d["roofline"]["frac"], d.get("whole_episodes", {}).get("nav_steps_per_s"))
PY
