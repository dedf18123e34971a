#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
(timeout 1500 python -m pytest tests/test_episode_gpu.py tests/test_auto_episode_gpu.py tests/test_episode_isolation_gpu.py tests/test_dp_gpu.py tests/test_parity_gpu.py tests/test_kvcache_gpu.py tests/test_round2_gpu.py -q -m gpu > $O/r6_tests_nocache.log 2>&1; echo "rc=$?" >> $O/r6_tests_nocache.log); tail -4 $O/r6_tests_nocache.log
(NAVILLM_POISON=1 timeout 900 python -m pytest tests/test_episode_gpu.py tests/test_auto_episode_gpu.py -q -m gpu > $O/r6_tests_nocache_poison.log 2>&1; echo "rc=$?" >> $O/r6_tests_nocache_poison.log); tail -4 $O/r6_tests_nocache_poison.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --infer-steps 0 > $O/r06_bench_v8_quick.json 2> $O/r06_bench_v8_quick.err
python - <<PY
import json
d = json.load(open("$O/r06_bench_v8_quick.json"))
print("QUICK", d["value"], d["roofline"]["frac"], d.get("whole_episodes"))
PY
