#!/bin/bash
# round 5: per-step forward through the episode-forward kernel -- bit-identity test, the episode tests, A/B of the per-step-forward rate
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_episode_gpu.py -x -q -m gpu > gpurun_out/stepfwd_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/stepfwd.txt
tail -4 gpurun_out/stepfwd_tests.log >> gpurun_out/stepfwd.txt
for rep in 1 2; do
for f in steps episode; do
NAVILLM_EPISODE_ATTN_FWD=$f timeout 300 python bench.py --steps 18 --warmup 6 --prewarm 6 --no-tf-batch --no-extras --no-cpu-baseline --infer-steps 0 --no-profile --no-other-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', d['value'], (d.get('whole_episodes') or {}).get('nav_steps_per_s'), (d.get('whole_episodes') or {}).get('per_step_forward_nav_steps_per_s'))" >> gpurun_out/stepfwd.txt 2>&1
done; done
cat gpurun_out/stepfwd.txt
