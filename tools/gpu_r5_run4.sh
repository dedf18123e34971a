#!/bin/bash
# round 5, GPU call 4: episode-forward attention kernel: bit-identity test, then ABAB of the headline mode (whole 6-step episodes)
mkdir -p gpurun_out
TAG=${1:-v4}
timeout 900 python -m pytest tests/test_episode_gpu.py -m gpu -q -x -k "one_launch_equals or teacher_forced" > gpurun_out/r5_epifwd_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r5_epifwd_tests_$TAG.log
tail -8 gpurun_out/r5_epifwd_tests_$TAG.log | cut -c1-300
ARGS="--steps 18 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode"
for rep in 1 2; do
  for form in steps episode; do
    NAVILLM_EPISODE_ATTN_FWD=$form python bench.py $ARGS > gpurun_out/ab_$form$rep.json 2> gpurun_out/ab_$form$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_$form$rep.json"))
print("$form", $rep, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["gemm_share_of_step"])
PY
  done
done
