#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_episode_gpu.py -m gpu -q -x -k "first_writer or teacher or one_launch" > gpurun_out/r5_tests_v17.log 2>&1; echo rc=$? >> gpurun_out/r5_tests_v17.log
tail -5 gpurun_out/r5_tests_v17.log | cut -c1-300
ARGS="--steps 18 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode"
for rep in 1 2; do
  for f in 0 1; do
    NAVILLM_WGRAD_STORE=$f python bench.py $ARGS > gpurun_out/abs_$f$rep.json 2> gpurun_out/abs_$f$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/abs_$f$rep.json"))
r = d["roofline"]
print("wgrad_store=$f", $rep, d["value"], d["ms_per_step"], r["frac"], r["by_layout_tflops"], r["gemm_share_of_step"])
PY
  done
done
