#!/bin/bash
# round 5, GPU call 7: the prefix joins the teacher-forced batch (lazy prefix): episode / G12 / DP tests, then ABAB of the headline mode
mkdir -p gpurun_out
TAG=${1:-v7}
timeout 1200 python -m pytest tests/test_episode_gpu.py tests/test_parity_gpu.py tests/test_dp_gpu.py tests/test_parity_r5_gpu.py -m gpu -q -x -k "teacher_forced or one_launch or g12 or prefix_reuse or segments or truncated or abort" > gpurun_out/r5_lazy_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r5_lazy_tests_$TAG.log
tail -8 gpurun_out/r5_lazy_tests_$TAG.log | cut -c1-300
ARGS="--steps 18 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode"
for rep in 1 2; do
  for lz in 0 1; do
    NAVILLM_EPISODE_LAZY_PREFIX=$lz python bench.py $ARGS > gpurun_out/abl_$lz$rep.json 2> gpurun_out/abl_$lz$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/abl_$lz$rep.json"))
r = d["roofline"]
print("lazy_prefix=$lz", $rep, d["value"], d["ms_per_step"], r["frac"], r["by_layout_tflops"], r["gemm_share_of_step"], r["launches"])
PY
  done
done
