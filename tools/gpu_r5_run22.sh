#!/bin/bash
# round 5: the reading GEMM epilogues (accumulate / residual / RoPE) issue their global loads eight passes ahead: kernel tests, then ABAB
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_episode_gpu.py tests/test_parity_gpu.py tests/test_fp8_gpu.py -m gpu -q -x -k "gemm or rope or first_writer or one_launch or g3_g4 or g12 or teacher or tile_gemm" > gpurun_out/r5_tests_v22.log 2>&1; echo rc=$? >> gpurun_out/r5_tests_v22.log
tail -5 gpurun_out/r5_tests_v22.log | cut -c1-300
ARGS="--steps 18 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode"
for rep in 1 2; do
  for f in 0 1; do
    NV_GEMM_EPI_PRELOAD=$f python bench.py $ARGS > gpurun_out/abp_$f$rep.json 2> gpurun_out/abp_$f$rep.err
    NV_GEMM_EPI_PRELOAD=$f python bench.py --mode recompute --steps 12 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode > gpurun_out/abpr_$f$rep.json 2> gpurun_out/abpr_$f$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/abp_$f$rep.json")); r = d["roofline"]
e = json.load(open("gpurun_out/abpr_$f$rep.json")); q = e["roofline"]
print("epi_preload=$f", $rep, "headline", d["value"], d["ms_per_step"], r["frac"], r["by_layout_tflops"], "| recompute", e["value"], q["frac"], q["by_layout_tflops"])
PY
  done
done
