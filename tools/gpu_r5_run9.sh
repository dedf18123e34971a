#!/bin/bash
# round 5, GPU call 9: ABAB of the headline mode, lazy prefix off / on, after removing the per-episode free-memory query
mkdir -p gpurun_out
ARGS="--steps 18 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode"
for rep in 1 2 3; do
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
