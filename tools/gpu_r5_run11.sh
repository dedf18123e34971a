#!/bin/bash
mkdir -p gpurun_out
for lz in 1; do echo "NAVILLM_EPISODE_LAZY_PREFIX=$lz"; NAVILLM_EPISODE_LAZY_PREFIX=$lz python tools/episode_host_probe.py 8 2>&1 | tail -9 | cut -c1-330; done
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
