#!/bin/bash
# ABAB: lazy prefix always (1) vs auto in the driver's window and over whole episodes
mkdir -p gpurun_out
ARGS="--steps 20 --warmup 5 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode"
for rep in 1 2 3; do
  for lz in 1 auto; do
    NAVILLM_EPISODE_LAZY_PREFIX=$lz python bench.py $ARGS > gpurun_out/abl_$lz$rep.json 2> gpurun_out/abl_$lz$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/abl_$lz$rep.json"))
r = d["roofline"]; w = d.get("whole_episodes") or {}
print("lazy_prefix=$lz", $rep, "K=20:", d["value"], d["ms_per_step"], "frac", r["frac"], r["gemm_share_of_step"], "| whole:", w.get("nav_steps_per_s"))
PY
  done
done
