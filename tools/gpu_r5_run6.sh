#!/bin/bash
# round 5, GPU call 6: B = 1 A/B of the episode-forward attention; AdamW non-temporal / unrolled variants
mkdir -p gpurun_out
ARGS="--batch 1 --steps 18 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode"
for rep in 1 2; do
  for form in steps episode; do
    NAVILLM_EPISODE_ATTN_FWD=$form python bench.py $ARGS > gpurun_out/ab1_$form$rep.json 2> gpurun_out/ab1_$form$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/ab1_$form$rep.json"))
print("B=1 $form", $rep, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
  done
done
for mode in 0 1 2 0 1 2; do
  echo "NV_ADAMW_MODE=$mode"; NV_ADAMW_MODE=$mode python tools/adamw_time.py 2>&1 | grep "step\|zero"
done
