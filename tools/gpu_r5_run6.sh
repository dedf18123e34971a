#!/bin/bash
# round 5, GPU call 6: B = 1 A/B of the episode-forward attention
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
# (the NV_ADAMW_MODE loop that followed -- AdamW with non-temporal loads / stores and two 8-element groups per thread in flight --
# measured 21.4 / 21.4 / 20.9 ms and 21.2 / 21.3 / 21.0 ms per clip + step for modes 0 / 1 / 2: noise; the variant kernels were reverted)
