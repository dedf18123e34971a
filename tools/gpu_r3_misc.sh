#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_episode_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "mixed_task or attention or g8" 2>&1 | tail -3
for V in 1 2; do
NV_ATTN_BWD_VARIANT=$V EPISODE_REPS=3 python tools/episode_profile.py 2>&1 | grep episode | sed "s/^/attn_bwd_variant=$V /"
done
python bench.py --steps 12 --warmup 6 --no-cpu-baseline --infer-steps 0 2>/dev/null > gpurun_out/r3_misc_bench.json
python - <<PY
import json
d = json.load(open("gpurun_out/r3_misc_bench.json"))
o = d.get("other_mode", {})
print("bench", d["config"]["training_mode"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["gemm_share_of_step"], "| other", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), o.get("ms_per_step"), (o.get("roofline") or {}).get("frac"), (o.get("roofline") or {}).get("gemm_share_of_step"))
print(json.dumps(d.get("mixed_task_training_config3"))[:1500])
PY
