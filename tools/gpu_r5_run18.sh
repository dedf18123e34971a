#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_episode_gpu.py tests/test_parity_gpu.py tests/test_round2_gpu.py tests/test_dp_gpu.py -m gpu -q -x -k "first_writer or g3_g4 or g10 or g12 or invariants or world2_shared" > gpurun_out/r5_tests_v18.log 2>&1; echo rc=$? >> gpurun_out/r5_tests_v18.log
tail -5 gpurun_out/r5_tests_v18.log | cut -c1-300
python bench.py --mode recompute --steps 12 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode > gpurun_out/rc1.json 2>/dev/null
NAVILLM_WGRAD_STORE=0 python bench.py --mode recompute --steps 12 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-other-mode > gpurun_out/rc0.json 2>/dev/null
python - <<PY
import json
for f in ("rc1", "rc0"):
    d = json.load(open(f"gpurun_out/{f}.json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r["by_layout_tflops"])
PY
