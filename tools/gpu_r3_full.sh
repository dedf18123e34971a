#!/bin/bash
# round 3: the whole GPU test suite, the fill diagnostic, the driver's bench command
mkdir -p gpurun_out
TAG=${1:-v2}
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3_gpu_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r3_gpu_tests_$TAG.log
tail -14 gpurun_out/r3_gpu_tests_$TAG.log
python tools/find_fills.py 2>&1 | tail -25 | tee gpurun_out/r3_fills.txt
for F in 1 0; do
NAVILLM_EPISODE_FUSE_KVACC=$F EPISODE_REPS=3 python tools/episode_profile.py 2>&1 | grep episode | sed "s/^/fuse_kvacc=$F /"
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_$TAG.json 2> gpurun_out/r03_bench_$TAG.err
python - <<PY
import json
d = json.load(open("gpurun_out/r03_bench_$TAG.json"))
print("HEADLINE", d["config"]["training_mode"], d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["by_layout_tflops"], "gemm share", d["roofline"]["gemm_share_of_step"])
o = d.get("other_mode", {})
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), o.get("ms_per_step"), (o.get("roofline") or {}).get("frac"), (o.get("roofline") or {}).get("by_layout_tflops"), o.get("error"))
for k in ("inference_forward_only", "inference_prefix_kv_reuse"):
    print(k, (d.get(k) or {}).get("nav_steps_per_s_per_gpu"))
for k in ("mixed_task_training_config3", "long_horizon_config4", "fp8_weight_only_13b_config5"):
    print(k, json.dumps(d.get(k))[:600])
PY
