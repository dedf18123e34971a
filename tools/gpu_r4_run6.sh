#!/bin/bash
# round 4, GPU call 6: whole GPU suite + smoke + the driver's bench command (after: fused zero_grad, episode buffer policy, fp8 default)
mkdir -p gpurun_out
TAG=${1:-v5}
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r4_gpu_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r4_gpu_tests_$TAG.log
tail -22 gpurun_out/r4_gpu_tests_$TAG.log | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_$TAG.json 2> gpurun_out/r04_bench_$TAG.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04_bench_$TAG.json"))
r = d["roofline"]
print("HEADLINE", d["config"]["training_mode"], d["config"].get("teacher_forced_forward_batched"), d["value"], d["ms_per_step"], "frac", r["frac"], r["by_layout_tflops"], "gemm share", r["gemm_share_of_step"], "traffic", r["traffic"], r.get("traffic_over_algorithmic"))
o = d.get("other_mode", {})
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), (o.get("roofline") or {}).get("frac"), (o.get("roofline") or {}).get("traffic_over_algorithmic"), o.get("error"))
print("WHOLE", d.get("whole_episodes"))
print("KV", json.dumps(d.get("inference_prefix_kv_reuse"))[:300])
f = d.get("fp8_weight_only_13b_config5", {})
for k, v in f.items():
    if isinstance(v, dict) and "kv_reuse_B8" in v: print(k, v)
print("T64", json.dumps(d["long_horizon_config4"].get("training_episode_T64_prefix_reuse"))[:500])
PY
tail -3 gpurun_out/r04_bench_$TAG.err | cut -c1-200
