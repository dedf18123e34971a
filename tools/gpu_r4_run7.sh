#!/bin/bash
# round 4, GPU call 7: the tests touched since the last full pass, bench, SQ counters of the headline episode
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_episode_gpu.py tests/test_dp_gpu.py -m gpu -q -x -k "adamw or episode or prefix" > gpurun_out/r4_run7_tests.log 2>&1; echo rc=$? >> gpurun_out/r4_run7_tests.log
tail -6 gpurun_out/r4_run7_tests.log | cut -c1-250
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_v6.json 2> gpurun_out/r04_bench_v6.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04_bench_v6.json"))
r = d["roofline"]
print("HEADLINE", d["config"]["training_mode"], d["value"], d["ms_per_step"], "frac", r["frac"], r["by_layout_tflops"], "gemm share", r["gemm_share_of_step"], "traffic", r["traffic"], r.get("traffic_over_algorithmic"))
o = d.get("other_mode", {})
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), (o.get("roofline") or {}).get("frac"), o.get("error"))
print("WHOLE", d.get("whole_episodes"))
f = d.get("fp8_weight_only_13b_config5", {})
print("FP8", json.dumps(f)[:200] if "error" in f else {k: v for k, v in f.items() if isinstance(v, dict) and "kv_reuse_B8" in v})
print("T64", json.dumps(d["long_horizon_config4"].get("training_episode_T64_prefix_reuse"))[:500])
c3 = d.get("mixed_task_training_config3", {}); print("C3", c3.get("nav_steps_per_s_per_gpu"), (c3.get("navigation_over_cached_prefix") or {}).get("nav_steps_per_s_per_gpu"), c3.get("error"))
PY
bash tools/gpu_pmc_sq_r4.sh 2>&1 | tail -40 | cut -c1-200
