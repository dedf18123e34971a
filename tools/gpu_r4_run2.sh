#!/bin/bash
# round 4, GPU call 2: fp8 tile GEMM tests + probe, episode segment / truncation tests, 8-layer gradient parity, the whole GPU suite,
# the driver's bench command
mkdir -p gpurun_out
python -m pytest tests/test_fp8_gpu.py -m gpu -q -x -s -k "tile_gemm" > gpurun_out/r4_run2_fp8.log 2>&1; echo rc=$? >> gpurun_out/r4_run2_fp8.log
grep "gemm_fp8w\|passed\|failed\|rc=" gpurun_out/r4_run2_fp8.log | cut -c1-330 | tail -12
timeout 600 python tools/gemm_fp8_probe.py > gpurun_out/r4_gemm_fp8_probe.txt 2>&1; tail -8 gpurun_out/r4_gemm_fp8_probe.txt | cut -c1-300
python -m pytest tests/test_episode_gpu.py -m gpu -q -x -s -k "long_episode or truncated" > gpurun_out/r4_run2_episode.log 2>&1; echo rc=$? >> gpurun_out/r4_run2_episode.log
grep "long episode\|truncation\|passed\|failed\|rc=\|Error" gpurun_out/r4_run2_episode.log | cut -c1-400 | tail -8
python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -x -s -k "eight_layer" > gpurun_out/r4_run2_8layer.log 2>&1; echo rc=$? >> gpurun_out/r4_run2_8layer.log
grep "^\[8-layer\|passed\|failed\|rc=" gpurun_out/r4_run2_8layer.log | cut -c1-300 | tail -16
python -m pytest tests -m gpu -q -x --deselect tests/test_parity_r4_gpu.py::test_full_depth_7b_b8_episode_prefix_reuse_vs_recompute_vs_oracle --deselect tests/test_parity_r4_gpu.py::test_eight_layer_7b_width_episode_gradients_vs_oracle_autograd > gpurun_out/r4_run2_suite.log 2>&1; echo rc=$? >> gpurun_out/r4_run2_suite.log
tail -6 gpurun_out/r4_run2_suite.log | cut -c1-300
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_v1.json 2> gpurun_out/r04_bench_v1.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04_bench_v1.json"))
r = d["roofline"]
print("HEADLINE", d["config"]["training_mode"], d["value"], d["ms_per_step"], "frac", r["frac"], r["by_layout_tflops"], "gemm share", r["gemm_share_of_step"], "traffic", r["traffic"], r.get("algorithmic_bytes_per_launch"), r.get("traffic_over_algorithmic"))
o = d.get("other_mode", {})
ro = o.get("roofline") or {}
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), o.get("ms_per_step"), ro.get("frac"), ro.get("traffic"), ro.get("algorithmic_bytes_per_launch"), ro.get("traffic_over_algorithmic"), o.get("error"))
print("WHOLE", d.get("whole_episodes"))
for k in ("inference_forward_only", "inference_prefix_kv_reuse", "fp8_weight_only_13b_config5", "long_horizon_config4"):
    print(k, json.dumps(d.get(k))[:900])
PY
