#!/bin/bash
# round 5, GPU call 1: the new parity tests (train-mode G14, per-layer re-synchronised 7B, guards, B x accum equivalence) + the
# regression tests of what this round touched, smoke, the driver's bench command, this GEMM vs hipBLASLt on the headline shapes
mkdir -p gpurun_out
TAG=${1:-v1}
python -m pytest tests/test_train_mode_gpu.py tests/test_parity_r5_gpu.py -m gpu -q -s --durations=8 > gpurun_out/r5_new_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r5_new_tests_$TAG.log
grep "^\[g14\]\|^\[per-layer\|^\[B=\|passed\|failed\|rc=\|Error\|assert" gpurun_out/r5_new_tests_$TAG.log | cut -c1-400 | tail -40
python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py tests/test_parity_gpu.py tests/test_episode_gpu.py -m gpu -q -x -k "mha or encoder or g1_ or g3_ or g12 or default_mode or tile_gemm or truncated or segments or kvcache_fp8 or fp8" > gpurun_out/r5_regr_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r5_regr_tests_$TAG.log
tail -5 gpurun_out/r5_regr_tests_$TAG.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_$TAG.json 2> gpurun_out/r05_bench_$TAG.err; echo bench rc=$?
python - <<PY
import json
d = json.load(open("gpurun_out/r05_bench_$TAG.json"))
r = d["roofline"]
print("HEADLINE", d["config"]["training_mode"], d["value"], d["ms_per_step"], "frac", r["frac"], r["by_layout_tflops"], "gemm share", r["gemm_share_of_step"])
o = d.get("other_mode", {})
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), (o.get("roofline") or {}).get("frac"), o.get("error"))
print("WHOLE", d.get("whole_episodes"))
print("BLEND", d.get("finetune_blend"))
print("REFLINE", json.dumps(d.get("reference_launch_line"))[:1200])
print("KV", json.dumps(d.get("inference_prefix_kv_reuse"))[:300])
PY
timeout 600 python tools/gemm_vs_blaslt.py gpurun_out/r05_gemm_vs_blaslt_$TAG.txt > gpurun_out/r5_gemm_vs_blaslt_$TAG.log 2>&1; tail -52 gpurun_out/r5_gemm_vs_blaslt_$TAG.log | cut -c1-200
