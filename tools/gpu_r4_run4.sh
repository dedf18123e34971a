#!/bin/bash
# round 4, GPU call 4: fp8 tests under the new default mode, the WHOLE GPU suite (incl. the full-depth episode test, timed), bench
mkdir -p gpurun_out
python -m pytest tests/test_fp8_gpu.py -m gpu -q -x -s > gpurun_out/r4_run4_fp8.log 2>&1; echo rc=$? >> gpurun_out/r4_run4_fp8.log
grep "fp8 g11\|13b\|served\|passed\|failed\|rc=" gpurun_out/r4_run4_fp8.log | cut -c1-400 | tail -12
python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/r4_run4_suite.log 2>&1; echo rc=$? >> gpurun_out/r4_run4_suite.log
tail -28 gpurun_out/r4_run4_suite.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_v3.json 2> gpurun_out/r04_bench_v3.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04_bench_v3.json"))
r = d["roofline"]
print("HEADLINE", d["config"]["training_mode"], d["config"].get("teacher_forced_forward_batched"), d["value"], d["ms_per_step"], "frac", r["frac"], r["by_layout_tflops"], "gemm share", r["gemm_share_of_step"])
o = d.get("other_mode", {})
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), (o.get("roofline") or {}).get("frac"), o.get("error"))
print("WHOLE", d.get("whole_episodes"))
print("KV", json.dumps(d.get("inference_prefix_kv_reuse"))[:500])
f = d.get("fp8_weight_only_13b_config5", {})
for k, v in f.items():
    if isinstance(v, dict) and "kv_reuse_B8" in v: print(k, v)
print("T64", json.dumps(d["long_horizon_config4"].get("training_episode_T64_prefix_reuse"))[:700])
PY
