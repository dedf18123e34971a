#!/bin/bash
# round 6: whole GPU suite + smoke, the driver's bench command, kernel trace of the headline mode (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out
TAG=${1:-v1}
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r6_gpu_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r6_gpu_tests_$TAG.log
tail -18 gpurun_out/r6_gpu_tests_$TAG.log | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_$TAG.json 2> gpurun_out/r06_bench_$TAG.err; echo bench rc=$?
python - <<PY
import json
d = json.load(open("gpurun_out/r06_bench_$TAG.json"))
r = d["roofline"]
print("HEADLINE", d["config"]["training_mode"], d["value"], d["ms_per_step"], "frac", r["frac"], r["by_layout_tflops"], "gemm share", r["gemm_share_of_step"], "traffic", r["traffic"], r.get("traffic_over_algorithmic"))
o = d.get("other_mode", {})
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), (o.get("roofline") or {}).get("frac"), (o.get("roofline") or {}).get("traffic"), o.get("error"))
print("WHOLE", {k: v for k, v in (d.get("whole_episodes") or {}).items() if "what" not in k})
print("REFLINE", {k: (v.get("nav_steps_per_s_per_gpu"), (v.get("roofline") or {}).get("frac")) for k, v in d.get("reference_launch_line", {}).items() if isinstance(v, dict)})
print("KV", json.dumps(d.get("inference_prefix_kv_reuse"))[:300])
f = d.get("fp8_weight_only_13b_config5", {})
print("FP8", json.dumps(f)[:200] if "error" in f else {k: v for k, v in f.items() if isinstance(v, dict) and "kv_reuse_B8" in v})
print("T64", json.dumps(d["long_horizon_config4"].get("training_episode_T64_prefix_reuse"))[:400])
c3 = d.get("mixed_task_training_config3", {}); print("C3", c3.get("nav_steps_per_s_per_gpu"), (c3.get("navigation_over_cached_prefix") or {}).get("nav_steps_per_s_per_gpu"), c3.get("error"))
print("CPU", d.get("cpu_baseline"))
u = d.get("unmodified_rollout", {})
for k in ("B8", "B1x8"):
    print("UNMOD", k, {f: (v.get("nav_steps_per_s_per_gpu"), v.get("gemm_frac_of_mfma_peak"), v.get("closed_by"), v.get("error")) for f, v in (u.get(k) or {}).items() if isinstance(v, dict)}, (u.get(k) or {}).get("automatic_over_explicit"))
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b -o b -- python bench.py --mode prefix_reuse --steps 12 --warmup 6 --prewarm 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-profile --no-other-mode > gpurun_out/prof_b.log 2>&1
DB=$(find gpurun_out/prof_b -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r06_bench_kernel_stats_prefix_reuse_$TAG.txt
find gpurun_out/prof_b -name "*.db" -delete
head -16 gpurun_out/r06_bench_kernel_stats_prefix_reuse_$TAG.txt | cut -c1-160
[ -n "$SKIP_PMC" ] || bash tools/gpu_pmc_bench_r4.sh r06 2>&1 | tail -8
