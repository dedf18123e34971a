#!/bin/bash
# round 3: full GPU suite + smoke, the driver's bench command, per-mode kernel traces, PMC traffic of the GEMM launches
mkdir -p gpurun_out
TAG=${1:-v3}
python -m pytest tests -m gpu -q > gpurun_out/r3_gpu_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r3_gpu_tests_$TAG.log
tail -4 gpurun_out/r3_gpu_tests_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_$TAG.json 2> gpurun_out/r03_bench_$TAG.err
python - <<PY
import json
d = json.load(open("gpurun_out/r03_bench_$TAG.json"))
print("HEADLINE", d["config"]["training_mode"], d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["by_layout_tflops"], "gemm share", d["roofline"]["gemm_share_of_step"])
o = d.get("other_mode", {})
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), o.get("ms_per_step"), (o.get("roofline") or {}).get("frac"), (o.get("roofline") or {}).get("by_layout_tflops"), o.get("error"))
print(d["config"].get("timed_window"))
print("WHOLE", d.get("whole_episodes"))
for k in ("inference_forward_only", "inference_prefix_kv_reuse"):
    print(k, (d.get(k) or {}).get("nav_steps_per_s_per_gpu"))
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for MODE in prefix_reuse recompute; do
rm -rf gpurun_out/prof_b
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b -o b -- python bench.py --mode $MODE --steps 12 --warmup 6 --prewarm 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-profile --no-other-mode > gpurun_out/prof_b.log 2>&1
DB=$(find gpurun_out/prof_b -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r03_bench_kernel_stats_${MODE}_$TAG.txt
find gpurun_out/prof_b -name "*.db" -delete
head -5 gpurun_out/r03_bench_kernel_stats_${MODE}_$TAG.txt
done
bash tools/gpu_pmc_bench_r3.sh 2>&1 | tail -3
