#!/bin/bash
# round 3: gemm kernel tests, the driver's bench command, an A/B of the launch planner, kernel traces of both training modes
mkdir -p gpurun_out
TAG=${1:-v1}
python -m pytest tests/test_kernels_gpu.py tests/test_dp_gpu.py -m gpu -q -k "gemm or rehearsal" > gpurun_out/r3_bench_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r3_bench_tests_$TAG.log
tail -3 gpurun_out/r3_bench_tests_$TAG.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_$TAG.json 2> gpurun_out/r03_bench_$TAG.err
tail -3 gpurun_out/r03_bench_$TAG.err
python - <<PY
import json
d = json.load(open("gpurun_out/r03_bench_$TAG.json"))
print("HEADLINE", d["config"]["training_mode"], d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["by_layout_tflops"], "gemm share", d["roofline"]["gemm_share_of_step"])
o = d.get("other_mode", {})
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), o.get("ms_per_step"), (o.get("roofline") or {}).get("frac"), (o.get("roofline") or {}).get("by_layout_tflops"), o.get("error"))
for k in ("inference_forward_only", "inference_prefix_kv_reuse"):
    print(k, (d.get(k) or {}).get("nav_steps_per_s_per_gpu"))
print("cpu", d.get("cpu_baseline"))
PY
NV_GEMM_TME=8 python bench.py --steps 12 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = d.get('other_mode', {})
print('TME=8  ', d['config']['training_mode'], d['value'], d['ms_per_step'], d['roofline']['frac'], '| other', o.get('mode'), o.get('nav_steps_per_s_per_gpu'), o.get('ms_per_step'), (o.get('roofline') or {}).get('frac'))"
python bench.py --steps 12 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = d.get('other_mode', {})
print('planned', d['config']['training_mode'], d['value'], d['ms_per_step'], d['roofline']['frac'], '| other', o.get('mode'), o.get('nav_steps_per_s_per_gpu'), o.get('ms_per_step'), (o.get('roofline') or {}).get('frac'))"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b -o b -- python bench.py --steps 12 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-profile > gpurun_out/prof_b.log 2>&1
DB=$(find gpurun_out/prof_b -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r03_bench_kernel_stats_$TAG.txt
find gpurun_out/prof_b -name "*.db" -delete
head -36 gpurun_out/r03_bench_kernel_stats_$TAG.txt
