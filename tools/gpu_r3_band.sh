#!/bin/bash
# round 3: band-distributed split-K reduction -- tests, probes (A/B against the last-arriver form and other slice limits), bench
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py -m gpu -q -k "gemm or invariants" > gpurun_out/r3_band_tests.log 2>&1; echo rc=$? >> gpurun_out/r3_band_tests.log
tail -3 gpurun_out/r3_band_tests.log
timeout 600 python tools/gemm_tme_probe.py > gpurun_out/r03_gemm_tme_probe_v2.txt 2>&1
echo "== band reduce (default), quick"; timeout 300 python tools/gemm_tme_probe.py quick 2>&1 | grep "M=" | tee gpurun_out/r3_band_quick_on.txt
echo "== NV_GEMM_BAND_REDUCE=0"; NV_GEMM_BAND_REDUCE=0 timeout 300 python tools/gemm_tme_probe.py quick 2>&1 | grep "M=" | tee gpurun_out/r3_band_quick_off.txt
echo "== MINSLICE=12 MAXSPLIT=8"; NV_GEMM_MINSLICE=12 NV_GEMM_MAXSPLIT=8 timeout 300 python tools/gemm_tme_probe.py quick 2>&1 | grep "M=" | tee gpurun_out/r3_band_quick_ms12.txt
echo "== MINSLICE=8 MAXSPLIT=8"; NV_GEMM_MINSLICE=8 NV_GEMM_MAXSPLIT=8 timeout 300 python tools/gemm_tme_probe.py quick 2>&1 | grep "M=" | tee gpurun_out/r3_band_quick_ms8.txt
for B in 1 0; do
NV_GEMM_BAND_REDUCE=$B python bench.py --steps 12 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = d.get('other_mode', {})
print('band=$B', d['config']['training_mode'], d['value'], d['ms_per_step'], d['roofline']['frac'], '| other', o.get('mode'), o.get('nav_steps_per_s_per_gpu'), o.get('ms_per_step'), (o.get('roofline') or {}).get('frac'))"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b -o b -- python bench.py --steps 12 --warmup 6 --prewarm 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-profile --no-other-mode > gpurun_out/prof_b.log 2>&1
DB=$(find gpurun_out/prof_b -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r03_bench_kernel_stats_prefix_v2.txt
find gpurun_out/prof_b -name "*.db" -delete
head -60 gpurun_out/r03_bench_kernel_stats_prefix_v2.txt
