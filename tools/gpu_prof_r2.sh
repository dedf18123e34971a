#!/bin/bash
# round 2: rocprofv3 kernel stats of the headline bench command (training steps only) -> gpurun_out/prof_r2/, summary text
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r2
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2 -o bench -- python bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-extras --infer-steps 0 > gpurun_out/bench_prof_r2.log 2>&1
DB=$(find gpurun_out/prof_r2 -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r02_bench_kernel_stats.txt
head -12 gpurun_out/r02_bench_kernel_stats.txt
find gpurun_out/prof_r2 -name "*.db" -delete
