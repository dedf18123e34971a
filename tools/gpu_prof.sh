#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 0 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
tail -1 gpurun_out/bench_prof.log | cut -c1-300
