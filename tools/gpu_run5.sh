#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/tests.log
tail -12 gpurun_out/tests.log
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -25 gpurun_out/smoke.log
python bench.py --steps 12 --warmup 6 2>&1 | tail -3 | tee gpurun_out/bench2.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
tail -1 gpurun_out/bench_prof.log
