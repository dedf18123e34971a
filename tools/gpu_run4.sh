#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/tests.log
tail -6 gpurun_out/tests.log
python __graft_entry__.py smoke 2>&1 | tail -3
python tools/gemm_probe.py 5152 --t3 > gpurun_out/gemm_probe_hot.log 2>&1; cat gpurun_out/gemm_probe_hot.log
python tools/gemm_probe.py 5152 --t3 --cold > gpurun_out/gemm_probe_cold.log 2>&1; cat gpurun_out/gemm_probe_cold.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
tail -2 gpurun_out/bench_prof.log
find gpurun_out/prof -name "*stats*" | head; 
