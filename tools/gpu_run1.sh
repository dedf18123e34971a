#!/bin/bash
# first GPU contact: kernel numerics + GEMM throughput probe
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf --tb=line -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/kernels_test.log
timeout 600 python tools/gemm_probe.py > gpurun_out/gemm_probe.log 2>&1
tail -60 gpurun_out/kernels_test.log
cat gpurun_out/gemm_probe.log
