#!/bin/bash
# round 5, GPU call 3: the 4-wave (128x128 per wave) full tile: bit-identity tests, then the 48-cell table with it as an extra column
mkdir -p gpurun_out
TAG=${1:-v3}
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "four_wave or (gemm_bf16_layouts and 10)" > gpurun_out/r5_w4_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r5_w4_tests_$TAG.log
tail -6 gpurun_out/r5_w4_tests_$TAG.log | cut -c1-300
timeout 900 python tools/gemm_vs_blaslt.py gpurun_out/r05_gemm_vs_blaslt_$TAG.txt > gpurun_out/r5_gemm_vs_blaslt_$TAG.log 2>&1; tail -52 gpurun_out/r5_gemm_vs_blaslt_$TAG.log | cut -c1-200
