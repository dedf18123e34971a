#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/kernels_test.log
python -m pytest tests/test_parity_gpu.py -m gpu -q -rA -s --tb=short -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/parity_test.log
tail -5 gpurun_out/kernels_test.log
cat gpurun_out/parity_test.log
