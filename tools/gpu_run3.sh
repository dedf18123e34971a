#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/tests.log
tail -6 gpurun_out/tests.log
python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python bench.py --steps 6 --warmup 1 2>&1 | tail -20 | tee gpurun_out/bench_first.log
