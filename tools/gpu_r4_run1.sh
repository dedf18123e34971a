#!/bin/bash
# round 4, GPU call 1: the new parity / guard / DP tests, PMC traffic of both training modes, then the full-depth episode test
mkdir -p gpurun_out
python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -x -s -k "guards or rope_frame" > gpurun_out/r4_run1_quick.log 2>&1; echo rc=$? >> gpurun_out/r4_run1_quick.log
tail -5 gpurun_out/r4_run1_quick.log
python -m pytest tests/test_dp_gpu.py -m gpu -q -x -s -k "prefix_reuse" > gpurun_out/r4_run1_dp.log 2>&1; echo rc=$? >> gpurun_out/r4_run1_dp.log
tail -3 gpurun_out/r4_run1_dp.log
bash tools/gpu_pmc_bench_r4.sh r04 2>&1 | tail -6
python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -x -s -k "eight_layer" > gpurun_out/r4_run1_8layer.log 2>&1; echo rc=$? >> gpurun_out/r4_run1_8layer.log
tail -5 gpurun_out/r4_run1_8layer.log
python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -x -s -k "full_depth" > gpurun_out/r4_run1_full.log 2>&1; echo rc=$? >> gpurun_out/r4_run1_full.log
tail -5 gpurun_out/r4_run1_full.log
