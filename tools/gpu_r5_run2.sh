#!/bin/bash
# round 5, GPU call 2: the two tests whose bounds were re-stated, hipBLASLt's kernel choice for NT at M = 7700 (where it wins), PMC traffic of both modes for the new kernel hash
mkdir -p gpurun_out
TAG=${1:-v2}
python -m pytest tests/test_train_mode_gpu.py tests/test_parity_r5_gpu.py -m gpu -q -s -k "g14_training_mode_step or per_layer" > gpurun_out/r5_new_tests_$TAG.log 2>&1; echo rc=$? >> gpurun_out/r5_new_tests_$TAG.log
grep "^\[g14\]\|^\[per-layer\|passed\|failed\|rc=\|Error\|assert" gpurun_out/r5_new_tests_$TAG.log | cut -c1-500 | tail -30
bash tools/gpu_blaslt_names.sh 7700 2>&1 | cut -c1-400 | tail -20
bash tools/gpu_blaslt_names.sh 4272 2>&1 | cut -c1-400 | tail -20
bash tools/gpu_pmc_bench_r4.sh r05 2>&1 | tail -12
