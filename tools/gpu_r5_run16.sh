#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_train_mode_gpu.py tests/test_episode_gpu.py tests/test_parity_gpu.py -m gpu -q -x -s -k "real_size or prefix_episode_matches or g12 or mixed or long_episode or teacher" > gpurun_out/r5_tests_v16.log 2>&1; echo rc=$? >> gpurun_out/r5_tests_v16.log
grep "train-mode encoder\|passed\|failed\|rc=\|Error" gpurun_out/r5_tests_v16.log | cut -c1-300 | tail
