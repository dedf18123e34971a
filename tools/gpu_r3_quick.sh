#!/bin/bash
# quick check: episode / G12 / mixed-task / DP / fp8 tests + whole-episode timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_episode_gpu.py tests/test_dp_gpu.py tests/test_parity_gpu.py tests/test_fp8_gpu.py -q -x -k "episode or g12 or dp_world2 or mixed or fp8" > gpurun_out/r3_quick_tests.log 2>&1; echo rc=$? >> gpurun_out/r3_quick_tests.log
tail -4 gpurun_out/r3_quick_tests.log
EPISODE_REPS=4 timeout 600 python tools/episode_profile.py 2>&1 | tail -3
