#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_episode_gpu.py tests/test_parity_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "episode or g12 or attention or mixed" > gpurun_out/r3_deferall_tests.log 2>&1; echo rc=$? >> gpurun_out/r3_deferall_tests.log
tail -25 gpurun_out/r3_deferall_tests.log
for D in all wgrad; do
NAVILLM_EPISODE_DEFER=$D EPISODE_REPS=4 python tools/episode_profile.py 2>&1 | grep "episode\|Error\|error" | tail -5 | sed "s/^/defer=$D /"
done
