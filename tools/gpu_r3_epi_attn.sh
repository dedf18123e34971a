#!/bin/bash
# round 3: the one-launch episode attention backward: kernel test, the episode / G12 / mixed-task / DP tests, then whole-episode timing A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_episode_gpu.py tests/test_dp_gpu.py "tests/test_parity_gpu.py" -q -x -k "episode or g12 or dp_world2 or mixed" -s > gpurun_out/r3_epi_attn_tests.log 2>&1; echo rc=$? >> gpurun_out/r3_epi_attn_tests.log
grep -n "attn_bwd_episode\|passed\|failed\|rc=\|Error" gpurun_out/r3_epi_attn_tests.log | tail -25
for MODE in steps episode; do
echo "== NAVILLM_EPISODE_ATTN_BWD=$MODE"
NAVILLM_EPISODE_ATTN_BWD=$MODE EPISODE_REPS=4 timeout 600 python tools/episode_profile.py 2>&1 | tail -3
done
