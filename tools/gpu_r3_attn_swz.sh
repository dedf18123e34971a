#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_episode_gpu.py -q -x -k "attn or attention or episode" > gpurun_out/r3_attn_tests.log 2>&1; echo rc=$? >> gpurun_out/r3_attn_tests.log
grep -n "passed\|failed\|rc=" gpurun_out/r3_attn_tests.log | tail -3
timeout 300 python tools/attn_probe.py 2>&1 | tail -3
bash tools/gpu_pmc_sq_r3.sh 2>&1 | grep -A1 "^epi_bwd\|^attn_bwd\|^attn_fwd" | cut -c1-210 | head -24
