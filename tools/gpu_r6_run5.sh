#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/torch_ops_in_episode.py gpurun_out/r06_torch_ops_in_episode.txt > gpurun_out/r6_torch_ops.log 2>&1; tail -45 gpurun_out/r6_torch_ops.log | cut -c1-230
bash tools/gpu_r6_final.sh v4
bash tools/gpu_pmc_sq_r4.sh r06 2>&1 | tail -30
