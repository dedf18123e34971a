#!/bin/bash
# round 6: this round's copy of the library comparison (48 cells) and the attention probe -- unchanged kernels, fresh numbers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/gemm_vs_blaslt.py gpurun_out/r06_gemm_vs_blaslt.txt > gpurun_out/r6_blaslt.log 2>&1; tail -20 gpurun_out/r6_blaslt.log | cut -c1-200
timeout 300 python tools/attn_probe.py > gpurun_out/r06_attn_probe.txt 2>&1; cat gpurun_out/r06_attn_probe.txt | tail -8
