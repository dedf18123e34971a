#!/bin/bash
# round 6, closing run on the committed tree: suite, smoke, bench, kernel trace (PMC skipped: kernels unchanged since tools/gpu_r6_run7.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SKIP_PMC=1 bash tools/gpu_r6_final.sh v10
