#!/bin/bash
# round 6, the last call: final tree -- the suite under poison in reversed order, then tools/gpu_r6_final.sh (suite, smoke, bench, kernel trace; PMC skipped: unchanged kernels)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
(NAVILLM_POISON=1 timeout 1800 python -m pytest tests -q -m gpu --nv-order reverse > gpurun_out/r6_suite_poison_reverse_final_tree.log 2>&1; echo "rc=$?" >> gpurun_out/r6_suite_poison_reverse_final_tree.log)
tail -4 gpurun_out/r6_suite_poison_reverse_final_tree.log
SKIP_PMC=1 bash tools/gpu_r6_final.sh v9
