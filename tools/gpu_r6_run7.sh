#!/bin/bash
# round 6, last verification of the final tree: the suite under poison in one more seeded-random order, then tools/gpu_r6_final.sh
# (suite, smoke, the driver's bench command through the automatic headline loop, kernel trace, PMC passes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
(NAVILLM_POISON=1 timeout 1800 python -m pytest tests -q -m gpu --nv-order random:3 > gpurun_out/r6_suite_poison_random_3_final_tree.log 2>&1; echo "rc=$?" >> gpurun_out/r6_suite_poison_random_3_final_tree.log)
tail -4 gpurun_out/r6_suite_poison_random_3_final_tree.log
bash tools/gpu_r6_final.sh v7
