#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
(timeout 1500 python -m pytest tests/test_episode_gpu.py tests/test_auto_episode_gpu.py tests/test_episode_isolation_gpu.py tests/test_parity_r4_gpu.py -k "not full_depth" -q -m gpu > $O/r6_tests_nocache2.log 2>&1; echo "rc=$?" >> $O/r6_tests_nocache2.log); tail -4 $O/r6_tests_nocache2.log
(NAVILLM_POISON=1 timeout 900 python -m pytest tests/test_episode_gpu.py tests/test_auto_episode_gpu.py -q -m gpu > $O/r6_tests_nocache2_poison.log 2>&1; echo "rc=$?" >> $O/r6_tests_nocache2_poison.log); tail -4 $O/r6_tests_nocache2_poison.log
