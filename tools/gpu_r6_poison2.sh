#!/bin/bash
# round 6, second GPU call: the new tests (automatic episodes, overlapped optimizer update, isolation), then the whole GPU suite under
# NAVILLM_POISON=1 in file order
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
(timeout 1200 python -m pytest tests/test_auto_episode_gpu.py tests/test_episode_isolation_gpu.py "tests/test_parity_gpu.py::test_g12_episode_accumulated_gradients_vs_reference" -x -q -s -m gpu > $O/r6_new_tests.log 2>&1; echo "rc=$?" >> $O/r6_new_tests.log)
tail -15 $O/r6_new_tests.log
(NAVILLM_POISON=1 timeout 2400 python -m pytest tests -q -m gpu > $O/r6_suite_poison_fileorder.log 2>&1; echo "rc=$?" >> $O/r6_suite_poison_fileorder.log)
tail -30 $O/r6_suite_poison_fileorder.log
