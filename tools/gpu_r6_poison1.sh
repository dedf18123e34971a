#!/bin/bash
# round 6, first GPU call: the re-applied per-step episode-forward change (ea654c2) under NAVILLM_POISON=1, the back-to-back
# isolation test, then the whole GPU suite in file order (the order that failed in round 5)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
(NAVILLM_POISON=1 timeout 900 python -m pytest tests/test_episode_isolation_gpu.py -k back_to_back -x -q -s -m gpu > $O/r6_iso_poison.log 2>&1; echo "rc=$?" >> $O/r6_iso_poison.log)
(timeout 900 python -m pytest tests/test_episode_isolation_gpu.py -k back_to_back -x -q -s -m gpu > $O/r6_iso_plain.log 2>&1; echo "rc=$?" >> $O/r6_iso_plain.log)
(NAVILLM_POISON=1 timeout 900 python -m pytest tests/test_parity_r4_gpu.py -k eight_layer -q -s -m gpu > $O/r6_8layer_poison.log 2>&1; echo "rc=$?" >> $O/r6_8layer_poison.log)
(NAVILLM_POISON=1 timeout 1200 python -m pytest tests/test_episode_gpu.py -q -m gpu > $O/r6_episode_poison.log 2>&1; echo "rc=$?" >> $O/r6_episode_poison.log)
(timeout 1500 python -m pytest tests -q -m gpu > $O/r6_suite_fileorder.log 2>&1; echo "rc=$?" >> $O/r6_suite_fileorder.log)
tail -5 $O/r6_iso_poison.log $O/r6_iso_plain.log $O/r6_8layer_poison.log $O/r6_episode_poison.log $O/r6_suite_fileorder.log
