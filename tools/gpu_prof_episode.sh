#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ep
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ep -o ep -- python tools/episode_profile.py > gpurun_out/episode_prof.log 2>&1
DB=$(find gpurun_out/prof_ep -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r02_episode_kernel_stats.txt
grep episode gpurun_out/episode_prof.log
head -24 gpurun_out/r02_episode_kernel_stats.txt
find gpurun_out/prof_ep -name "*.db" -delete
