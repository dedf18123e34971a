#!/bin/bash
# round 3: idle time between kernels of the prefix-reuse training episodes (steady state)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_g
EPISODE_REPS=4 rocprofv3 --kernel-trace -d gpurun_out/prof_g -o g -- python tools/episode_profile.py > gpurun_out/prof_g.log 2>&1
tail -4 gpurun_out/prof_g.log | head -3
DB=$(find gpurun_out/prof_g -name "*.db" | head -1)
python tools/gap_summary.py "$DB" gpurun_out/r03_kernel_gaps_prefix_episode.txt
find gpurun_out/prof_g -name "*.db" -delete
head -40 gpurun_out/r03_kernel_gaps_prefix_episode.txt
