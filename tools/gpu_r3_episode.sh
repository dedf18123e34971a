#!/bin/bash
# round 3: prefix-reuse episode mode, deferred vs per-step weight gradients (wall time), then a kernel trace of the default
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in 1 0; do
  echo "== NAVILLM_EPISODE_DEFER_WGRAD=$D" >> gpurun_out/r3_episode_ab.log
  NAVILLM_EPISODE_DEFER_WGRAD=$D EPISODE_REPS=4 python tools/episode_profile.py 2>&1 | grep episode >> gpurun_out/r3_episode_ab.log
done
rm -rf gpurun_out/prof_ep
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ep -o ep -- python tools/episode_profile.py > gpurun_out/episode_prof.log 2>&1
DB=$(find gpurun_out/prof_ep -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r03_episode_kernel_stats_v1.txt
find gpurun_out/prof_ep -name "*.db" -delete
cat gpurun_out/r3_episode_ab.log
head -30 gpurun_out/r03_episode_kernel_stats_v1.txt
