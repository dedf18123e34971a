#!/bin/bash
# round 2: host profile + rocprofv3 kernel stats of the K/V-reuse inference episode (tools/infer_profile.py)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/infer_profile.py > gpurun_out/infer_prof.log 2>&1
rm -rf gpurun_out/prof_inf
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_inf -o inf -- python tools/infer_profile.py > gpurun_out/infer_prof_rocprof.log 2>&1
DB=$(find gpurun_out/prof_inf -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r02_infer_kernel_stats.txt
head -14 gpurun_out/r02_infer_kernel_stats.txt
find gpurun_out/prof_inf -name "*.db" -delete
