#!/bin/bash
# rocprofv3 kernel trace of the headline bench command -> tools/step_trace.py summary of one steady-state episode
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_st
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_st -o bench -- python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-extras --infer-steps 0 > gpurun_out/bench_prof_st.log 2>&1
DB=$(find gpurun_out/prof_st -name "*.db" | head -1)
python tools/step_trace.py "$DB" gpurun_out/step_trace.txt
python tools/rocprof_summary.py "$DB" gpurun_out/step_kernel_stats.txt
head -60 gpurun_out/step_trace.txt
find gpurun_out/prof_st -name "*.db" -delete
