#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/kv_trace.py 2>&1 | grep episode
rm -rf gpurun_out/prof_kv
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kv -o kv -- python tools/kv_trace.py > gpurun_out/kv_trace.log 2>&1
DB=$(find gpurun_out/prof_kv -name "*.db" | head -1)
python tools/kv_trace_summary.py "$DB" gpurun_out/kv_trace_summary.txt
cat gpurun_out/kv_trace_summary.txt
find gpurun_out/prof_kv -name "*.db" -delete
