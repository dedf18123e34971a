#!/bin/bash
# rocprofv3 kernel stats of greedy decoding (tools/decode_probe.py, 7B bf16, B=8, ~600-token context)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_dec
NAVILLM_DECODE_GRAPH=0 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dec -o dec -- python tools/decode_probe.py $@ > gpurun_out/decode_prof.log 2>&1
DB=$(find gpurun_out/prof_dec -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/decode_kernel_stats.txt
head -24 gpurun_out/decode_kernel_stats.txt
find gpurun_out/prof_dec -name "*.db" -delete
