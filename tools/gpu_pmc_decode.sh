#!/bin/bash
mkdir -p gpurun_out/pmcd
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NAVILLM_DECODE_GRAPH=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcd -o fetch -- python tools/decode_probe.py > gpurun_out/pmcd/fetch.log 2>&1
F=$(find gpurun_out/pmcd -name "fetch*.db" | head -1)
python tools/pmc_decode_traffic.py "$F" gpurun_out/r02_decode_pmc_traffic.txt
find gpurun_out/pmcd -name "*.db" -delete
