#!/bin/bash
# round 3: FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only) of the bench command -> profiles/r03_gemm_*
mkdir -p gpurun_out/pmcb3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--mode recompute --steps 3 --warmup 0 --prewarm 1 --no-cpu-baseline --no-profile --infer-steps 0 --no-extras --no-other-mode"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcb3 -o fetch -- python bench.py $ARGS > gpurun_out/pmcb3/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcb3 -o write -- python bench.py $ARGS > gpurun_out/pmcb3/write.log 2>&1
F=$(find gpurun_out/pmcb3 -name "fetch*.db" | head -1); W=$(find gpurun_out/pmcb3 -name "write*.db" | head -1)
python tools/pmc_traffic.py "$F" "$W" gpurun_out/r03_gemm_pmc_traffic.txt gpurun_out/r03_gemm_traffic.json
find gpurun_out/pmcb3 -name "*.db" -delete
