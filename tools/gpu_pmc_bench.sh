#!/bin/bash
mkdir -p gpurun_out/pmcb
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcb -o fetch -- python bench.py --steps 3 --warmup 0 --prewarm 1 --no-cpu-baseline --no-profile --infer-steps 0 > gpurun_out/pmcb/fetch.log 2>&1
tail -1 gpurun_out/pmcb/fetch.log | cut -c1-200
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcb -o write -- python bench.py --steps 3 --warmup 0 --prewarm 1 --no-cpu-baseline --no-profile --infer-steps 0 > gpurun_out/pmcb/write.log 2>&1
tail -1 gpurun_out/pmcb/write.log | cut -c1-200
ls -la gpurun_out/pmcb
