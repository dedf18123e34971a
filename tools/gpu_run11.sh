#!/bin/bash
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sed -i 's/tile_cfg=3/tile_cfg=0/g' tools/gemm_pmc.py
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
tail -1 gpurun_out/bench_prof.log | cut -c1-400
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc -o f1 -- python tools/gemm_pmc.py > gpurun_out/pmc/f1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc -o f2 -- python tools/gemm_pmc.py > gpurun_out/pmc/f2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc -o f3 -- python tools/gemm_pmc.py > gpurun_out/pmc/f3.log 2>&1
ls gpurun_out/pmc | head -20
