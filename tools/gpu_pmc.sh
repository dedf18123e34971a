#!/bin/bash
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python __graft_entry__.py smoke 2>&1 | tail -2
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d gpurun_out/pmc -o p1 -- python tools/gemm_pmc.py > gpurun_out/pmc/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d gpurun_out/pmc -o p2 -- python tools/gemm_pmc.py > gpurun_out/pmc/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc -o p3 -- python tools/gemm_pmc.py > gpurun_out/pmc/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc -o p4 -- python tools/gemm_pmc.py > gpurun_out/pmc/p4.log 2>&1
ls -la gpurun_out/pmc; tail -3 gpurun_out/pmc/p1.log
