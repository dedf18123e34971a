#!/bin/bash
mkdir -p gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -d gpurun_out/pmc2 -o a -- python tools/gemm_pmc.py > gpurun_out/pmc2/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d gpurun_out/pmc2 -o b -- python tools/gemm_pmc.py > gpurun_out/pmc2/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT -d gpurun_out/pmc2 -o c -- python tools/gemm_pmc.py > gpurun_out/pmc2/c.log 2>&1
ls gpurun_out/pmc2
