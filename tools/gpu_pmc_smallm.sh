#!/bin/bash
mkdir -p gpurun_out/pmcs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d gpurun_out/pmcs -o p1 -- python tools/gemm_pmc_smallm.py > gpurun_out/pmcs/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d gpurun_out/pmcs -o p2 -- python tools/gemm_pmc_smallm.py > gpurun_out/pmcs/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcs -o p3 -- python tools/gemm_pmc_smallm.py > gpurun_out/pmcs/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmcs -o p4 -- python tools/gemm_pmc_smallm.py > gpurun_out/pmcs/p4.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/pmcs -o p5 -- python tools/gemm_pmc_smallm.py > gpurun_out/pmcs/p5.log 2>&1
python tools/pmc_sq_summary.py gpurun_out/pmcs gpurun_out/r03_gemm_smallm_pmc.txt
python tools/rocprof_summary.py $(find gpurun_out/pmcs -name "p5*.db" | head -1) gpurun_out/r03_gemm_smallm_durations.txt
find gpurun_out/pmcs -name "*.db" -delete
cat gpurun_out/r03_gemm_smallm_pmc.txt; cat gpurun_out/r03_gemm_smallm_durations.txt
