#!/bin/bash
# SQ counters (one pass, --kernel-trace only) of two teacher-forced prefix-reuse training episodes at Vicuna-7B
mkdir -p gpurun_out/pmcsq
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
EPISODE_REPS=2 timeout 900 rocprofv3 --kernel-trace --pmc $CNT -d gpurun_out/pmcsq -o sq -- python tools/episode_profile.py > gpurun_out/pmcsq/sq.log 2>&1
tail -3 gpurun_out/pmcsq/sq.log
DB=$(find gpurun_out/pmcsq -name "sq*.db" | head -1)
python tools/pmc_sq_summary.py "$DB" gpurun_out/${1:-r04}_sq_counters_prefix_episode.txt "rocprofv3 --kernel-trace --pmc $CNT -- python tools/episode_profile.py (2 teacher-forced prefix-reuse training episodes, Vicuna-7B, B=8; MI355X)" epi_bwd epi_fwd attn_bwd attn_fwd gemm_bf16_kernel adamw swiglu rmsnorm | head -60
find gpurun_out/pmcsq -name "*.db" -delete
