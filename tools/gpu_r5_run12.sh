#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lz in 1 0; do
  rm -rf gpurun_out/prof_l$lz
  NAVILLM_EPISODE_LAZY_PREFIX=$lz timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_l$lz -o b -- python bench.py --mode prefix_reuse --steps 18 --warmup 6 --prewarm 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-profile --no-other-mode > gpurun_out/prof_l$lz.log 2>&1
  DB=$(find gpurun_out/prof_l$lz -name "*.db" | head -1)
  python tools/gap_summary.py "$DB" gpurun_out/r05_gaps_lazy$lz.txt
  find gpurun_out/prof_l$lz -name "*.db" -delete
  head -22 gpurun_out/r05_gaps_lazy$lz.txt | cut -c1-170
done
