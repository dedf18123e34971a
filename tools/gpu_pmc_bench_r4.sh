#!/bin/bash
# round 4: FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only) of the bench command in BOTH training modes
#   -> gpurun_out/r04_gemm_pmc_traffic_{mode}.txt + r04_gemm_traffic_{mode}.json (copy to profiles/)
# each run = two whole 6-step episodes of the mode (prewarm 6 + 6 timed), so the per-launch average is over the mode's own launch mix
TAG=${1:-r04}
mkdir -p gpurun_out/pmcb4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for MODE in ${MODES:-prefix_reuse recompute}; do
  if [ "$MODE" = recompute ]; then ST=3; PW=3; else ST=6; PW=6; fi
  ARGS="--mode $MODE --steps $ST --warmup 0 --prewarm $PW --no-cpu-baseline --no-profile --infer-steps 0 --no-extras --no-other-mode"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcb4 -o fetch_$MODE -- python bench.py $ARGS > gpurun_out/pmcb4/fetch_$MODE.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcb4 -o write_$MODE -- python bench.py $ARGS > gpurun_out/pmcb4/write_$MODE.log 2>&1
  F=$(find gpurun_out/pmcb4 -name "fetch_$MODE*.db" | head -1); W=$(find gpurun_out/pmcb4 -name "write_$MODE*.db" | head -1)
  python tools/pmc_traffic.py "$F" "$W" gpurun_out/${TAG}_gemm_pmc_traffic_$MODE.txt gpurun_out/${TAG}_gemm_traffic_$MODE.json $MODE "$ARGS"
  find gpurun_out/pmcb4 -name "*.db" -delete
done
