#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -6
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --steps 12 --warmup 6 2>&1 | tail -1 | tee gpurun_out/bench6.log
