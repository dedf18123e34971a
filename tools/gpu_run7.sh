#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k adamw 2>&1 | tail -5
python bench.py --steps 12 --warmup 6 2>&1 | tail -1 | tee gpurun_out/bench3.log
