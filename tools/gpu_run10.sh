#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=line -p no:cacheprovider -k "gemm_bf16" 2>&1 | tail -6
python tools/gemm_probe.py 5152 --cold --t68 --noblas 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_probe_v5.log
