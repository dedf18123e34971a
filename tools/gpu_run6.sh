#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/tests.log
tail -8 gpurun_out/tests.log
for gm in 1 4 8; do echo "== group_m $gm"; NV_GEMM_GROUP_M=$gm python tools/gemm_probe.py 5152 --cold --t234 --noblas 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/gemm_probe_v2.log
