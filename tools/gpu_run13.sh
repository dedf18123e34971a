#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=line -p no:cacheprovider -k "gemm_bf16" 2>&1 | tail -3
for o in 0 1; do echo "== order $o"; NV_GEMM_ORDER=$o python tools/gemm_probe.py 5152 --cold --t0 --noblas 2>&1 | grep -v amdgpu.ids; done
for o in 0 1 x; do if [ $o = x ]; then unset NV_GEMM_ORDER; else export NV_GEMM_ORDER=$o; fi; echo "== bench order $o"; python bench.py --steps 12 --warmup 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200; done
