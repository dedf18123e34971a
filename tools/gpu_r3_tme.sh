#!/bin/bash
# round 3: cut-off tiles (TME) -- kernel tests, per-shape probe, then A/B of the dense training step and the prefix-reuse episode
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" > gpurun_out/r3_tme_tests.log 2>&1; echo rc=$? >> gpurun_out/r3_tme_tests.log
tail -4 gpurun_out/r3_tme_tests.log
timeout 600 python tools/gemm_tme_probe.py > gpurun_out/r03_gemm_tme_probe.txt 2>&1
cat gpurun_out/r03_gemm_tme_probe.txt
for T in 0 8; do
  echo "== NV_GEMM_TME=$T (0 = planned)" | tee -a gpurun_out/r3_tme_ab.log
  NV_GEMM_TME=$T python bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline --infer-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dense step', d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'))" | tee -a gpurun_out/r3_tme_ab.log
  NV_GEMM_TME=$T EPISODE_REPS=3 python tools/episode_profile.py 2>&1 | grep episode | tee -a gpurun_out/r3_tme_ab.log
done
