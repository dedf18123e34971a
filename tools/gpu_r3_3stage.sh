#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" > gpurun_out/r3_3stage_tests.log 2>&1; echo rc=$? >> gpurun_out/r3_3stage_tests.log
tail -5 gpurun_out/r3_3stage_tests.log
timeout 900 python tools/gemm_tme_probe.py > gpurun_out/r03_gemm_tme_probe_v3.txt 2>&1
cat gpurun_out/r03_gemm_tme_probe_v3.txt
for T in 1 0; do
NV_GEMM_3STAGE=$T python bench.py --steps 12 --warmup 6 --no-extras --no-cpu-baseline --infer-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = d.get('other_mode', {})
print('3stage=$T', d['config']['training_mode'], d['value'], d['ms_per_step'], d['roofline']['frac'], '| other', o.get('mode'), o.get('nav_steps_per_s_per_gpu'), o.get('ms_per_step'), (o.get('roofline') or {}).get('frac'))"
done
