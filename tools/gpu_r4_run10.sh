#!/bin/bash
# round 4, GPU call 10: the fp8 tile GEMM on the FULL tile (three stages): parity, probe at prefill sizes, the 13B extra
mkdir -p gpurun_out
python -m pytest tests/test_fp8_gpu.py -m gpu -q -x -s > gpurun_out/r4_run10_fp8.log 2>&1; echo rc=$? >> gpurun_out/r4_run10_fp8.log
grep "gemm_fp8w\|passed\|failed\|rc=\|Error" gpurun_out/r4_run10_fp8.log | cut -c1-330 | tail -16
PROBE_M=1300,2600,5200 PROBE_TILES=85,88 timeout 500 python tools/gemm_fp8_probe.py > gpurun_out/r4_gemm_fp8_probe_fulltile.txt 2>&1; grep "N= 15360\|N=  5120\|N= 27648" gpurun_out/r4_gemm_fp8_probe_fulltile.txt | cut -c1-330
python - <<'PY' > gpurun_out/r4_fp8_13b_extra.json 2> gpurun_out/r4_fp8_13b_extra.err
import json, sys, types, torch
sys.argv = ["bench.py"]
import bench
a = bench.parse()
torch.cuda.set_device(0)
out = bench.fp8_13b_extra(a, torch.device("cuda:0"), 1234)
print(json.dumps(out))
PY
python - <<'PY'
import json
try:
    f = json.load(open("gpurun_out/r4_fp8_13b_extra.json"))
    for k, v in f.items():
        if isinstance(v, dict) and "kv_reuse_B8" in v: print(k, v)
except Exception as e:
    print("extra failed", e); print(open("gpurun_out/r4_fp8_13b_extra.err").read()[-1500:])
PY
