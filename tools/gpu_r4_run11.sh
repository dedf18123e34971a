#!/bin/bash
# round 4, GPU call 11: pruned tail in the native decoder loop + cached prefix tokenisation: K/V-cache tests, fp8 tests, inference numbers
mkdir -p gpurun_out
python -m pytest tests/test_kvcache_gpu.py tests/test_fp8_gpu.py tests/test_round2_gpu.py -m gpu -q -x > gpurun_out/r4_run11_tests.log 2>&1; echo rc=$? >> gpurun_out/r4_run11_tests.log
tail -5 gpurun_out/r4_run11_tests.log | cut -c1-250
for P in 1 0; do
NV_DECODER_PRUNE_TAIL=$P python bench.py --steps 6 --warmup 0 --prewarm 6 --no-cpu-baseline --no-extras --no-other-mode --no-profile > gpurun_out/r4_run11_bench_p$P.json 2> gpurun_out/r4_run11_bench_p$P.err
python - <<PY
import json
d = json.load(open("gpurun_out/r4_run11_bench_p$P.json"))
print("PRUNE=$P", "value", d["value"], "fwd-only", d["inference_forward_only"]["nav_steps_per_s_per_gpu"], "KV", d["inference_prefix_kv_reuse"]["nav_steps_per_s_per_gpu"], d["inference_prefix_kv_reuse"]["two_batches_in_flight"]["nav_steps_per_s_per_gpu"])
PY
done
