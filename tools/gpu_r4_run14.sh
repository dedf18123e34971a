#!/bin/bash
# round 4, GPU call 14: pruned top layer in the teacher-forced forward / batched backward
mkdir -p gpurun_out
python -m pytest tests/test_episode_gpu.py tests/test_parity_r4_gpu.py tests/test_dp_gpu.py -m gpu -q -x -s -k "teacher_forced or g12_teacher or (mixed_task and True) or (prefix_reuse_episode and True)" > gpurun_out/r4_run14_tests.log 2>&1; echo rc=$? >> gpurun_out/r4_run14_tests.log
grep "teacher-forced\|g12 teacher\|mixed \|passed\|failed\|rc=\|Error" gpurun_out/r4_run14_tests.log | cut -c1-420 | tail -14
for P in 1 0; do
NAVILLM_EPISODE_PRUNE_TOP=$P python bench.py --steps 12 --warmup 0 --prewarm 6 --no-cpu-baseline --no-extras --no-other-mode --infer-steps 0 > gpurun_out/r4_run14_bench_p$P.json 2> gpurun_out/r4_run14_bench_p$P.err
python - <<PY
import json
d = json.load(open("gpurun_out/r4_run14_bench_p$P.json"))
print("PRUNE_TOP=$P", "value", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["by_layout_tflops"], "loss", d["config"]["loss"])
PY
done
