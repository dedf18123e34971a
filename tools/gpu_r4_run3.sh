#!/bin/bash
# round 4, GPU call 3: teacher-forced batched forward (tests + bench), segment / truncation tests, fp8 tests, host profile of cached
# inference, fill sources
mkdir -p gpurun_out
python -m pytest tests/test_episode_gpu.py -m gpu -q -x -s -k "teacher_forced or long_episode or truncated" > gpurun_out/r4_run3_episode.log 2>&1; echo rc=$? >> gpurun_out/r4_run3_episode.log
grep "teacher-forced\|long episode\|truncation\|passed\|failed\|rc=\|Error" gpurun_out/r4_run3_episode.log | cut -c1-600 | tail -10
python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -x -s -k "g12_teacher or guards" > gpurun_out/r4_run3_g12tf.log 2>&1; echo rc=$? >> gpurun_out/r4_run3_g12tf.log
grep "g12 teacher\|passed\|failed\|rc=\|Error" gpurun_out/r4_run3_g12tf.log | cut -c1-600 | tail -6
python -m pytest tests/test_fp8_gpu.py -m gpu -q -x > gpurun_out/r4_run3_fp8.log 2>&1; echo rc=$? >> gpurun_out/r4_run3_fp8.log
tail -4 gpurun_out/r4_run3_fp8.log | cut -c1-300
python tools/infer_profile.py > gpurun_out/r4_infer_prof.log 2>&1; head -45 gpurun_out/r4_infer_prof.log | cut -c1-200
timeout 300 python tools/find_fills.py > gpurun_out/r4_fill_sources.txt 2>&1; tail -25 gpurun_out/r4_fill_sources.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_v2.json 2> gpurun_out/r04_bench_v2.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04_bench_v2.json"))
r = d["roofline"]
print("HEADLINE", d["config"]["training_mode"], d["config"].get("teacher_forced_forward_batched"), d["value"], d["ms_per_step"], "frac", r["frac"], r["by_layout_tflops"], "gemm share", r["gemm_share_of_step"], "loss", d["config"]["loss"])
o = d.get("other_mode", {})
ro = o.get("roofline") or {}
print("OTHER", o.get("mode"), o.get("nav_steps_per_s_per_gpu"), o.get("ms_per_step"), ro.get("frac"), o.get("error"))
print("WHOLE", d.get("whole_episodes"))
for k in ("inference_prefix_kv_reuse", "fp8_weight_only_13b_config5", "long_horizon_config4"):
    print(k, json.dumps(d.get(k))[:1600])
PY
tail -5 gpurun_out/r04_bench_v2.err | cut -c1-300
