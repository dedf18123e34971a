#!/bin/bash
# round 5: the accumulation-window test + the reference launch line with it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_episode_gpu.py -x -q -m gpu -k "accumulation_window or teacher_forced or first_writer or mixed_task" -s 2>&1 | tail -25
timeout 900 python tools/refline_probe.py > gpurun_out/refline_window.json 2> gpurun_out/refline_window.err; echo "probe rc=$?"
tail -5 gpurun_out/refline_window.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/refline_window.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, v.get("nav_steps_per_s_per_gpu"), (v.get("roofline") or {}).get("frac"), v.get("error"))
PY
