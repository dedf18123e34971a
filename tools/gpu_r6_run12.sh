#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
(timeout 1500 python -m pytest tests/test_dp_gpu.py -q -m gpu > $O/r6_tests_dp3.log 2>&1; echo "rc=$?" >> $O/r6_tests_dp3.log); tail -4 $O/r6_tests_dp3.log
NAVILLM_BENCH_REHEARSAL=1 NAVILLM_BUILD_REUSE=1 timeout 600 python bench.py --gpus 2 --model tiny --steps 7 --warmup 1 --prewarm 1 --instr-len 40 --batch 2 --no-cpu-baseline --infer-steps 0 > $O/r6_rehearsal_n2.json 2> $O/r6_rehearsal_n2.err; echo "rehearsal rc=$?"
python - <<PY
import json
d = json.load(open("$O/r6_rehearsal_n2.json"))
print("REHEARSAL n_gpus", d["n_gpus"], "value", d["value"], "episodes:", (d["config"].get("episodes") or "")[:60], "dp", {k: d["dp"].get(k) for k in ("transport", "reduce", "control_plane")})
PY
