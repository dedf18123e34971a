#!/bin/bash
# First contact of the N > 1 path with real RCCL, wherever >= 2 GPUs are visible: the tiny model through
# `bench.py --gpus 2` (one process per GPU, torch.distributed.run, the C-ABI communicator nv_comm_* over RCCL, the per-layer
# exchange from inside finish_episode()'s backward), then the world-2 GPU test that is skipped on one-GPU boxes.
# Prints the `dp` block of the JSON line: transport, rs_ag-vs-allreduce calibration, per-rank RCCL version, exchange time per
# optimizer step and the part of it the backward did not hide.
#   tools/scale_smoke.sh [N]        (default N = 2)
set -u
N=${1:-2}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
HAVE=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
if [ "$HAVE" -lt "$N" ]; then
    echo "scale_smoke: $HAVE GPU(s) visible, $N needed -- nothing run"; exit 3
fi
mkdir -p gpurun_out
timeout 900 python bench.py --gpus "$N" --model tiny --steps 12 --warmup 6 --no-cpu-baseline --infer-steps 0 \
    > gpurun_out/scale_smoke_n$N.json 2> gpurun_out/scale_smoke_n$N.err
rc=$?
echo "bench rc=$rc"
python - "$N" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/scale_smoke_n{n}.json"))
except Exception as e:
    print("no JSON line:", e); sys.exit(1)
print("n_gpus", d["n_gpus"], "value", d["value"], d["unit"], "ms/step", d["ms_per_step"])
print("dp", json.dumps(d.get("dp"), indent=1))
assert d["n_gpus"] == int(n) and d.get("dp") and d["dp"].get("exchange"), "no dp block"
PY
rc2=$?
timeout 900 python -m pytest tests/test_dp_gpu.py -q -m gpu -k "world2_real_backward" 2>&1 | tail -3
exit $(( rc | rc2 ))
