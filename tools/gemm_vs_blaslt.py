"""this GEMM kernel vs hipBLASLt (through torch.matmul) on the headline episode's shapes, same box, same operands:
    M in {4272 (prefix forward), 4200 (batched teacher-forced forward), 7700 (the episode's batched backward), 5152 (recompute step)}
    N x K in {12288x4096 (q|k|v), 4096x4096 (o), 22016x4096 (gate|up), 4096x11008 (down)}, layouts NT (forward y = x W^T),
    NN (dgrad dx = dy W), TN (wgrad dW = dy^T x, contraction over the M rows).
N(0,1)-scaled operands, 8 rotating operand sets (no cache-resident repeats), interleaved rounds, median of 7.
Also: this kernel with forced tile heights (tile_cfg 85..88 = 160..256 rows) to see what the launch planner leaves on the table.
usage: python tools/gemm_vs_blaslt.py [out.txt]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
NSET = 6
out_lines = []


def say(s):
    print(s, flush=True)
    out_lines.append(s)


def bench(fn, n=6):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


say("# layout  M      N      K     | ours(auto)  blaslt  ratio | forced tile rows 160 / 192 / 224 / 256  (TFLOP/s, median of 7 interleaved rounds)")
wins = cells = 0
for M in (4272, 4200, 7700, 5152):
    for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
        X = [(torch.randn(M, K, device=dev, generator=g)).bfloat16() for _ in range(NSET)]
        W = [(torch.randn(N, K, device=dev, generator=g)).bfloat16() for _ in range(NSET)]
        dY = [(torch.randn(M, N, device=dev, generator=g)).bfloat16() for _ in range(NSET)]
        Cf = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        Cd = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        Cw = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        cases = {
            "NT": (lambda i, c=0: ops.gemm_bf16(ops.NT, X[i % NSET], W[i % NSET], out=Cf, tile_cfg=c), lambda i: torch.matmul(X[i % NSET], W[i % NSET].t(), out=Cf), True),
            "NN": (lambda i, c=0: ops.gemm_bf16(ops.NN, dY[i % NSET], W[i % NSET], out=Cd, tile_cfg=c), lambda i: torch.matmul(dY[i % NSET], W[i % NSET], out=Cd), True),
            "TN": (lambda i, c=0: ops.gemm_bf16(ops.TN, dY[i % NSET], X[i % NSET], out=Cw, tile_cfg=c), lambda i: torch.matmul(dY[i % NSET].t(), X[i % NSET], out=Cw), False),
        }
        for lay, (ours, lib, has_tme) in cases.items():
            fl = 2.0 * M * N * K
            ours(0); lib(0)
            forced = [85, 86, 87, 88] if has_tme else []
            for c in forced:
                ours(0, c)
            r_o, r_l, r_f = [], [], {c: [] for c in forced}
            for _ in range(7):
                r_o.append(fl / bench(ours) / 1e12)
                r_l.append(fl / bench(lib) / 1e12)
                for c in forced:
                    r_f[c].append(fl / bench(lambda i, c=c: ours(i, c), 4) / 1e12)
            o, l_ = statistics.median(r_o), statistics.median(r_l)
            cells += 1
            wins += o >= l_
            ftxt = " / ".join(f"{statistics.median(r_f[c]):6.0f}" for c in forced) if forced else "(wgrad layout: full tile only)"
            say(f"{lay}  {M:5d} {N:6d} {K:6d} | {o:8.0f}  {l_:8.0f}  {o / l_:5.2f} | {ftxt}")
        del X, W, dY, Cf, Cd, Cw
        torch.cuda.empty_cache()
say(f"# this kernel >= hipBLASLt on {wins} of {cells} cells")
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("\n".join(out_lines) + "\n")
