"""GPU probe (round 4): the NT GEMMs of a K/V-reuse inference step on weight-only fp8 weights -- bf16 GEMM on a resident de-quantised
operand, the pre-pass (nv_fp8_dequant_rows) + bf16 GEMM, and the tile GEMM on the codes (nv_gemm_fp8w, modes 7 / 8 / 9), same tile.
Usage: python tools/gemm_fp8_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops, fp8
from gemm_probe import bench

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for d, ff in ((5120, 13824), (4096, 11008)):
    for M in [int(x) for x in os.environ.get("PROBE_M", "500,670,800,1000").split(",")]:
        for (N, K) in ((3 * d, d), (d, d), (2 * ff, d), (d, ff)):
            X = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(4)]
            W = [(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() for _ in range(4)]
            QS = [fp8.quantize_rows(w) for w in W]
            WD = [fp8.dequantize_rows(q, s) for q, s in QS]
            scratch = torch.empty_like(W[0])
            fl = 2.0 * M * N * K
            line = f"M={M:5d} N={N:6d} K={K:6d}:"
            for tile in [int(x) for x in os.environ.get("PROBE_TILES", "84,85").split(",")]:
                t_b = bench([lambda i=i: ops.gemm_bf16(0, X[i], WD[i], tile_cfg=tile) for i in range(4)], iters=12)
                t_p = bench([lambda i=i: ops.gemm_bf16(0, X[i], fp8.dequantize_rows(QS[i][0], QS[i][1], out=scratch), tile_cfg=tile) for i in range(4)], iters=12)
                line += f"  tile{tile}: bf16 {t_b * 1e6:6.1f} us ({fl / t_b / 1e12:5.0f} TF) prepass+bf16 {t_p * 1e6:6.1f}"
                for mode in (7, 8, 9):
                    t = bench([lambda i=i: fp8.gemm_fp8w(X[i], QS[i][0], QS[i][1], mode=mode, tile_cfg=tile) for i in range(4)], iters=12)
                    line += f" fp8m{mode} {t * 1e6:6.1f}"
            t0 = bench([lambda i=i: ops.gemm_bf16(0, X[i], WD[i]) for i in range(4)], iters=12)
            line += f"  planned bf16: {t0 * 1e6:6.1f} us"
            print(line, flush=True)
            del X, W, QS, WD
