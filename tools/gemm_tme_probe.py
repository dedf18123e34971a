"""GPU probe for the cut-off tiles of the 256-wide GEMM kernel (TME = 4..8 fragment rows per wave = 128..256 x 256 tiles):
duration of forward (NT) and dgrad (NN) GEMMs per TME (84 / 85: three-stage loop, 94 / 95: the same tiles on the two-stage loop) on (a) the few-hundred-row shapes of K/V-reuse / suffix steps and (b) the
dense training shapes, next to the planner's choice (tile_cfg 0) and its estimate.  -> profiles/r03_gemm_tme_probe.txt
Usage: python tools/gemm_tme_probe.py [quick]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
from gemm_probe import bench

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
d, ff = 4096, 11008
Ms = (500, 670, 860, 1100, 4272, 4760, 4914, 5134) if len(sys.argv) < 2 else (670, 4760)
shapes = (("qkv", 3 * d, d), ("o", d, d), ("gate_up", 2 * ff, d), ("down", d, ff))
for M in Ms:
    for name, N, K in shapes:
        X = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(3)]
        W = [(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() for _ in range(3)]
        dY = [torch.randn(M, N, device=dev, generator=g).bfloat16() for _ in range(3)]
        fl = 2.0 * M * N * K
        for lay, A, B in ((0, X, W), (1, dY, W)):
            line = f"M={M:5d} {name:8s} {'NT' if lay == 0 else 'NN'} N={N if lay == 0 else K:6d} K={K if lay == 0 else N:6d}:"
            best = None
            for tile in (1, 94, 95, 84, 85, 86, 87, 88, 0):
                t = bench([lambda i=i: ops.gemm_bf16(lay, A[i], B[i], tile_cfg=tile) for i in range(3)], iters=12)
                line += f"  {tile if tile else 'plan'}: {t * 1e6:6.1f}us"
                if tile not in (0,) and (best is None or t < best[1]):
                    best = (tile, t)
            line += f"   best {best[0]} {fl / best[1] / 1e12:6.0f} TF"
            print(line, flush=True)
        del X, W, dY
