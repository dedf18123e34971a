"""GPU probe: the NT GEMMs of a K/V-reuse inference step (M = the few hundred NEW tokens of a batch) per tile config.
tile 1 = 128x128, tile 8 = 256x256 (+ split-K when the launch is a partial round).  Usage: python tools/gemm_smallm.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
from gemm_probe import bench

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for d, ff in ((4096, 11008), (5120, 13824)):
    for M in (512, 800, 1024, 1536):
        for (N, K) in ((d, d), (3 * d, d), (2 * ff, d), (d, ff)):
            X = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(4)]
            W = [(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() for _ in range(4)]
            fl = 2.0 * M * N * K
            line = f"M={M:5d} N={N:6d} K={K:6d} t256={((M + 255) // 256) * ((N + 255) // 256):4d}:"
            for tile in (1, 8):
                t = bench([lambda i=i: ops.gemm_bf16(0, X[i], W[i], tile_cfg=tile) for i in range(4)], iters=16)
                line += f"  tile{tile}: {t * 1e6:7.1f} us {fl / t / 1e12:7.1f} TF"
            print(line, flush=True)
            del X, W
