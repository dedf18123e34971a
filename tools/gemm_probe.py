"""GPU probe: TF/s of the bf16 GEMM family at the Vicuna-7B step shapes, per tile config.
Usage (GPU box): python tools/gemm_probe.py [M]"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops


def bench(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 5600
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [(4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008)]
    for (N, K) in shapes:
        X = torch.randn(M, K, device=dev, generator=g).bfloat16()
        W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        dY = torch.randn(M, N, device=dev, generator=g).bfloat16()
        fl = 2.0 * M * N * K
        line = f"M={M} N={N} K={K}:"
        for tile in (1, 2, 3):
            t_nt = bench(lambda: ops.gemm_bf16(0, X, W, tile_cfg=tile))
            t_nn = bench(lambda: ops.gemm_bf16(1, dY, W, tile_cfg=tile))
            t_tn = bench(lambda: ops.gemm_bf16(2, dY, X, tile_cfg=tile))
            line += f"  tile{tile}: NT {fl/t_nt/1e12:7.1f} NN {fl/t_nn/1e12:7.1f} TN {fl/t_tn/1e12:7.1f} TF"
        t_ref = bench(lambda: X @ W.t())
        line += f"  | torch(hipBLASLt) NT {fl/t_ref/1e12:7.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
