"""GPU probe: TF/s of the bf16 GEMM family at the Vicuna-7B step shapes, per tile config.
Usage (GPU box): python tools/gemm_probe.py [M] [--cold]
--cold cycles through 12 distinct operand sets (> 256 MiB Infinity Cache) like a real training step."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops


def bench(fns, iters=12, warm=3):
    n = len(fns)
    for i in range(warm):
        fns[i % n]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % n]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cold = "--cold" in sys.argv
    tiles = (0,) if "--t0" in sys.argv else (3,) if "--t3" in sys.argv else ((6, 8) if "--t68" in sys.argv else (1, 2, 3, 4, 6))
    M = int(args[0]) if args else 5600
    nset = 12 if cold else 1
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [(4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008)]
    if "--noblas" in sys.argv:
        pass
    for (N, K) in shapes:
        X = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(nset)]
        W = [(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() for _ in range(nset)]
        dY = [torch.randn(M, N, device=dev, generator=g).bfloat16() for _ in range(nset)]
        G = [torch.zeros(N, K, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
        fl = 2.0 * M * N * K
        line = f"M={M} N={N} K={K} {'cold' if cold else 'hot'}:"
        for tile in tiles:
            t_nt = bench([lambda i=i: ops.gemm_bf16(0, X[i], W[i], tile_cfg=tile) for i in range(nset)])
            t_nn = bench([lambda i=i: ops.gemm_bf16(1, dY[i], W[i], tile_cfg=tile) for i in range(nset)])
            t_tn = bench([lambda i=i: ops.gemm_bf16(2, dY[i], X[i], out=G[i], epilogue=1, tile_cfg=tile) for i in range(nset)])
            line += f"  tile{tile}: NT {fl/t_nt/1e12:7.1f} NN {fl/t_nn/1e12:7.1f} TN+acc {fl/t_tn/1e12:7.1f} TF"
        if "--noblas" not in sys.argv:
            t_ref = bench([lambda i=i: X[i] @ W[i].t() for i in range(nset)])
            t_ref2 = bench([lambda i=i: dY[i] @ W[i] for i in range(nset)])
            t_ref3 = bench([lambda i=i: dY[i].t() @ X[i] for i in range(nset)])
            line += f"  | hipBLASLt NT {fl/t_ref/1e12:7.1f} NN {fl/t_ref2/1e12:7.1f} TN {fl/t_ref3/1e12:7.1f} TF"
        print(line, flush=True)
        del X, W, dY, G


if __name__ == "__main__":
    main()
