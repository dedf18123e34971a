"""one full round of 256 tiles (M=N=4096), sweep K: T = a + b*K separates per-tile fixed cost from the K-loop."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def bench(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
LAYOUTS = [int(x) for x in os.environ.get('KS_LAYOUTS', '0,1,2').split(',')]
SHAPES = ((16, 16), (16, 32)) if not os.environ.get('KS_ONE') else ((16, 16),)
for tiles_m, tiles_n in SHAPES:
    M, N = tiles_m * 256, tiles_n * 256
    for layout in LAYOUTS:
        res = []
        for K in (512, 1024, 2048, 4096, 8192):
            nset = 6
            if layout == 0:
                A = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(nset)]
                B = [torch.randn(N, K, device=dev, generator=g).bfloat16() for _ in range(nset)]
            elif layout == 1:
                A = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(nset)]
                B = [torch.randn(K, N, device=dev, generator=g).bfloat16() for _ in range(nset)]
            else:
                A = [torch.randn(K, M, device=dev, generator=g).bfloat16() for _ in range(nset)]
                B = [torch.randn(K, N, device=dev, generator=g).bfloat16() for _ in range(nset)]
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            it = [0]
            def run():
                i = it[0] % nset; it[0] += 1
                ops.gemm_bf16(layout, A[i], B[i], out=C, tile_cfg=int(os.environ.get("KS_CFG", "0")))
            t = bench(run)
            res.append((K, t))
            del A, B
        (k1, t1), (k2, t2) = res[-2], res[-1]
        b = (t2 - t1) / (k2 - k1)
        a = t1 - b * k1
        print(f"tiles {tiles_m}x{tiles_n} layout {layout}: " + "  ".join(f"K={k}: {t:7.1f}us ({2*M*N*k/t/1e6:6.0f}TF)" for k, t in res) +
              f"  | fixed a={a:6.1f}us  per-64-K step b={b*64:6.3f}us", flush=True)
