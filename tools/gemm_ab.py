"""interleaved A/B of GEMM tile configs on one shape: python tools/gemm_ab.py LAYOUT M N K cfgA cfgB ..."""
import sys, os, torch, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
layout, M, N, K = (int(x) for x in sys.argv[1:5])
cfgs = [int(x) for x in sys.argv[5:]]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
nset = 8
if layout == 0: A = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(nset)]; B = [torch.randn(N, K, device=dev, generator=g).bfloat16() for _ in range(nset)]
elif layout == 1: A = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(nset)]; B = [torch.randn(K, N, device=dev, generator=g).bfloat16() for _ in range(nset)]
else: A = [torch.randn(K, M, device=dev, generator=g).bfloat16() for _ in range(nset)]; B = [torch.randn(K, N, device=dev, generator=g).bfloat16() for _ in range(nset)]
if os.environ.get("AB_DATA") == "zeros":
    for t in A + B: t.zero_()
elif os.environ.get("AB_DATA") == "ones":
    for t in A + B: t.fill_(1.0)
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
res = {c: [] for c in cfgs}
def run(c, n=16):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): ops.gemm_bf16(layout, A[i % nset], B[i % nset], out=C, tile_cfg=c)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for c in cfgs: run(c, 8)
for r in range(12):
    for c in cfgs: res[c].append(2.0 * M * N * K / run(c) / 1e12)
print(f"layout {layout} {M}x{N}x{K}: " + "  ".join(f"cfg{c}: med {statistics.median(v):7.1f} max {max(v):7.1f}" for c, v in res.items()))
