"""standalone timings of the non-GEMM kernels of one decoder layer at the bench shape (B=8, S=656, d=4096, ff=11008),
with the HBM bytes each moves -> achieved GB/s.  python tools/rowop_probe.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
dev = torch.device("cuda:0")
B, S, H, hd, d, ff = 8, 656, 32, 128, 4096, 11008
M = B * S
g = torch.Generator(device=dev).manual_seed(0)
bf = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
def bench(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def report(name, us, nbytes):
    print(f"{name:<28} {us:8.1f} us   {nbytes/1e6:8.1f} MB   {nbytes/us/1e3:7.0f} GB/s", flush=True)
x, dy, w = bf(M, d), bf(M, d), bf(d)
gw = torch.zeros(d, device=dev, dtype=torch.bfloat16)
out, rstd = ops.rmsnorm_fwd(x, w, 1e-6)
report("rmsnorm_fwd", bench(lambda: ops.rmsnorm_fwd(x, w, 1e-6, out=out, rstd=rstd)), 2 * M * d * 2)
dx = torch.empty_like(x)
report("rmsnorm_bwd (+dw reduce)", bench(lambda: ops.rmsnorm_bwd(dy, x, w, rstd, gw, resid_grad=x, out=dx)), 4 * M * d * 2)
gu, dh = bf(M, 2 * ff), bf(M, ff)
h = torch.empty(M, ff, device=dev, dtype=torch.bfloat16); dgu = torch.empty_like(gu)
report("swiglu_fwd", bench(lambda: ops.swiglu_fwd(gu, out=h)), 3 * M * ff * 2)
report("swiglu_bwd", bench(lambda: ops.swiglu_bwd(gu, dh, out=dgu)), 5 * M * ff * 2)
qkv = bf(M, 3 * d)
pos = torch.arange(S, device=dev, dtype=torch.float32)
inv = 1.0 / (10000 ** (torch.arange(0, hd, 2, device=dev, dtype=torch.float32) / hd))
fr = torch.outer(pos, inv); emb = torch.cat([fr, fr], -1)
cos_t, sin_t = emb.cos().bfloat16().contiguous(), emb.sin().bfloat16().contiguous()
report("rope (q,k in place)", bench(lambda: ops.rope_(qkv, cos_t, sin_t, S, H, hd)), 4 * M * d * 2)
kvs = torch.tensor([0, 3, 10, 0, 25, 7, 0, 1], device=dev, dtype=torch.int32)
o = torch.empty(M, d, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
fl = 4.0 * B * H * S * S * hd / 2
t = bench(lambda: ops.attn_fwd(qkv, kvs, B, S, H, hd, out=o, lse2=lse))
report("attn_fwd", t, 4 * M * d * 2); print(f"    causal flops {fl/1e9:.0f} GF -> {fl/t/1e6:.0f} TF")
do = bf(M, d); dqkv = torch.empty_like(qkv)
t = bench(lambda: ops.attn_bwd(qkv, o, do, lse, kvs, B, S, H, hd, dqkv=dqkv))
report("attn_bwd (prep+dkv+dq)", t, 8 * M * d * 2); print(f"    causal flops {2.5*fl/1e9:.0f} GF -> {2.5*fl/t/1e6:.0f} TF")
