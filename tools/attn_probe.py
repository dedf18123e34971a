"""attention forward / backward at the recompute-mode shape (B=8, S=656, H=32, head_dim=128): us per call and causal TFLOP/s."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
dev = torch.device("cuda:0")
B, S, H, hd = 8, int(os.environ.get("S", "656")), 32, 128
d, M = H * hd, B * int(os.environ.get("S", "656"))
g = torch.Generator(device=dev).manual_seed(0)
bf = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()


def bench(fn, iters=40, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


qkv, do = bf(M, 3 * d), bf(M, d)
kvs = torch.tensor([0, 3, 10, 0, 25, 7, 0, 1], device=dev, dtype=torch.int32)
o = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
dqkv = torch.empty_like(qkv)
fl = 4.0 * B * H * S * S * hd / 2
t = bench(lambda: ops.attn_fwd(qkv, kvs, B, S, H, hd, out=o, lse2=lse))
print(f"attn_fwd               {t:7.1f} us  {fl / t / 1e6:6.0f} TF")
for var in ("1", "2"):
    os.environ["NV_ATTN_BWD_VARIANT"] = var
    t = bench(lambda: ops.attn_bwd(qkv, o, do, lse, kvs, B, S, H, hd, dqkv=dqkv))
    print(f"attn_bwd variant {var}     {t:7.1f} us  {2.5 * fl / t / 1e6:6.0f} TF   (prep + dK/dV + dQ)")
