import sys, torch
sys.path.insert(0, "/root/repo")
from navillm_amd import ops
dev = torch.device("cuda:0")
def bench(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32064, 4096)):
    Ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(6)]      # rotate: weights come from HBM, not from cache
    x = torch.randn(8, K, device=dev).bfloat16()
    out = torch.empty(8, N, device=dev, dtype=torch.bfloat16)
    i = [0]
    def run():
        i[0] += 1
        ops.gemm_bf16(ops.NT, x, Ws[i[0] % 6], out=out)
    t = bench(run)
    print(f"gemv M=8 N={N} K={K}: {t:6.1f} us  {N*K*2/t/1e6:6.2f} TB/s")
