"""GPU probe: decode-step weight streamers at the 7B / 13B layer shapes, M = 8 rows: nv_gemv_bf16 vs nv_gemv_fp8w
(run with NV_GEMV_FP8_UNROLL=2|4|8 to compare loads in flight).  Weights rotate over sets larger than the Infinity Cache."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops, fp8
dev = torch.device("cuda:0")


def bench(fn, n=60):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("NV_GEMV_NTILE =", os.environ.get("NV_GEMV_NTILE", "(auto)"))
for d, ff in ((4096, 11008), (5120, 13824)):
    for (N, K) in ((3 * d, d), (d, d), (2 * ff, d), (d, ff)):
        nset = max(2, int(600e6 // (N * K)) + 1)
        Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(nset)]
        Qs = [fp8.quantize_rows(W) for W in Ws]
        x = torch.randn(8, K, device=dev).bfloat16()
        out = torch.empty(8, N, device=dev, dtype=torch.bfloat16)
        i = [0]

        def run16():
            i[0] += 1
            ops.gemm_bf16(ops.NT, x, Ws[i[0] % nset], out=out)

        def run8():
            i[0] += 1
            q, s = Qs[i[0] % nset]
            fp8.gemv_fp8w(x, q, s, out=out)
        t16, t8 = bench(run16), bench(run8)
        print(f"M=8 N={N:6d} K={K:6d}: bf16 {t16:6.1f} us {N * K * 2 / t16 / 1e6:5.2f} TB/s | fp8 {t8:6.1f} us {N * K / t8 / 1e6:5.2f} TB/s  ({t16 / t8:.2f}x)")
        del Ws, Qs
