"""average shader clock during the real GEMM kernel (NV_GEMM_DEBUG=4): randn vs zero operands.  python tools/gemm_clock.py"""
import os, sys, torch
os.environ["NV_GEMM_DEBUG"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
dev = torch.device("cuda:0")
M, N, K = 5152, 12288, 4096
g = torch.Generator(device=dev).manual_seed(0)
for data in ("randn", "zeros"):
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    B = torch.randn(N, K, device=dev, generator=g).bfloat16()
    if data == "zeros":
        A.zero_(); B.zero_()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for layout, (a, b) in ((0, (A, B)),):
        for _ in range(20):
            ops.gemm_bf16(layout, a, b, out=C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm_bf16(layout, a, b, out=C)
        e1.record(); torch.cuda.synchronize()
        ws = list(ops._gemm_ws_cache.values())[0]
        ck = ws[4080:4096].view(torch.int64).tolist()
        print(f"{data}: {2.0*M*N*K/(e0.elapsed_time(e1)/20*1e-3)/1e12:7.1f} TF, block 0: {ck[0]} cycles / {ck[1]} ticks -> shader clock {0.1*ck[0]/max(ck[1],1):.2f} GHz")
