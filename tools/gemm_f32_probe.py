import sys, torch
sys.path.insert(0, "/root/repo")
from navillm_amd import ops, lib
L = lib.load()
dev = torch.device("cuda:0")
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (lay, M, N, K) in ((0, 288, 1024, 1024), (0, 288, 1024, 4096), (1, 288, 1024, 3072), (1, 288, 1024, 4096), (0, 288, 4096, 1024), (2, 1024, 4096, 288)):
    if lay == 0: A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif lay == 1: A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else: A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    t_split = bench(lambda: ops.gemm_f32(lay, A, B, out=C))
    t_plain = bench(lambda: L.nv_gemm_f32(lay, A.data_ptr(), B.data_ptr(), C.data_ptr(), 0, M, N, K, A.stride(0), B.stride(0), N, 0, ops._st()))
    print(f"layout {lay} {M}x{N}x{K}: with workspace {t_split:6.1f} us, without {t_plain:6.1f} us")
