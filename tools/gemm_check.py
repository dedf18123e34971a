"""correctness of the bf16 GEMM tail variants (knobs from the environment) against an fp32 reference + bitwise against a second run"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K, lay) in ((4744, 12288, 4096, 0), (4744, 4096, 4096, 0), (4744, 4096, 11008, 2), (4744, 22016, 4096, 2), (3000, 12288, 4096, 1), (4744, 4096, 4096, 1)):
    if lay == 0:
        A = torch.randn(M, K, device=dev, generator=g).bfloat16(); B = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        ref = A.float() @ B.float().t(); run = lambda: ops.gemm_bf16(0, A, B)
    elif lay == 1:
        A = torch.randn(M, N, device=dev, generator=g).bfloat16(); B = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        ref = A.float() @ B.float(); run = lambda: ops.gemm_bf16(1, A, B)
    else:
        dY = (torch.randn(M, N, device=dev, generator=g) * 0.1).bfloat16(); X = torch.randn(M, K, device=dev, generator=g).bfloat16()
        ref = dY.float().t() @ X.float()
        def run():
            o = torch.zeros(N, K, device=dev, dtype=torch.bfloat16); ops.gemm_bf16(2, dY, X, out=o, epilogue=1); return o
    got = run(); again = [run() for _ in range(3)]
    torch.cuda.synchronize()
    err = (got.float() - ref).abs().max().item(); sc = ref.abs().max().item()
    same = all(torch.equal(got, a) for a in again)
    tiles = ((got.shape[0] + 255) // 256) * ((got.shape[1] + 255) // 256)
    print(f"layout {lay} M={M} N={N} K={K}: tiles {tiles} rem {tiles % 256}: rel err {err/sc:.5f} repeatable {same}")
