"""Launch each GEMM layout a few times (for rocprofv3 --pmc runs). Usage: gemm_pmc.py [M N K]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (5152, 12288, 4096)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(M, K, device=dev, generator=g).bfloat16()
W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
dY = torch.randn(M, N, device=dev, generator=g).bfloat16()
G = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm_bf16(0, X, W, tile_cfg=0)
    ops.gemm_bf16(1, dY, W, tile_cfg=0)
    ops.gemm_bf16(2, dY, X, out=G, epilogue=1, tile_cfg=0)
torch.cuda.synchronize()
