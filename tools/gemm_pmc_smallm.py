"""The few-hundred-row forward GEMM of a suffix / K/V-reuse step (M=670, N=12288, K=4096) under rocprofv3 --pmc: the full tile (TME 8,
tile_cfg 88) and the planner's cut-off tile (TME 5, tile_cfg 85), plus the large-M launch of the same weight for comparison.
tools/gpu_pmc_smallm.sh -> profiles/r03_gemm_smallm_pmc.txt"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
N, K = 12288, 4096
W = [(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() for _ in range(3)]
for M, tiles in ((670, (85, 88)), (4760, (88,))):
    X = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(3)]
    for tile in tiles:
        for i in range(9):
            ops.gemm_bf16(0, X[i % 3], W[i % 3], tile_cfg=tile)
        torch.cuda.synchronize()
