"""which hipBLASLt (Tensile) solutions torch picks for the Vicuna-7B step shapes -- kernel names via rocprofv3 --kernel-trace"""
import torch
dev = torch.device("cuda:0")
import sys
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4744
for (N, K) in ((4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008)):
    X = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16(); dY = torch.randn(M, N, device=dev).bfloat16()
    for _ in range(3):
        a = X @ W.t(); b = dY @ W; c = dY.t() @ X
    torch.cuda.synchronize()
