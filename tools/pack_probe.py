import time, torch, sys
sys.path.insert(0, "/root/repo")
from navillm_amd import ops
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
B, S = 8, 650
g = torch.Generator().manual_seed(0)
worst = []
for it in range(200):
    S = 600 + int(torch.randint(0, 60, (1,), generator=g))
    pads = torch.randint(0, 40, (B,), generator=g)
    am = (torch.arange(S)[None] >= pads[:, None])
    ids = torch.randint(3, 32000, (B, S), generator=g)
    t0 = time.perf_counter()
    kv_start = (am.int().cumsum(1) == 0).sum(1).to(torch.int32)
    ok = bool((am == (torch.arange(S)[None] >= kv_start[:, None])).all())
    t1 = time.perf_counter()
    keep = torch.nonzero(am.reshape(-1)).view(-1)
    lens = am.sum(1)
    cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0).to(torch.int32)
    pos = torch.arange(S, dtype=torch.int32)[None].expand(B, S).reshape(-1)[keep]
    flat = ids.reshape(-1)[keep]
    t2 = time.perf_counter()
    a = ops.h2d(cu, dev); b = ops.h2d(pos.contiguous(), dev); c = ops.h2d(flat, dev, torch.int32)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    worst.append(((t4 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
worst.sort(reverse=True)
print("worst 5 (total, mask, pack, h2d-enqueue, sync) ms:")
for w in worst[:5]: print("  " + "  ".join(f"{x:7.2f}" for x in w))
print("median total", sorted(x[0] for x in worst)[100])
