"""greedy decoding speed at Vicuna-7B, B=8: prefill of a ~600-token prompt, then N single-token steps through the K/V cache."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg
from navillm_amd.nav_model import NavModel
from navillm_amd.kvcache import KVCacheLM
dev = torch.device("cuda:0")
cfg = nvcfg.vicuna_7b()
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.eval()
B, L, N = 8, 600, 24
g = torch.Generator().manual_seed(0)
ids = [[1] + torch.randint(3, cfg.base_vocab_size, (L - 1 + b,), generator=g).tolist() for b in range(B)]
kv = KVCacheLM(model, B, capacity=1024)
for rep in range(2):
    kv.reset()
    seqs = [list(x) for x in ids]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Hs = kv.extend(seqs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for step in range(N):
        nxt = torch.argmax(kv.logits_last(Hs), -1).tolist()
        for b in range(B): seqs[b].append(nxt[b])
        Hs = kv.extend(seqs)
    torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"prefill {B}x~{L}: {(t1-t0)*1e3:.1f} ms; decode: {(t2-t1)/N*1e3:.2f} ms per step of {B} tokens -> {B*N/(t2-t1):.0f} tokens/s "
      f"(weights streamed once per step = {13.5e9/((t2-t1)/N)/1e12:.2f} TB/s effective)")
