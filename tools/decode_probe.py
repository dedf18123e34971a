"""greedy decoding speed at Vicuna-7B (or --13b, optionally --fp8), B=8: prefill of a ~600-token prompt, then N single-token steps
through the K/V cache -- host loop (token choice on the host, one sync per token) vs device loop (eager) vs device loop replayed
from a hipGraph (navillm_amd/kvcache.py)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg
import navillm_amd.kvcache as kvm
from navillm_amd.nav_model import NavModel
from navillm_amd.kvcache import KVCacheLM
dev = torch.device("cuda:0")
cfg = nvcfg.vicuna_13b() if "--13b" in sys.argv else nvcfg.vicuna_7b()
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.eval()
if "--fp8" in sys.argv:
    model.to_fp8_weight_only()
B = int(next((a.split('=')[1] for a in sys.argv if a.startswith('--batch=')), 8))
L, N = 600, 64
g = torch.Generator().manual_seed(0)
ids = [[1] + torch.randint(3, cfg.base_vocab_size, (L - 1 + b,), generator=g).tolist() for b in range(B)]
wbytes = 2 * cfg.num_layers * (4 * cfg.hidden_size ** 2 + 3 * cfg.hidden_size * cfg.intermediate_size) / (2 if "--fp8" in sys.argv else 1)
kv = KVCacheLM(model, B, capacity=1024)
kv.extend([list(x) for x in ids]); torch.cuda.synchronize()
t0 = time.perf_counter(); kv.reset(); kv.extend([list(x) for x in ids]); torch.cuda.synchronize(); tp = time.perf_counter() - t0
print(f"{'13b' if '--13b' in sys.argv else '7b'}{' fp8' if '--fp8' in sys.argv else ''}: prefill {B}x~{L}: {tp*1e3:.1f} ms")
for tag, devloop, graph in (("host loop", False, False), ("device loop, eager", True, False), ("device loop, hipGraph", True, True)):
    kvm.DEVICE_GREEDY, kvm.USE_HIP_GRAPH = devloop, graph
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = kv.generate(ids, max_new_tokens=N, eos_token_id=-1, pad_token_id=0)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    per = (dt - tp) / N
    print(f"  {tag:<24}: {per*1e3:.2f} ms per step of {B} tokens -> {B/per:.0f} tokens/s  (decoder weights once per step = {wbytes/per/1e12:.2f} TB/s)")
