import sys, os, time, torch
sys.path.insert(0, "/root/repo")
from navillm_amd import config as nvcfg, ops, functions as Fn
from navillm_amd.nav_model import NavModel
from navillm_amd.losses import CrossEntropyLoss
from navillm_amd.synthetic import SyntheticEpisodes, nav_step
dev = torch.device("cuda:0")
cfg = nvcfg.vicuna_7b()
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.eval()
model.reserve_activations(8, 768)
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev)
crit = CrossEntropyLoss()
T = {}
def wrap(obj, name, key):
    orig = getattr(obj, name)
    def f(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig(*a, **k)
        torch.cuda.synchronize(); T[key] = T.get(key, 0) + (time.perf_counter() - t0) * 1e3
        return r
    setattr(obj, name, f)
wrap(model, "_lm", "lm")
wrap(Fn.LlamaStack, "apply", "stack")
wrap(Fn.EmbedVis, "apply", "embed")
wrap(ops, "h2d", "h2d")
wrap(model, "forward_panorama_per_step", "pano")
for epi in range(4):
    ep.reset()
    with torch.no_grad():
        for i in range(6):
            T.clear()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nav_step(model, crit, ep, train=False)
            torch.cuda.synchronize(); tt = (time.perf_counter() - t0) * 1e3
            print(f"ep{epi} step{i} total {tt:6.1f}  " + "  ".join(f"{k} {v:6.1f}" for k, v in T.items()) + f"  M'={int(model._row_map.numel())}")
