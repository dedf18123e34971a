"""cProfile of training nav steps at the bench config (host side only; the GPU runs behind)."""
import sys, os, time, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg
from navillm_amd.nav_model import NavModel
from navillm_amd.losses import CrossEntropyLoss
from navillm_amd.optim import FlatAdamW
from navillm_amd.synthetic import SyntheticEpisodes, nav_step
dev = torch.device("cuda:0")
cfg = nvcfg.vicuna_7b()
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.train()
model.reserve_activations(8, 768)
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev)
crit = CrossEntropyLoss()
opt = FlatAdamW(model, lr=1e-5)
def steps(n):
    for i in range(n):
        nav_step(model, crit, ep, train=True, last=False)
for _ in range(2): steps(3); torch.cuda.synchronize()
t0 = time.perf_counter(); steps(5); th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
print(f"5 steps: host enqueue {th*1e3:.1f} ms, until GPU done {tt*1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable(); steps(5); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:5000])
