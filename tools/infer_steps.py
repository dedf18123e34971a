import sys, os, time, torch
sys.path.insert(0, "/root/repo")
from navillm_amd import config as nvcfg
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
for pack in (True, False):
    model.pack_rows = pack
    for epi in range(3):
        ep.reset(); ts = []
        with torch.no_grad():
            for i in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                nav_step(model, crit, ep, train=False)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print("pack", pack, "episode", epi, " ".join(f"{t:6.1f}" for t in ts), f"-> {48/sum(ts)*1e3:.1f} nav-steps/s")
