import sys, os, time, torch, cProfile, pstats, io
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
with torch.no_grad():
    for i in range(6): nav_step(model, crit, ep, train=False)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for epi in range(3):
        ep.reset()
        for i in range(6): nav_step(model, crit, ep, train=False)
    torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
