"""cProfile of steps 20-39 of a K/V-reuse inference episode (7B, B=8): where the HOST time of a late step goes"""
import sys, os, time, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg
from navillm_amd.nav_model import NavModel
from navillm_amd.losses import CrossEntropyLoss
from navillm_amd.synthetic import SyntheticEpisodes, nav_step
dev = torch.device("cuda:0")
cfg = nvcfg.vicuna_7b()
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.eval()
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev, max_frontier=35)
crit = CrossEntropyLoss()
model.enable_kv_cache(8, 1024)
pr = cProfile.Profile()
with torch.no_grad():
    for rep in range(2):
        ep.reset(); model.reset_kv_cache()
        for i in range(40):
            if rep == 1 and i == 20:
                torch.cuda.synchronize(); t0 = time.perf_counter(); pr.enable()
            nav_step(model, crit, ep, train=False)
        torch.cuda.synchronize()
        if rep == 1:
            pr.disable(); dt = time.perf_counter() - t0
print(f"steps 20-39: {dt/20*1e3:.2f} ms per step (with cProfile overhead)")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
