"""cProfile of one cached-inference episode at the bench config: where does the host time go?"""
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
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev)
crit = CrossEntropyLoss()
model.enable_kv_cache(8, 1024)
def episode():
    ep.reset(); model.reset_kv_cache()
    with torch.no_grad():
        for i in range(6):
            nav_step(model, crit, ep, train=False)
    torch.cuda.synchronize()
episode()
t0 = time.perf_counter(); episode(); dt = time.perf_counter() - t0
print(f"episode: {dt*1e3:.1f} ms -> {48/dt:.1f} nav-steps/s")
# GPU-only time of the same episode (events around it, host included, vs. sum of kernel time is in rocprof); host profile:
pr = cProfile.Profile(); pr.enable(); episode(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
