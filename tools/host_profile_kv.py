"""Where does the HOST time of a K/V-reuse inference nav step go?  (one batch of 8 episodes in flight: the step is host-bound,
profiles/r02_kv_reuse_step_trace.txt).  cProfile over 3 six-step episodes at Vicuna-7B."""
import cProfile
import io
import os
import pstats
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as C
from navillm_amd.nav_model import NavModel
from navillm_amd.losses import CrossEntropyLoss
from navillm_amd.synthetic import SyntheticEpisodes, nav_step

dev = torch.device("cuda:0")
torch.set_num_threads(16)          # as bench.py does: the default (all hardware threads) makes tiny CPU ops 100x slower now and then
cfg = C.vicuna_7b(image_feat_size=768)
m = NavModel(nav_config=cfg, device=dev, seed=0)
m.eval()
crit = CrossEntropyLoss()
B = 8
ep = SyntheticEpisodes(cfg, B, seed=1234, instr_len=512, device=dev)
m.enable_kv_cache(B, capacity=1024)


def episode():
    ep.reset(); m.reset_kv_cache()
    for i in range(6):
        nav_step(m, crit, ep, train=False)


with torch.no_grad():
    for _ in range(2):
        episode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        episode()
    t_host = time.perf_counter() - t0                 # host done enqueueing
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"3 episodes: host returns after {t_host * 1e3:.1f} ms, GPU done after {t_all * 1e3:.1f} ms -> {B * 18 / t_all:.1f} nav-steps/s, "
          f"{t_all / 18 * 1e3:.2f} ms per step of {B}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        episode()
    torch.cuda.synchronize()
    pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
