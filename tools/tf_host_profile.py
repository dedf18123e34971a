"""cProfile of the HOST side of teacher-forced prefix-reuse episodes at the bench config (begin + 6 steps + finish + optimizer), sorted by own
time and by cumulative time.  usage: python tools/tf_host_profile.py"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg  # noqa: E402
from navillm_amd.losses import CrossEntropyLoss  # noqa: E402
from navillm_amd.nav_model import NavModel  # noqa: E402
from navillm_amd.optim import FlatAdamW  # noqa: E402
from navillm_amd.synthetic import SyntheticEpisodes, nav_step  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
cfg = nvcfg.vicuna_7b(image_feat_size=768)
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.train()
model.reserve_activations(8, 768)
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev)
crit = CrossEntropyLoss()
opt = FlatAdamW(model, lr=1e-5)


def episode(steps_only=False):
    model.begin_episode(ep.prefix_ids(), teacher_forced=True)
    for t in range(6):
        nav_step(model, crit, ep, train=True, last=(t == 5))
    if steps_only:
        model.episode_abort()
    else:
        model.finish_episode()
        opt.clip_grad_norm_(40.0)
        opt.step()
    opt.zero_grad()
    ep.reset()


for _ in range(3):
    episode()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(3):
    episode(steps_only=True)            # the six steps' host side alone (what the GPU waits for right after a synchronisation)
th = time.perf_counter() - t0
pr.disable()
torch.cuda.synchronize()
print(f"3 x (begin + 6 steps), host only: {th * 1e3 / 3:.1f} ms per episode = {th * 1e3 / 18:.2f} ms per step")
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:4500])
