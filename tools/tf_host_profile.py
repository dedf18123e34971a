"""cProfile of the HOST side of teacher-forced prefix-reuse episodes at the bench config (begin + 6 steps + finish + optimizer), sorted by own
time and by cumulative time.  usage: [TF_B=1] python tools/tf_host_profile.py   (TF_B: prompts per episode, default 8)"""
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
B = int(os.environ.get("TF_B", "8"))
model.reserve_activations(B, 768)
ep = SyntheticEpisodes(cfg, B, seed=1234, instr_len=512, device=dev)
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


ACC = int(os.environ.get("TF_ACC", "1"))
if ACC > 1:
    # an accumulation window (begin_episode(..., accumulate=ACC)): the host side of whole windows -- recording + the batched finish + optimizer
    def window():
        for e in range(ACC):
            model.begin_episode(ep.prefix_ids(), teacher_forced=True, accumulate=ACC)
            for t in range(6):
                nav_step(model, crit, ep, train=True, last=(t == 5), accum=ACC)
            model.finish_episode()
            ep.reset()
        opt.clip_grad_norm_(40.0)
        opt.step()
        opt.zero_grad()

    for _ in range(2):
        window()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    t0 = time.perf_counter()
    for _ in range(3):
        window()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    pr.disable()
    print(f"3 windows of {ACC} episodes x {B} prompts: host {th * 1e3 / 3:.1f} ms per window, with the GPU drained {tt * 1e3 / 3:.1f} ms per window")
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(40)
        print(s.getvalue()[:6500])
    sys.exit(0)
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
