"""Where the HOST time of a B = 1 accumulation window goes (no profiler: wall-clock around the big pieces).
usage: python tools/window_host_split.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg, episode as epi_mod  # noqa: E402
from navillm_amd.losses import CrossEntropyLoss  # noqa: E402
from navillm_amd.nav_model import NavModel  # noqa: E402
from navillm_amd.optim import FlatAdamW  # noqa: E402
from navillm_amd.synthetic import SyntheticEpisodes, nav_step  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
cfg = nvcfg.vicuna_7b(image_feat_size=768)
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.train()
B, ACC = int(os.environ.get("TF_B", "1")), int(os.environ.get("TF_ACC", "8"))
ep = SyntheticEpisodes(cfg, B, seed=1234, instr_len=512, device=dev)
crit = CrossEntropyLoss()
opt = FlatAdamW(model, lr=1e-5)
T = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return w


_ab = timed("autograd.backward (heads + every step's encoder graph)", torch.autograd.backward)
if os.environ.get("SPLIT_SYNC") == "1":
    # drain the GPU in front of every backward call: what is measured is then the host's own cost, not time spent blocked on a full
    # launch queue behind the LM batch
    def _ab_sync(*a, _f=_ab, **k):
        torch.cuda.synchronize()
        return _f(*a, **k)
    torch.autograd.backward = _ab_sync
else:
    torch.autograd.backward = _ab
epi_mod.PrefixEpisode._forward_lazy = timed("_forward_lazy (LM forward launches + heads)", epi_mod.PrefixEpisode._forward_lazy)
epi_mod.PrefixEpisode._finish_batched = timed("_finish_batched (all of the window's finish)", epi_mod.PrefixEpisode._finish_batched)
epi_mod.PrefixEpisode._embed_grad = timed("_embed_grad", epi_mod.PrefixEpisode._embed_grad)
epi_mod.PrefixEpisode._seal_window = timed("_seal_window", epi_mod.PrefixEpisode._seal_window)


def window():
    for e in range(ACC):
        t0 = time.perf_counter()
        model.begin_episode(ep.prefix_ids(), teacher_forced=True, accumulate=ACC)
        for t in range(6):
            nav_step(model, crit, ep, train=True, last=(t == 5), accum=ACC)
        T["recording (begin + 6 steps)"] = T.get("recording (begin + 6 steps)", 0.0) + time.perf_counter() - t0
        model.finish_episode()
        ep.reset()
    t0 = time.perf_counter()
    opt.clip_grad_norm_(40.0)
    opt.step()
    opt.zero_grad()
    T["clip + step + zero_grad"] = T.get("clip + step + zero_grad", 0.0) + time.perf_counter() - t0


for _ in range(2):
    window()
torch.cuda.synchronize()
T.clear()
N = 4
t0 = time.perf_counter()
for _ in range(N):
    window()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print(f"{N} windows of {ACC} episodes x {B} prompt(s): host {th * 1e3 / N:.1f} ms per window, GPU drained after {tt * 1e3 / N:.1f} ms per window "
      f"= {N * ACC * 6 * B / tt:.1f} nav-steps/s")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {v * 1e3 / N:8.1f} ms per window  {k}")
