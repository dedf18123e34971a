"""Diagnostic: which host calls produce the large fill kernels of a prefix-reuse training episode?  Wraps torch's zero / fill entry
points, logs every fill >= 1 MB of the SECOND episode with its call site.  Usage (GPU): python tools/find_fills.py"""
import collections
import os
import sys
import traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg
from navillm_amd.nav_model import NavModel
from navillm_amd.losses import CrossEntropyLoss
from navillm_amd.optim import FlatAdamW
from navillm_amd.synthetic import SyntheticEpisodes, prefix_reuse_episode

LOG = collections.Counter()
ON = [False]


def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "navillm_amd" in f.filename or "bench" in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno}"
    return "?"


def note(kind, t):
    if ON[0] and t.is_cuda and t.numel() * t.element_size() >= (1 << 20):
        LOG[(kind, site(), tuple(t.shape), str(t.dtype))] += 1


_zero, _fill = torch.Tensor.zero_, torch.Tensor.fill_
torch.Tensor.zero_ = lambda self: (note("zero_", self), _zero(self))[1]
torch.Tensor.fill_ = lambda self, v: (note("fill_", self), _fill(self, v))[1]
for name in ("zeros", "zeros_like", "full", "full_like", "ones"):
    orig = getattr(torch, name)
    setattr(torch, name, (lambda o, n: lambda *a, **k: (lambda r: (note(n, r), r)[1])(o(*a, **k)))(orig, name))

dev = torch.device("cuda:0")
cfg = nvcfg.vicuna_7b(image_feat_size=768)
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.train()
opt = FlatAdamW(model, lr=3e-5)
crit = CrossEntropyLoss()
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev)
for rep in range(2):
    ON[0] = rep == 1
    ep.reset()
    prefix_reuse_episode(model, crit, ep, 6)
    opt.clip_grad_norm_(40.0); opt.step(); opt.zero_grad()
torch.cuda.synchronize()
for (kind, where, shape, dt), n in sorted(LOG.items(), key=lambda kv: -kv[1] * int(torch.tensor(kv[0][2]).prod())):
    print(f"{n:4d} x {kind:10s} {where:28s} {shape} {dt}")
