"""Which bf16 zero-fills a teacher-forced headline episode issues (sizes, where from) and how fast torch's fill kernel runs them.
usage: python tools/fill_probe.py"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg  # noqa: E402
from navillm_amd.losses import CrossEntropyLoss  # noqa: E402
from navillm_amd.nav_model import NavModel  # noqa: E402
from navillm_amd.optim import FlatAdamW  # noqa: E402
from navillm_amd.synthetic import SyntheticEpisodes, nav_step  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
for mb in (6, 33, 105, 262):
    t = torch.empty(mb * (1 << 20) // 2, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        t.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        t.zero_()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"torch zero_ of {mb} MB bf16: {us:.1f} us = {mb * (1 << 20) / us / 1e6:.2f} TB/s")
    del t
cfg = nvcfg.vicuna_7b(image_feat_size=768)
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.train()
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev)
crit = CrossEntropyLoss()
opt = FlatAdamW(model, lr=1e-5)


def episode():
    model.begin_episode(ep.prefix_ids(), teacher_forced=True)
    for t in range(6):
        nav_step(model, crit, ep, train=True, last=(t == 5))
    model.finish_episode()
    opt.clip_grad_norm_(40.0)
    opt.step()
    opt.zero_grad()
    ep.reset()


for _ in range(3):
    episode()
torch.cuda.synchronize()
seen = collections.Counter()
real = torch.Tensor.zero_


def spy(self):
    if self.dtype == torch.bfloat16 and self.is_cuda:
        fr = [f for f in traceback.extract_stack(limit=4)[:-1] if "navillm_amd" in f.filename or "tools" in f.filename]
        where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
        seen[(where, self.numel() * 2 >> 20)] += 1
    return real(self)


torch.Tensor.zero_ = spy
episode()
torch.Tensor.zero_ = real
torch.cuda.synchronize()
tot = 0
for (where, mb), n in sorted(seen.items(), key=lambda kv: -kv[0][1] * kv[1]):
    print(f"  {n:3d} x {mb:5d} MB  {where}")
    tot += n * mb
print(f"bf16 zero_() calls in one episode: {sum(seen.values())}, {tot} MB")
