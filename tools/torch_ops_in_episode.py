"""Diagnostic: which TORCH ops run device kernels inside a steady-state teacher-forced headline episode + its
optimizer step, with their shapes, call sites and device time?  Everything arithmetic is a libnavillm_hip.so launch; what torch still
launches should be data movement on tiny tensors -- this lists it.  Usage (GPU): python tools/torch_ops_in_episode.py [out.txt]"""
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg
from navillm_amd.nav_model import NavModel
from navillm_amd.losses import CrossEntropyLoss
from navillm_amd.optim import FlatAdamW
from navillm_amd.synthetic import SyntheticEpisodes, prefix_reuse_episode

dev = torch.device("cuda:0")
cfg = nvcfg.vicuna_7b(image_feat_size=768)
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.train()
model.auto_episode = False
opt = FlatAdamW(model, lr=3e-5)
crit = CrossEntropyLoss()
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev)


def episode():
    ep.reset()
    prefix_reuse_episode(model, crit, ep, 6, teacher_forced=True)
    opt.clip_grad_norm_(40.0)
    opt.step()
    opt.zero_grad()


for _ in range(3):
    episode()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    episode()
    torch.cuda.synchronize()

rows = collections.defaultdict(lambda: [0, 0.0])
total_dev = 0.0
for e in prof.events():
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0.0)
    if not dt or not e.name.startswith("aten::"):
        continue
    site = "?"
    for fr in (e.stack or []):
        if "navillm_amd" in fr or "bench.py" in fr:
            site = fr.strip().split("/")[-1][:70]
            break
    key = (e.name, str(e.input_shapes)[:90], site)
    rows[key][0] += 1
    rows[key][1] += dt
    total_dev += dt
out = [f"# torch (aten) ops with device time inside ONE steady-state teacher-forced episode (B = 8, Vicuna-7B) + clip + AdamW + zero_grad",
       f"# total device time of torch ops: {total_dev / 1e3:.2f} ms", f"# {'calls':>5} {'dev_ms':>8}  op  shapes  site"]
for (name, shapes, site), (n, dt) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
    out.append(f"{n:7d} {dt / 1e3:8.3f}  {name}  {shapes}  {site}")
text = "\n".join(out)
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text + "\n")
