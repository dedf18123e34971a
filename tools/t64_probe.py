"""A 64-step teacher-forced prefix-reuse training episode (BASELINE config 4; bench.py::long_horizon_extra's last measurement) alone:
python tools/t64_probe.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg  # noqa: E402
from navillm_amd.losses import CrossEntropyLoss  # noqa: E402
from navillm_amd.nav_model import NavModel  # noqa: E402
from navillm_amd.synthetic import SyntheticEpisodes, prefix_reuse_episode  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
cfg = nvcfg.vicuna_7b(image_feat_size=768)
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.train()
ep = SyntheticEpisodes(cfg, 8, seed=1434, instr_len=512, device=dev, max_frontier=35)
crit = CrossEntropyLoss()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ep.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prefix_reuse_episode(model, crit, ep, 64, teacher_forced=True)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.zero_grad()
    st = model.episode.stats
    print(f"rep {rep}: {8 * 64 / dt:.1f} nav-steps/s ({dt:.2f} s, host {th:.2f} s), segments {st.get('segments_flushed')}, ring {getattr(getattr(ep, '_collator', None), 'RING', '?')}", flush=True)
