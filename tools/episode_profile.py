"""Three prefix-reuse training episodes at the bench config (Vicuna-7B, B=8, 6 steps) for rocprofv3 --kernel-trace --stats:
where does the time of navillm_amd/episode.py's mode go?  (tools/gpu_prof_episode.sh)"""
import os
import sys
import time
import torch
sys.path.insert(0, os.environ.get("NAVILLM_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (A/B runs: another checkout)
from navillm_amd import config as nvcfg
from navillm_amd.nav_model import NavModel
from navillm_amd.losses import CrossEntropyLoss
from navillm_amd.optim import FlatAdamW
from navillm_amd.synthetic import SyntheticEpisodes, prefix_reuse_episode
dev = torch.device("cuda:0")
cfg = nvcfg.vicuna_7b(image_feat_size=768)
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.train()
opt = FlatAdamW(model, lr=3e-5)
crit = CrossEntropyLoss()
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev)
reps = int(os.environ.get("EPISODE_REPS", "3"))
for rep in range(reps):
    ep.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prefix_reuse_episode(model, crit, ep, 6, teacher_forced=os.environ.get("EPISODE_TF", "1") != "0")
    opt.clip_grad_norm_(40.0); opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    print(f"episode {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms -> {48 / (time.perf_counter() - t0):.1f} nav-steps/s", flush=True)
