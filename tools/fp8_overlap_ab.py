"""Vicuna-13B inference at B=8, weight-only fp8 (12.7 GB of decoder weights): the de-quantisation pre-pass in line (round 2) vs
overlapped with the previous GEMM on a side stream (round 3), against the bf16 model.  (tools/gpu_r3_fp8_overlap.sh)"""
import os
import sys
import time
import types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as C
from navillm_amd.nav_model import NavModel
from navillm_amd.losses import CrossEntropyLoss
from navillm_amd.synthetic import SyntheticEpisodes, nav_step

dev = torch.device("cuda:0")
cfg = C.vicuna_13b(image_feat_size=768)
m = NavModel(nav_config=cfg, device=dev, seed=0)
m.eval()
m.lang_model.tokenizer = types.SimpleNamespace(eos_token_id=2, unk_token_id=0)
crit = CrossEntropyLoss()
B, STEPS = 8, 6


def measure(tag):
    ep = SyntheticEpisodes(cfg, B, seed=1234, instr_len=512, device=dev)
    r = {}
    with torch.no_grad():
        for rep in range(3):
            ep.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(STEPS):
                nav_step(m, crit, ep, train=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        r["forward_only_B8"] = round(B * STEPS / dt, 2)
        m.enable_kv_cache(B, capacity=1024)
        for rep in range(3):
            ep.reset(); m.reset_kv_cache()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(STEPS):
                nav_step(m, crit, ep, train=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        r["kv_reuse_B8"] = round(B * STEPS / dt, 2)
        m.kv = None
    print(f"{tag:34s} {r}", flush=True)


measure("bf16")
f8 = m.to_fp8_weight_only()
torch.cuda.synchronize()
print(f"decoder weights: {f8.bytes / 1e9:.2f} GB of codes + scales; allocated now {torch.cuda.memory_allocated(dev) / 1e9:.1f} GB", flush=True)
for ov in (False, True, False, True):
    f8.overlap = ov
    measure(f"fp8 weight-only, overlap={int(ov)}")
print(f"allocated with the two panels: {torch.cuda.memory_allocated(dev) / 1e9:.1f} GB")
