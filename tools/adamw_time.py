"""clip + AdamW + zero_grad of the Vicuna-7B flat store: ms per optimizer step and the implied HBM rate."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as C
from navillm_amd.nav_model import NavModel
from navillm_amd.optim import FlatAdamW
dev = torch.device("cuda:0")
m = NavModel(nav_config=C.vicuna_7b(image_feat_size=768), device=dev, seed=0)
opt = FlatAdamW(m, lr=3e-5)
st = m.store
st.touch_layers(); st.touch("lang_model.model.embed_tokens.weight", "out_head.0.weight", "out_head.0.bias")
st.grad["lm"].normal_(0, 1e-3)
n = st.grad["lm"].numel()
for what in ("clip", "step", "zero"):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        ev[0].record()
        if what == "clip":
            opt.clip_grad_norm_(40.0)
        elif what == "step":
            opt.clip_grad_norm_(40.0); opt.step()
        else:
            opt.zero_grad()
        ev[1].record()
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    print(f"{what:5s}: {min(ts[1:]):.2f} ms (min of 5)  ->", {"clip": f"{2 * n / min(ts[1:]) / 1e9:.2f} TB/s read", "step": f"(clip + AdamW) AdamW alone moves {14 * n / 1e9:.1f} GB", "zero": f"{2 * n / min(ts[1:]) / 1e9:.2f} TB/s written"}[what])
