"""Where does the host fall behind the GPU in a teacher-forced prefix-reuse episode?  Runs the bench's episode loop (B = 8, Vicuna-7B) and
records, at every phase boundary, the host clock and whether the stream is already EMPTY (`stream.query()`: the GPU has run out of queued
work -- it idles until the host's next launch).  usage: python tools/episode_host_probe.py [episodes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as C  # noqa: E402
from navillm_amd.losses import CrossEntropyLoss  # noqa: E402
from navillm_amd.nav_model import NavModel  # noqa: E402
from navillm_amd.optim import FlatAdamW  # noqa: E402
from navillm_amd.synthetic import SyntheticEpisodes, nav_step  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
cfg = C.vicuna_7b(image_feat_size=768)
m = NavModel(nav_config=cfg, device=dev, seed=0)
m.train()
m.reserve_activations(8, 768)
opt = FlatAdamW(m, lr=3e-5)
crit = CrossEntropyLoss()
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 6
st = torch.cuda.current_stream()
rows = []
for e in range(E):
    marks = []

    def mark(tag):
        marks.append((tag, time.perf_counter(), st.query()))
    mark("start")
    m.begin_episode(ep.prefix_ids(), teacher_forced=True)
    mark("begin")
    for t in range(6):
        nav_step(m, crit, ep, train=True, last=(t == 5))
        mark(f"step{t}")
    m.finish_episode()
    mark("finish")
    opt.clip_grad_norm_(40.0)
    opt.step()
    opt.zero_grad()
    mark("optim")
    ep.reset()
    mark("reset")
    rows.append(marks)
torch.cuda.synchronize()
t_end = time.perf_counter()
for e, marks in enumerate(rows):
    t0 = marks[0][1]
    print(f"episode {e}: " + "  ".join(f"{tag}+{(t - t0) * 1e3:6.1f}ms{'*IDLE*' if idle else ''}" for tag, t, idle in marks[1:]))
print(f"wall per episode over the last {E - 2}: {(t_end - rows[2][0][1]) / (E - 2) * 1e3:.1f} ms (host enqueue of an episode: "
      f"{(rows[-1][-1][1] - rows[-1][0][1]) * 1e3:.1f} ms)")
