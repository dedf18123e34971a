"""One FULL-DEPTH (32-layer Vicuna-7B) oracle training step on the box's usable CPUs, timed once (bench.py's
cpu_baseline times a subset of the layers and scales).  Usage: python tools/cpu_full_depth.py > profiles/r02_cpu_full_depth.txt"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from navillm_amd import config as C  # noqa: E402

a = types.SimpleNamespace(instr_len=512)
cfg = C.vicuna_7b(image_feat_size=768)
orig_min = min
bench.min = lambda *x: cfg.num_layers if x == (16, cfg.num_layers) else orig_min(*x)     # time all layers
t0 = time.time()
r = bench.cpu_baseline(a, cfg, 1234)
print(f"wall incl. weight generation {time.time() - t0:.1f} s")
print(r)
