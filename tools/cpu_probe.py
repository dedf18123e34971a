"""Host-side probe for bench.py's cpu_baseline: what does the box give a CPU-only torch process?  Prints the CPU budget
(affinity, cgroup quota, load) and times two Vicuna-7B-shaped oracle decoder layers fwd+bwd at several thread counts."""
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "loadavg", os.getloadavg())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu.stat"):
    try:
        print(f, open(f).read().strip().replace("\n", " | ")[:200])
    except OSError:
        pass
print("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), "MKL", os.environ.get("MKL_NUM_THREADS"), "torch threads", torch.get_num_threads())
spec = importlib.util.spec_from_file_location("navillm_oracle", os.path.join(ROOT, "oracle", "navillm_oracle.py"))
O = importlib.util.module_from_spec(spec)
spec.loader.exec_module(O)
from navillm_amd.config import vicuna_7b
from navillm_amd.params import synth_state_dict
c = vicuna_7b(num_layers=2, base_vocab_size=1000)
P = {k: v.requires_grad_(True) for k, v in synth_state_dict(c, 1).items() if k.startswith("lang_model.model")}
x = torch.randn(1, 596, 4096).bfloat16()
am = torch.ones(1, 596, dtype=torch.long)
for nt in (256, 64, 32, 256):
    torch.set_num_threads(nt)
    for rep in range(2):
        t0 = time.time()
        y = O.llama_decoder(P, c, x, am)
        t1 = time.time()
        y.float().sum().backward()
        t2 = time.time()
    print(f"threads {nt:4d}: 2 layers fwd {t1 - t0:.2f} s  bwd {t2 - t1:.2f} s", flush=True)
