"""one 40-step K/V-reuse inference episode at the bench config (7B, B=8) for rocprofv3: host wall per step printed; the trace's
busy/idle split is computed by tools/kv_trace_summary.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navillm_amd import config as nvcfg
from navillm_amd.nav_model import NavModel
from navillm_amd.losses import CrossEntropyLoss
from navillm_amd.synthetic import SyntheticEpisodes, nav_step
dev = torch.device("cuda:0")
cfg = nvcfg.vicuna_7b()
model = NavModel(nav_config=cfg, device=dev, seed=0)
model.eval()
ep = SyntheticEpisodes(cfg, 8, seed=1234, instr_len=512, device=dev, max_frontier=35)
crit = CrossEntropyLoss()
model.enable_kv_cache(8, 1024)
T = 40
with torch.no_grad():
    for rep in range(2):
        ep.reset(); model.reset_kv_cache()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        marks = []
        for i in range(T):
            nav_step(model, crit, ep, train=False)
            if i in (19, 39):
                torch.cuda.synchronize(); marks.append(time.perf_counter())
        dt = time.perf_counter() - t0
print(f"episode of {T} steps: {dt*1e3:.1f} ms -> {8*T/dt:.1f} nav-steps/s; steps 20-39: {(marks[1]-marks[0])/20*1e3:.2f} ms per step; new rows {model.kv.last_stats['new']}")
