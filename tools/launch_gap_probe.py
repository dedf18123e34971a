"""How much of a decode step is launch gap?  A 32-layer decoder with TINY layers (d=256, ff=512: every kernel is a few us of work)
run as (a) nv_decoder_extend on a stream (352 dependent launches enqueued by one C++ loop) and (b) the same call captured in a
hipGraph and replayed.  Per-launch cost = step time / 352.  Usage: python tools/launch_gap_probe.py"""
import sys, time
sys.path.insert(0, ".")
import torch
from navillm_amd import config as nvcfg, ops
from navillm_amd.nav_model import NavModel
from navillm_amd.kvcache import KVCacheLM

dev = torch.device("cuda:0")
for (d, H, ff, L) in ((256, 2, 512, 32), (4096, 32, 11008, 32)):
    cfg = nvcfg.NavConfig(hidden_size=d, num_layers=L, num_heads=H, intermediate_size=ff, base_vocab_size=1000, enc_hidden_size=256,
                          enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    m = NavModel(nav_config=cfg, device=dev, seed=1)
    m.eval()
    B = 8
    kv = KVCacheLM(m, B, capacity=256)
    prompt = [[5 + (b + j) % 900 for j in range(40)] for b in range(B)]
    kv.extend(prompt)
    rec = {}
    Lh = ops._L()
    real = Lh.nv_decoder_extend
    class Spy:
        def __call__(self, *a):
            rec["args"] = a
            return real(*a)
    import navillm_amd.kvcache as kvm
    class LW:
        def __getattr__(self, k):
            return Spy() if k == "nv_decoder_extend" else getattr(Lh, k)
    old = ops._L
    ops._L = lambda: LW()
    kv.extend([p + [7] for p in prompt])
    ops._L = old
    a = list(rec["args"])
    nlaunch = 11 * L + 2
    def run(n, stream_ptr):
        a[-1] = stream_ptr
        for _ in range(n):
            rc = real(*a)
            assert rc == 0
    torch.cuda.synchronize()
    run(5, ops._st()); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(50, ops._st()); t_host = time.perf_counter() - t0; torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print(f"d={d} ff={ff} L={L} M=8: stream launches: {t1 / 50 * 1e3:.3f} ms per step ({t1 / 50 / nlaunch * 1e6:.2f} us per launch; host enqueue {t_host / 50 * 1e3:.3f} ms)")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(2, s.cuda_stream)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g, stream=s):
            run(1, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        t2 = time.perf_counter() - t0
        print(f"                      hipGraph replay : {t2 / 50 * 1e3:.3f} ms per step ({t2 / 50 / nlaunch * 1e6:.2f} us per launch)")
    except Exception as e:
        print("graph capture failed:", type(e).__name__, e)
    del kv, m
