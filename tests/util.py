"""Shared helpers for the parity tests (fixtures -> tensors, oracle import)."""
import os
import sys
import json
import importlib.util
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from navillm_amd import config as nvcfg  # noqa: E402
from navillm_amd.params import synth_state_dict  # noqa: E402

GOLDEN_SEED = 11  # tests/golden/make_golden.py: gen_precision(seed=11)


def usable_cpus():
    """CPUs this process may actually use: min(cpu_count, affinity mask, cgroup quota) -- the MI355X boxes show 256 hardware threads
    behind a cgroup quota of 16 CPUs, and torch's default of one thread per visible CPU then spends its time throttled (the same
    oracle layers: 36 s at 256 threads, 0.7 s at 32; profiles/r02_cpu_probe.txt)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def load_oracle():
    if torch.get_num_threads() > usable_cpus():
        torch.set_num_threads(usable_cpus())          # the oracle is the only CPU-heavy thing a test process runs
    spec = importlib.util.spec_from_file_location("navillm_oracle", os.path.join(ROOT, "oracle", "navillm_oracle.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def gold(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if dtype is not None else t


def tiny_cfg(tag, **over):
    return nvcfg.tiny(precision="fp32" if tag == "fp32" else "amp_bf16", **over)


def tiny_weights(tag, **over):
    cfg = tiny_cfg(tag, **over)
    return cfg, synth_state_dict(cfg, GOLDEN_SEED)


def meta_of(z):
    return json.loads(str(z["meta"]))


def hist_lists(flat, hist_t):
    out, o = [], 0
    for t in hist_t:
        out.append([flat[o + i] for i in range(t)])
        o += t
    return out


def nav_batch_from_gold(z, pano_embeds):
    """Rebuild the `model('navigation', batch)` dict of g3_nav_*.npz around `pano_embeds`."""
    m = meta_of(z)
    vp_img = torch.cat([torch.zeros_like(pano_embeds[:, :1]), pano_embeds], 1)
    return dict(
        gmap_vpids=m["gmap_vpids"], gmap_img_embeds=T(z["gmap_img_embeds"]), gmap_step_ids=T(z["gmap_step_ids"]),
        gmap_pos_fts=T(z["gmap_pos_fts"]), gmap_visited_masks=T(z["gmap_visited_masks"]),
        gmap_masks=T(z["gmap_masks"]), vp_img_embeds=vp_img, pano_masks=T(z["nav_pano_masks"]),
        vp_pos_fts=T(z["vp_pos_fts"]), vp_cand_vpids=m["vp_cand_vpids"],
        hist_vis=hist_lists(T(z["hist_vis_flat"]), m["hist_t"]),
        history=[["<hist>"] * t for t in m["hist_t"]], data_type=["r2r"] * len(m["hist_t"]),
        prompts=m["prompts"],
    ), m


def episode_step_batch(z, meta, t, pano_embeds, hist_vis):
    """`model('navigation', batch)` dict of step t of g12_episode_*.npz around the caller's own `pano_embeds` and the history
    rows `hist_vis` (B lists of [d] tensors) the caller's previous steps produced -- as the rollout loop builds it
    (mp3d_agent.py:702-728)."""
    ms = meta["steps"][t]
    pre = f"s{t}/"
    vp_img = torch.cat([torch.zeros_like(pano_embeds[:, :1]), pano_embeds], 1)
    assert [len(h) for h in hist_vis] == ms["hist_t"]
    return dict(
        gmap_vpids=ms["gmap_vpids"], gmap_img_embeds=T(z[pre + "gmap_img_embeds"]), gmap_step_ids=T(z[pre + "gmap_step_ids"]),
        gmap_pos_fts=T(z[pre + "gmap_pos_fts"]), gmap_visited_masks=T(z[pre + "gmap_visited_masks"]),
        gmap_masks=T(z[pre + "gmap_masks"]), vp_img_embeds=vp_img, pano_masks=T(z[pre + "nav_pano_masks"]),
        vp_pos_fts=T(z[pre + "vp_pos_fts"]), vp_cand_vpids=ms["vp_cand_vpids"],
        hist_vis=[list(h) for h in hist_vis], history=[["<hist>"] * n for n in ms["hist_t"]],
        data_type=["r2r"] * len(ms["hist_t"]), prompts=ms["prompts"],
    ), ms


def grad_fixture_errors(z, prefix, get_grad):
    """relative errors of the gradients stored by make_golden.grad_fixture under `prefix/`:
    -> {name: rel err} over `grad/` (whole tensor), `gradsub/` ([::3, ::5] sub-block) and `rownorm/` entries."""
    errs = {}
    for k in z:
        if not k.startswith(prefix + "/"):
            continue
        kind, _, name = k[len(prefix) + 1:].partition("/")
        if kind not in ("grad", "gradsub", "rownorm"):
            continue
        ref = torch.from_numpy(np.ascontiguousarray(z[k])).float()
        g = get_grad(name).detach().float().cpu()
        if kind == "gradsub":
            g = g[::3, ::5]
        elif kind == "rownorm":
            g = g.norm(dim=1)
        errs[f"{kind}/{name}"] = ((g - ref).norm() / (ref.norm() + 1e-20)).item()
    return errs


def bf16_ulps_at_scale(a, ref):
    """max |a - ref| over the finite entries, in units of the bf16 spacing at the magnitude of the largest |ref| (bf16 has 8
    significant bits: values in [2^k, 2^(k+1)) are 2^(k-7) apart).  For a vector of logits / hidden states whose entries share
    one scale this is the natural unit of "how many roundings apart": 1.0 = the last bit of the largest entries."""
    a = torch.as_tensor(a).float().cpu()
    ref = torch.as_tensor(ref).float().cpu()
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(a), fin), "inf pattern differs"
    scale = ref[fin].abs().max().item()
    ulp = 2.0 ** (int(np.floor(np.log2(max(scale, 1e-30)))) - 7)
    return (a[fin] - ref[fin]).abs().max().item() / ulp


def dropout_masks_from_gold(z, device=None):
    """keep masks of fixture G14 (stored batch-first as uint8 under `mask/<site>`) -> {site: float tensor}; sites are the keys of
    `NavModel.injected_dropout` and of the oracle's `dmasks` (drop_env.view, drop_env.obj, emb.drop, l{i}.attn / drop1 / drop / drop2)"""
    out = {}
    for k in z:
        if k.startswith("mask/"):
            t = torch.from_numpy(np.ascontiguousarray(z[k])).float()
            out[k[5:]] = t.to(device) if device is not None else t
    return out
