"""GPU parity, round 5: PER-LAYER RE-SYNCHRONISED parity at Vicuna-7B width.

The end-to-end full-depth tests (tests/test_parity_r4_gpu.py) can only assert statistical closeness: a 32-layer random-weight
decoder amplifies last-bit differences to 9-26 output spacings, so a 0.1-sized systematic error in a deep layer would pass.
Here every checked layer starts from the ORACLE's own input to that layer: the oracle (bf16, the reference's rounding points)
runs all 32 layers forward on the host and records the hidden state entering layers 0, 15 and 31; the HIP `LlamaStack` then runs
THAT layer alone (its weights, d = 4096, 32 heads, ff = 11008) on THAT input, forward and backward, against the oracle's
single-layer forward + autograd: nothing accumulates across layers, so single-step tolerances apply --
output as close to the fp32 output as the bf16 oracle's own (x 1.25) and <= 4 bf16 spacings from the bf16 oracle's at the output's
scale (measured on MI355X: see the printed lines; VERDICT r4 suggested 2.5 -- layer 0 measures 2.9 with the HIP output CLOSER to the fp32
truth than the oracle's, 0.077 vs 0.104: two independent bf16 evaluations of a 11008-wide MLP differ by their own roundings), every
gradient (input, the seven weight matrices, the two norm weights) <= 2.1 % from the bf16 oracle's and as close to the fp32 gradient as
the oracle's own bf16 gradient is."""
import numpy as np
import pytest
import torch

from util import load_oracle, bf16_ulps_at_scale

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LAYER_KEYS = ["self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
              "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight", "input_layernorm.weight",
              "post_attention_layernorm.weight"]


def _layer_weights(cfg, k, seed, dst_layer=None):
    """layer k's tensors (named as layer `dst_layer`, default k) from the per-name seeded generator, bf16"""
    from navillm_amd.params import param_specs, synth_tensor
    shapes = {n: s for n, s, _ in param_specs(cfg)}
    out = {}
    for key in LAYER_KEYS:
        src = f"lang_model.model.layers.{k}.{key}"
        dst = f"lang_model.model.layers.{k if dst_layer is None else dst_layer}.{key}"
        out[dst] = synth_tensor(src, shapes[src], seed).bfloat16()
    return out


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_per_layer_resynchronised_forward_backward_7b_width_vs_oracle():
    from navillm_amd import config as nvcfg
    from navillm_amd import functions as Fn
    from navillm_amd import ops
    from navillm_amd.nav_model import NavModel
    from navillm_amd.params import synth_tensor
    O = load_oracle()
    seed = 5
    cfg = nvcfg.vicuna_7b(image_feat_size=768)
    cfg1 = nvcfg.vicuna_7b(image_feat_size=768, num_layers=1)
    d, L = cfg.hidden_size, cfg.num_layers
    g = torch.Generator().manual_seed(99)
    lens = [293, 187]
    B, S = len(lens), max(lens)
    am = torch.zeros(B, S, dtype=torch.long)
    for b, n in enumerate(lens):
        am[b, S - n:] = 1
    real = am.bool()
    x0 = (0.02 * torch.randn(B, S, d, generator=g)).bfloat16()           # embedding-sized rows (N(0, 0.02) init)
    x0[~real] = 0
    norm_w = synth_tensor("lang_model.model.norm.weight", (d,), seed).bfloat16()
    checked = (0, 15, 31)
    # ---- oracle, bf16, all 32 layers forward on the host, one layer's weights resident at a time
    xin = {}
    x = x0
    with torch.no_grad():
        for k in range(L):
            Pk = _layer_weights(cfg, k, seed)
            if k in checked:
                xin[k] = x.clone()
            x = O.llama_decoder(Pk, cfg, x, am, layers=[k], final_norm=False)
            del Pk
    scale_in = {k: float(xin[k][real].float().abs().max()) for k in checked}
    print(f"[per-layer] input scale per checked layer: {scale_in}")
    # ---- the HIP side: a one-layer model of the same width
    m = NavModel(nav_config=cfg1, device=torch.device(DEV), seed=seed)
    m.eval()
    kv_start = torch.tensor([S - n for n in lens], dtype=torch.int32)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    pos = torch.cat([torch.arange(S - n, S, dtype=torch.int32) for n in lens])
    packed = (cu.to(DEV), pos.to(DEV), S)
    G = torch.randn(B, S, d, generator=g) * 0.05                          # upstream gradient of the (normed) layer output
    G[~real] = 0
    for k in checked:
        P16 = _layer_weights(cfg, k, seed, dst_layer=0)
        P16["lang_model.model.norm.weight"] = norm_w
        # oracle: this layer alone, bf16 (reference rounding points) and fp32, forward + autograd
        res = {}
        for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
            Pd = {n: v.to(dt).clone().requires_grad_(True) for n, v in P16.items()}
            xi = xin[k].to(dt).clone().requires_grad_(True)
            out = O.llama_decoder(Pd, cfg1, xi, am)
            (out.float() * G).sum().backward()
            res[tag] = dict(out=out.detach(), dx=xi.grad, **{n: p.grad for n, p in Pd.items()})
        with torch.no_grad():
            assert m.load_reference_state_dict(P16) == len(P16)
        m.zero_grad()
        E = xin[k][real].to(DEV).contiguous().requires_grad_(True)       # packed rows, sample after sample
        Hs = Fn.LlamaStack.apply(E, m, B, S, kv_start.to(DEV), None, packed)
        Hs.backward(G[real].to(DEV).to(torch.bfloat16))
        torch.cuda.synchronize()
        o16, o32 = res["bf16"]["out"][real], res["fp32"]["out"][real]
        ulps = bf16_ulps_at_scale(Hs.detach(), o16)
        e_hip, e_ref = (Hs.detach().float().cpu() - o32).abs().max().item(), (o16.float() - o32).abs().max().item()
        line = f"[per-layer k={k}] output: {ulps:.2f} bf16 spacings from the bf16 oracle; |hip - fp32| {e_hip:.4f} vs |oracle bf16 - fp32| {e_ref:.4f}"
        assert ulps <= 4.0 and e_hip <= 1.25 * e_ref + 1e-3, line
        worst = 0.0
        grads = {"dx": E.grad}
        for n in P16:
            grads[n] = m.store.g(n)
        for n, gh in grads.items():
            r16 = res["bf16"]["dx"][real] if n == "dx" else res["bf16"][n]
            r32 = res["fp32"]["dx"][real] if n == "dx" else res["fp32"][n]
            a, b_, base = _rel(gh, r16), _rel(gh, r32), _rel(r16, r32)
            worst = max(worst, a)
            assert a < 2.1e-2 and b_ < 1.5 * base + 1e-2, (k, n, a, b_, base)
        print(line + f"; worst gradient rel err vs the bf16 oracle's autograd {worst:.4f} over {len(grads)} tensors")
        del res, P16
    del m
    torch.cuda.empty_cache()


def test_abort_after_a_flushed_segment_taints_the_gradients_and_stale_fused_zero_grad(monkeypatch):
    """ADVICE r4.  (medium) `episode_abort()` after a long episode has already FLUSHED a segment leaves `.grad` holding part of the
    episode's gradients while the prefix's own backward never runs: clip / step must refuse until `zero_grad()`.  (low) the fused
    zero-grad of `FlatAdamW.step()` is only valid until the next gradient write: `step(); backward(); zero_grad()` must clear the
    new gradients too."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.optim import FlatAdamW
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    from test_round2_gpu import _mid_cfg
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=3)
    m.eval()
    opt = FlatAdamW(m, lr=1e-4)
    crit = CrossEntropyLoss()
    ep = SyntheticEpisodes(cfg, 2, seed=5, instr_len=60, device=torch.device(DEV))
    m.zero_grad()
    m.begin_episode(ep.prefix_ids())
    nav_step(m, crit, ep, train=True, last=False)
    st = m.episode.stats
    monkeypatch.setenv("NAVILLM_EPISODE_MAX_ROWS", str(st["prefix_rows"] + max(st["suffix_rows"]) + 8))
    for _ in range(3):
        nav_step(m, crit, ep, train=True, last=False)
    monkeypatch.delenv("NAVILLM_EPISODE_MAX_ROWS")
    assert m.episode.stats["segments_flushed"] >= 1
    m.episode_abort()
    assert float(m.store.grad["lm"].float().abs().max()) > 0          # the flushed segments' gradients are there
    with pytest.raises(RuntimeError, match="inconsistent"):
        opt.clip_grad_norm_(40.0)
    with pytest.raises(RuntimeError, match="inconsistent"):
        opt.step()
    opt.zero_grad()
    assert float(m.store.grad["lm"].float().abs().max()) == 0.0
    # an abort with nothing written yet does not taint
    ep.reset()
    m.begin_episode(ep.prefix_ids())
    nav_step(m, crit, ep, train=True, last=False)
    m.episode_abort()
    opt.clip_grad_norm_(40.0)
    # ---- stale fused zero-grad
    ep.reset()
    m.begin_episode(ep.prefix_ids())
    nav_step(m, crit, ep, train=True, last=True)
    m.finish_episode()
    opt.clip_grad_norm_(40.0)
    opt.step()                                     # zeroes the updated segments itself (fused)
    ep.reset()
    m.begin_episode(ep.prefix_ids())
    nav_step(m, crit, ep, train=True, last=True)
    m.finish_episode()                             # new gradients written AFTER step() ...
    assert float(m.store.grad["lm"].float().abs().max()) > 0
    opt.zero_grad()                                # ... and before zero_grad(): everything must go
    assert float(m.store.grad["lm"].float().abs().max()) == 0.0 and float(m.store.grad["f32"].abs().max()) == 0.0
    # the ordinary order keeps the fused path: step(); zero_grad() leaves all-zero gradients as well
    ep.reset()
    m.begin_episode(ep.prefix_ids())
    nav_step(m, crit, ep, train=True, last=True)
    m.finish_episode()
    opt.clip_grad_norm_(40.0)
    opt.step()
    opt.zero_grad()
    assert float(m.store.grad["lm"].float().abs().max()) == 0.0 and float(m.store.grad["f32"].abs().max()) == 0.0


def _slice_batch(d, b, B):
    """sample b of a model-input dict as a batch of one: tensors / arrays with a leading batch dimension and per-sample lists"""
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B:
            out[k] = v[b:b + 1].clone()
        elif isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B:
            out[k] = v[b:b + 1].copy()
        elif isinstance(v, (list, tuple)) and len(v) == B:
            out[k] = [v[b]]
        else:
            out[k] = v
    return out


def test_batch8_accum1_gives_the_gradient_of_batch1_accum8(monkeypatch):
    """the reference's launch line is `--batch_size 1 --gradient_accumulation_step 8`
    (scripts/multi_wo_pretrain.sh:16); every loss is scaled `/ batch_size / gradient_accumulation_step` (mp3d_agent.py:750) and the
    optimizer steps once per window (train.py:68,86-89).  The same episodes as ONE batch (`--batch_size B --gradient_accumulation_step
    1`) carry the same scale 1/B and no op of the path mixes samples, so the accumulated gradient is the same -- which is how B = 1
    gets its GEMM rows back on this hardware without touching the kernels.  Here: B episodes x 3 steps through the reference's
    formulation once as a batch and once one episode at a time (same inputs, same candidate permutations), every gradient buffer
    compared."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.synthetic import SyntheticEpisodes
    from navillm_amd import ops
    from test_round2_gpu import _mid_cfg
    cfg = _mid_cfg()
    dev = torch.device(DEV)
    m = NavModel(nav_config=cfg, device=dev, seed=3)
    m.eval()
    crit = CrossEntropyLoss()
    B, T = 4, 3
    ep = SyntheticEpisodes(cfg, B, seed=21, instr_len=70, device=dev)
    real_randperm = torch.randperm
    drawn = []

    def rec_randperm(n, *a, **k):
        p = real_randperm(n, *a, **k)
        drawn.append(p.clone())
        return p

    # ---- one batch of B, accumulation 1
    monkeypatch.setattr(torch, "randperm", rec_randperm)
    m.zero_grad()
    record = []
    for t in range(T):
        pin = ep.panorama_inputs()
        pano = m("panorama", pin)
        ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        ids, am = ep.tokenise(nav, m.lang_model.cls_token[0])
        nav["input_ids"], nav["attention_mask"] = ids, am
        nav["hist_vis"] = [list(h) for h in nav["hist_vis"]]
        nav["history"] = [list(h) for h in nav["history"]]
        drawn.clear()
        out = m("navigation", nav)
        assert len(drawn) == B
        targets = ep.teacher_targets(nav, t == T - 1)
        (crit(out["fuse_logits"], ops.h2d(targets, dev)) / B / 1).backward()
        record.append(dict(pin={k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in pin.items()},
                           nav={k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in nav.items() if k != "vp_img_embeds"},
                           targets=targets.clone(), perms=[p.clone() for p in drawn], logits=out["fuse_logits"].detach().float().cpu()))
        ep.advance(nav, targets, out["fuse_embeds"])
    torch.cuda.synchronize()
    g_batch = {g: t.detach().float().clone() for g, t in m.store.grad.items()}
    # ---- the same B episodes one at a time, accumulation B
    m.zero_grad()
    worst_logit = 0.0
    for b in range(B):
        for t in range(T):
            r = record[t]
            pin_b = _slice_batch(r["pin"], b, B)
            pano = m("panorama", pin_b)
            nav_b = _slice_batch(r["nav"], b, B)
            nav_b["vp_img_embeds"] = torch.cat([torch.zeros_like(pano["pano_embeds"][:, :1]), pano["pano_embeds"]], 1)
            # (the sample keeps the left padding it had in the batch: same RoPE position frame, so the comparison is not loosened by
            # the frame effect of tests/test_parity_gpu.py's ULPS_FRAME)
            monkeypatch.setattr(torch, "randperm", lambda n, *a, _p=r["perms"][b], **k: _p.clone())
            out = m("navigation", nav_b)
            worst_logit = max(worst_logit, bf16_ulps_at_scale(out["fuse_logits"], r["logits"][b:b + 1, :out["fuse_logits"].shape[1]]))
            (crit(out["fuse_logits"], ops.h2d(r["targets"][b:b + 1], dev)) / 1 / B).backward()
    monkeypatch.setattr(torch, "randperm", real_randperm)
    torch.cuda.synchronize()
    rel = {g: _rel(m.store.grad[g], g_batch[g]) for g in g_batch}
    print(f"[B={B} x accum 1 vs B=1 x accum {B}] logits: worst {worst_logit:.2f} bf16 spacings; gradient buffers rel err {rel}")
    assert worst_logit <= 2.5
    for g, v in rel.items():
        assert v < 1.5e-2, (g, v)
