"""GPU, round 2: the parity holes VERDICT r1 listed.

  * real-size scene encoder (h=1024, 16x64, ff=4096, 36 ragged views) against the reference fixture
  * loss AND gradients of object_grounding / summarization / fgr2r / 3dqa training steps against the reference's backward
  * a Vicuna-7B-shaped decoder layer against the oracle, and the FULL 32-layer 7B model's invariants
  * a 64-step long-horizon episode: K/V-reuse logits vs full recompute, map growing to the 100-slot action head
  * the attention parity instrument: with HF's rounding points the distance to the reference's bf16 run collapses
  * optimizer semantics (first-gradient tracking) and the args-constructor / HF-checkpoint path
Every `print` line of a gap is what the asserted tolerance was derived from (measured on MI355X x 1.5)."""
import os
import shutil
import types

import numpy as np
import pytest
import torch

from util import gold, T, tiny_cfg, meta_of, hist_lists, load_oracle, GOLD, GOLDEN_SEED, grad_fixture_errors, bf16_ulps_at_scale
from test_parity_gpu import build, maxerr, relerr, dev, pano_batch, _nav_forward, _nav_vs_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mid_cfg(**over):
    from navillm_amd import config as nvcfg
    kw = dict(hidden_size=512, num_layers=3, num_heads=4, intermediate_size=1408, base_vocab_size=1000,
              enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    kw.update(over)
    return nvcfg.NavConfig(**kw)


# ------------------------------------------------------------------------------------------------ G1 at real size
def test_g1_real_size_scene_encoder_vs_reference():
    for F_ in (1024, 768):
        z = gold(f"g1_encoder_real_F{F_}.npz")
        m = build(tiny_cfg("bf16", enc_hidden_size=1024, enc_num_heads=16, enc_intermediate_size=4096, image_feat_size=F_,
                           obj_feat_size=768))
        with torch.no_grad():
            out = m("panorama", pano_batch(z, with_obj=(F_ == 1024)))
        e = maxerr(out["pano_embeds"], z["pano_embeds"])
        scale = float(np.abs(z["pano_embeds"]).max())
        print(f"[g1-real F={F_}] max|hip-ref|={e:.3e} (scale {scale:.2f})")
        assert e < 3e-5 and e < 1e-3 * scale
        assert np.array_equal(out["pano_masks"].cpu().numpy(), z["pano_masks"])
        if F_ == 1024:
            assert maxerr(out["obj_embeds"], z["obj_embeds"]) < 3e-5
            assert np.array_equal(out["obj_masks"].cpu().numpy(), z["obj_masks"])
        del m


# ------------------------------------------------------------------------------------------------ G10: every mode's backward
def _check_mode_grads(m, prefix, z16, z32, loss, gtol):
    lref = float(z16[prefix + "/loss"])
    assert abs(float(loss.detach()) - lref) <= 0.012 * max(abs(lref), 1.0), (prefix, float(loss.detach()), lref)
    loss.backward()
    torch.cuda.synchronize()
    e16 = grad_fixture_errors(z16, prefix, m.store.g)
    e32 = grad_fixture_errors(z32, prefix, m.store.g)
    base = {k: relerr(T(z16[prefix + "/" + k]), z32[prefix + "/" + k]) for k in e16}
    print(f"[g10 {prefix}] worst rel err vs ref bf16 {max(e16.values()):.4f}, vs ref fp32 {max(e32.values()):.4f} "
          f"(ref bf16 vs ref fp32: {max(base.values()):.4f})")
    for k in e16:
        # bf16-vs-bf16 at identical rounding points; and as close to the reference's fp32 gradient as its own bf16 run is (x1.5)
        assert e16[k] < gtol and e32[k] < 1.5 * base[k] + 0.01, (prefix, k, e16[k], e32[k], base[k])
    with_grad = {str(s) for s in z16[prefix + "/grad_names_with_grad"]}
    for n in m.store.offsets:
        if n not in with_grad:
            assert float(m.store.g(n).float().abs().max()) == 0.0, (prefix, n)
    assert m.store.touched >= with_grad and not (m.store.touched - with_grad), (prefix, m.store.touched ^ with_grad)


@pytest.mark.parametrize("lm_head_chunk", [2048, 16])
def test_g10_training_gradients_of_every_mode_vs_reference(lm_head_chunk, monkeypatch):
    """lm_head_chunk=16: the LM-loss modes' token rows (~100) go through K9 in 16-row chunks -- the chunked lm_head + CE + dH +
    dW accumulation must give the same loss and gradients as one chunk (and as the reference)."""
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd import functions as Fn
    monkeypatch.setattr(Fn, "LM_HEAD_CHUNK_ROWS", lm_head_chunk)
    z16, z32 = gold("g10_grads_bf16.npz"), gold("g10_grads_fp32.npz")
    m = build(tiny_cfg("bf16"))
    B, GT = 3, 1.7e-2          # measured worst 0.0112 (object grounding) on MI355X, x1.5

    def reset():
        m.zero_grad()
        m.store.touched.clear()

    # ---- object grounding (mp3d_agent.py:788-842)
    z = gold("g5_og_bf16.npz")
    meta = meta_of(z)
    reset()
    po = m("panorama", pano_batch(z, True))
    b = dict(obj_embeds=po["obj_embeds"], obj_masks=po["obj_masks"], obj_loc_fts=po["obj_loc_fts"],
             hist_vis=hist_lists(dev(z["hist_vis_flat"]), meta["hist_t"]), input_ids=T(z["input_ids"]),
             attention_mask=T(z["attention_mask"]), prompts=meta["prompts"])
    oo = m("object_grounding", b)
    loss = CrossEntropyLoss()(oo["obj_logits"], T(z16["og/targets"]).to(DEV)) * 0.5 / B / 1
    _check_mode_grads(m, "og", z16, z32, loss, GT)

    # ---- summarization and fgr2r (mp3d_agent.py:845-909)
    z = gold("g5_sum_bf16.npz")
    meta = meta_of(z)
    for prefix, mode, key, hv, dt in (("sum", "summarization", "sum", hist_lists(dev(z["hist_vis_flat"]), meta["hist_t"]), "r2r"),
                                      ("fgr2r", "embodied_qa", "qa", [[] for _ in range(B)], "fgr2r")):
        reset()
        ps = m("panorama", pano_batch(z))
        vp = torch.cat([torch.zeros_like(ps["pano_embeds"][:, :1]), ps["pano_embeds"]], 1)
        out = m(mode, dict(vp_img_embeds=vp, vp_nav_masks=T(z["vp_nav_masks"]), instruction=["x"] * B, answer=["y"] * B, hist_vis=hv,
                           data_type=[dt] * B, input_ids=T(z[key + "_input_ids"]), attention_mask=T(z[key + "_attention_mask"]),
                           token_type_ids=T(z[key + "_token_type_ids"])), training=True)
        _check_mode_grads(m, prefix, z16, z32, out["loss"] * float(z16[prefix + "/coef"]) / B / 1, GT)

    # ---- 3dqa (llava.py:38-42)
    q = gold("g5_qa_bf16.npz")
    feats = [dev(q["features"])[i, :int(n)] for i, n in enumerate(q["feat_lens"])]
    reset()
    out = m("3dqa", dict(features=feats, question=["q"] * B, input_ids=T(q["input_ids"]), attention_mask=T(q["attention_mask"]),
                         token_type_ids=T(q["token_type_ids"])), training=True)
    _check_mode_grads(m, "qa", z16, z32, out.loss * float(z16["qa/coef"]) / 1, GT)


# ------------------------------------------------------------------------------------------------ Vicuna-7B shapes
def test_7b_shaped_layer_vs_oracle():
    """BASELINE config 2's layer shape (d=4096, 32 heads, ff=11008) at the bench's batch (B=8, 512-token instructions, S~650):
    16 / 43 / 86 column tiles, a ragged 21st tile row, split-K tails -- against the oracle in bf16 and fp32."""
    from navillm_amd import config as nvcfg
    cfg = nvcfg.NavConfig(hidden_size=4096, num_layers=1, num_heads=32, intermediate_size=11008, base_vocab_size=1000,
                          enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    _nav_vs_oracle(cfg, 8, 512, 1, "7b-layer")


def test_full_vicuna_7b_training_step_invariants():
    """The FULL 32-layer Vicuna-7B model (BASELINE config 2: B=8, 36x768 views, 512-token instructions) through one training
    step: finite, bit-deterministic across two runs from the same state, a non-zero gradient in every decoder layer's slice
    and every tensor the reference gives a gradient, and the packed-rows LM == the padded [B, S] layout at full width."""
    from navillm_amd import config as nvcfg
    from navillm_amd.nav_model import NavModel
    from navillm_amd.synthetic import SyntheticEpisodes
    from navillm_amd.losses import CrossEntropyLoss
    cfg = nvcfg.vicuna_7b(image_feat_size=768)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=0)
    m.eval()
    crit = CrossEntropyLoss()
    B = 8

    def run(pack):
        m.pack_rows = pack
        m.zero_grad()
        ep = SyntheticEpisodes(cfg, B, seed=1234, instr_len=512, device=torch.device(DEV))
        for b in range(B):
            ep.instr[b] = ep.instr[b][: 512 - 9 * b]                 # ragged prompts: the padded layout has real padding rows
        pin = ep.panorama_inputs()
        pano = m("panorama", pin)
        ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        nav["input_ids"], nav["attention_mask"] = ep.tokenise(nav, "<cls_1>")
        torch.manual_seed(3)
        out = m("navigation", nav)
        tg = ep.teacher_targets(nav, last=False)
        loss = crit(out["fuse_logits"], tg.to(DEV)) / B
        loss.backward()
        torch.cuda.synchronize()
        return out["fuse_logits"].detach().float().cpu(), {g: t.detach().clone() for g, t in m.store.grad.items()}, float(loss.detach()), \
            nav["input_ids"].shape[1]

    l1, g1, loss1, S = run(True)
    l2, g2, loss2, _ = run(True)
    assert S > 560 and np.isfinite(loss1)
    fin = torch.isfinite(l1)
    assert fin.any(dim=1).all()
    assert torch.equal(l1[fin], l2[fin]) and torch.equal(torch.isfinite(l2), fin), "logits differ between two identical runs"
    for g in g1:
        assert torch.isfinite(g1[g].float()).all()
        assert torch.equal(g1[g], g2[g]), f"gradient buffer {g} is not bit-deterministic"
    st = m.store
    for i in range(cfg.num_layers):
        s, e = st.layer_slice(i)
        assert float(g1["lm"][s:e].float().abs().max()) > 0, f"decoder layer {i}: all-zero gradient slice"
    for n in st.touched:
        o, k = st.offsets[n], st.sizes[n]
        assert float(g1[st.group_of[n]][o:o + k].float().abs().max()) > 0, n
    assert "lang_model.lm_head.weight" not in st.touched and "og_head.0.weight" not in st.touched
    # packed rows vs the reference's padded [B, S] layout at full depth.  The two layouts run different GEMM tilings, i.e. differ in
    # the last bit of some bf16 activations; a RANDOM-weight 32-layer decoder (per-layer gain 0.02*sqrt(4096) = 1.28) amplifies
    # such flips layer after layer, so at full depth only statistical closeness can be asserted (a wrong position / mask /
    # row map would decorrelate the outputs completely: gap ~ logit scale, gradient error ~ 1.4).  The tight equality is
    # asserted where it is meaningful: test_packed_rows_match_padded_layout (3 layers) and the two-layer 7B-width test below.
    lp, gp, _, _ = run(False)
    assert torch.equal(torch.isfinite(lp), fin)
    gap = (lp[fin] - l1[fin]).abs().max().item()
    scale = l1[fin].abs().max().item()
    rels = {g: ((gp[g].float() - g1[g].float()).norm() / (g1[g].float().norm() + 1e-20)).item() for g in g1}
    print(f"[7b full] S={S} loss={loss1:.4f} logit scale {scale:.2f}; packed vs padded: logits {gap:.4f}, grad rel {rels}")
    assert gap < 0.25 * scale and max(rels.values()) < 0.5
    m.pack_rows = True
    del m
    torch.cuda.empty_cache()


def test_7b_width_two_layers_packed_rows_equal_padded_layout():
    """packed rows == padded layout at the FULL Vicuna-7B width (d=4096, 32 heads, ff=11008, B=8, S~650, ragged prompts), with two
    decoder layers so that rounding flips are not amplified: logits and gradients must agree tightly."""
    from navillm_amd import config as nvcfg
    from navillm_amd.nav_model import NavModel
    from navillm_amd.synthetic import SyntheticEpisodes
    from navillm_amd.losses import CrossEntropyLoss
    cfg = nvcfg.vicuna_7b(image_feat_size=768, num_layers=2)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=0)
    m.eval()
    crit = CrossEntropyLoss()
    B, res = 8, {}
    for pack in (True, False):
        m.pack_rows = pack
        m.zero_grad()
        ep = SyntheticEpisodes(cfg, B, seed=1234, instr_len=512, device=torch.device(DEV))
        for b in range(B):
            ep.instr[b] = ep.instr[b][: 512 - 9 * b]
        pin = ep.panorama_inputs()
        pano = m("panorama", pin)
        ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        nav["input_ids"], nav["attention_mask"] = ep.tokenise(nav, "<cls_1>")
        torch.manual_seed(3)
        out = m("navigation", nav)
        (crit(out["fuse_logits"], ep.teacher_targets(nav, last=False).to(DEV)) / B).backward()
        torch.cuda.synchronize()
        res[pack] = (out["fuse_logits"].detach().float().cpu(), {g: t.detach().float().clone() for g, t in m.store.grad.items()})
    lp, lf = res[True][0], res[False][0]
    fin = torch.isfinite(lf)
    assert torch.equal(torch.isfinite(lp), fin)
    gap, u = (lp[fin] - lf[fin]).abs().max().item(), bf16_ulps_at_scale(lp, lf)
    rels = {g: ((res[True][1][g] - res[False][1][g]).norm() / (res[False][1][g].norm() + 1e-20)).item() for g in ("lm", "f32")}
    print(f"[7b width x2 layers] packed vs padded: logits {gap:.5f} = {u:.2f} bf16 ulps, grad rel {rels}")
    assert u <= 3.0 and max(rels.values()) < 2e-2
    m.pack_rows = True
    del m
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ long horizon (config 4)
def test_64_step_episode_kv_reuse_vs_full_recompute():
    """BASELINE config 4: a 64-step episode (history K/V reuse over 64 steps, the map growing until the 100-way action head
    is full: 1 stop + 64 visited + <= 35 frontier slots).  At every step the no-grad navigation runs through the K/V cache;
    at every 8th step the same inputs are also recomputed from scratch and compared; steps 31 and 63 also against the oracle."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.params import synth_state_dict
    from navillm_amd.synthetic import SyntheticEpisodes
    from navillm_amd import config as nvcfg
    O = load_oracle()
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=5)
    m.eval()
    P16 = synth_state_dict(cfg, 5)
    with torch.no_grad():
        m.load_reference_state_dict(P16)
    P32 = {k: v.float() for k, v in P16.items()}
    cfg32 = nvcfg.NavConfig(**{**cfg.__dict__, "precision": "fp32"})
    B, T_ = 2, 64
    ep = SyntheticEpisodes(cfg, B, seed=5, instr_len=300, device=torch.device(DEV), max_frontier=35)
    kv = m.enable_kv_cache(B, capacity=1024)
    worst, worst_o, new_tokens, Gs = 0.0, 0.0, [], []
    for t in range(T_):
        pin = ep.panorama_inputs()
        with torch.no_grad():
            pano = m("panorama", pin)
        ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        ids, am = ep.tokenise(nav, "<cls_1>")
        nav["input_ids"], nav["attention_mask"] = ids, am
        assert ids.shape[1] <= 1024
        G = nav["gmap_masks"].shape[1]
        Gs.append(G)
        assert int((nav["_gmask_cpu"] & ~nav["_gvis_cpu"]).sum(1).max()) <= 100
        torch.manual_seed(900 + t)
        with torch.no_grad():
            out = m("navigation", nav)
        new_tokens.append(max(kv.last_stats["new"]))
        if t % 8 == 7:
            m.kv = None
            torch.manual_seed(900 + t)
            with torch.no_grad():
                full = m("navigation", nav)
            m.kv = kv
            worst = max(worst, maxerr(out["fuse_logits"], full["fuse_logits"].float().cpu()))
            if t in (31, 63):
                cpu = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in nav.items()}
                cpu["hist_vis"] = [[v.cpu() for v in vis] for vis in nav["hist_vis"]]
                torch.manual_seed(900 + t)
                with torch.no_grad():
                    o32 = O.navigation(P32, cfg32, cpu, ids, am)
                    torch.manual_seed(900 + t)
                    o16 = O.navigation(P16, cfg, cpu, ids, am)
                e_ref = maxerr(o16["fuse_logits"], o32["fuse_logits"])
                e_kv, e_full = maxerr(out["fuse_logits"], o32["fuse_logits"]), maxerr(full["fuse_logits"], o32["fuse_logits"])
                print(f"[t64 step {t}] S={ids.shape[1]} G={G} |kv-orc32|={e_kv:.4f} |full-orc32|={e_full:.4f} |orc16-orc32|={e_ref:.4f}")
                assert e_kv <= 1.5 * e_ref + 3e-3 and e_full <= 1.5 * e_ref + 3e-3
                worst_o = max(worst_o, e_kv)
        targets = ep.teacher_targets(nav, last=False)
        ep.advance(nav, targets, out["fuse_embeds"])
    print(f"[t64] G: {Gs[0]} -> {max(Gs)}; new tokens per step after the prefill: max {max(new_tokens[1:])}, first {new_tokens[0]}; "
          f"max |kv - full recompute| over steps 7,15,..,63 = {worst:.5f}")
    assert 90 <= max(Gs) <= 110
    assert max(new_tokens[1:]) < 260 and new_tokens[0] > 300          # only the first step is a full prefill
    assert worst < 1.5e-2          # measured 0.0098 (about one bf16 spacing of the logits)
    m.kv = None


# ------------------------------------------------------------------------------------------------ attention rounding points
def _hf_eager_attention(qkv, B, S, H, hd, kv_start):
    """HF eager LlamaAttention core in bf16 on packed qkv [B*S, 3*H*hd] (CPU, same rounding points as modeling_llama)."""
    q, k, v = (qkv.view(B, S, 3, H, hd)[:, :, i].transpose(1, 2) for i in range(3))       # [B,H,S,hd] bf16
    w = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    neg = torch.finfo(torch.bfloat16).min
    allowed = torch.tril(torch.ones(S, S, dtype=torch.bool))[None] & (torch.arange(S)[None, :] >= kv_start[:, None])[:, None, :]
    w = w + torch.zeros(B, 1, S, S, dtype=torch.bfloat16).masked_fill(~allowed[:, None], neg)
    w = torch.softmax(w, dim=-1, dtype=torch.float32).to(torch.bfloat16)
    return torch.matmul(w, v).transpose(1, 2).reshape(B * S, H * hd)


def test_attention_parity_instrument_matches_hf_eager_rounding():
    """kernel level: nv_attn_fwd_hfround_bf16 == HF eager attention (bf16) to ~1 ulp, while the product kernel (fp32 scores)
    sits a few ulps away from it -- and closer to the fp32 truth."""
    from navillm_amd import ops
    torch.manual_seed(0)
    B, S, H, hd = 2, 330, 4, 128
    qkv = (torch.randn(B * S, 3 * H * hd) * 1.5).to(torch.bfloat16)
    kvs = torch.tensor([0, 37], dtype=torch.int32)
    ref16 = _hf_eager_attention(qkv, B, S, H, hd, kvs).float()
    q32 = qkv.float()
    qf, kf, vf = (q32.view(B, S, 3, H, hd)[:, :, i].transpose(1, 2) for i in range(3))
    allowed = torch.tril(torch.ones(S, S, dtype=torch.bool))[None] & (torch.arange(S)[None, :] >= kvs[:, None])[:, None, :]
    w = (qf @ kf.transpose(2, 3) * hd ** -0.5).masked_fill(~allowed[:, None], float("-inf"))
    ref32 = (torch.softmax(w, -1) @ vf).transpose(1, 2).reshape(B * S, H * hd)
    real = (torch.arange(S)[None] >= kvs[:, None]).reshape(-1)
    d = qkv.to(DEV)
    out_p = torch.empty((B * S, H * hd), dtype=torch.bfloat16, device=DEV)
    out_h = torch.empty_like(out_p)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=DEV)
    ops.attn_fwd(d, kvs.to(DEV), B, S, H, hd, out=out_p, lse2=lse)
    ops.attn_fwd_hfround(d, kvs.to(DEV), None, B, S, H, hd, out_h, lse)
    torch.cuda.synchronize()
    op, oh = out_p.float().cpu()[real], out_h.float().cpu()[real]
    r16, r32 = ref16[real], ref32[real]
    e_h16, e_p16 = (oh - r16).abs().max().item(), (op - r16).abs().max().item()
    e_h32, e_p32, e_r = (oh - r32).abs().max().item(), (op - r32).abs().max().item(), (r16 - r32).abs().max().item()
    print(f"[attn] vs HF-eager bf16: instrument {e_h16:.5f}, product {e_p16:.5f};  vs fp32 truth: instrument {e_h32:.5f}, "
          f"product {e_p32:.5f}, HF-eager bf16 itself {e_r:.5f}")
    assert e_h16 <= 0.5 * e_p16 + 1e-6 or e_h16 < 2 ** -8 * r16.abs().max().item()
    assert e_p32 <= e_r + 2 ** -9 * r32.abs().max().item()     # the product kernel is at least as accurate as HF's bf16 path


def test_bf16_gap_to_reference_is_last_bit_of_the_output_not_attention_rounding():
    """End to end.  VERDICT r1 asked to PROVE what separates the HIP logits from the reference's bf16 logits (DESIGN r1 blamed the
    flash kernel's fp32 scores).  Measured here: it is not the attention.  With the attention forward switched to HF's exact
    rounding points (the parity instrument) the distance does not move -- at the tiny fixture AND at the multi-tile size --
    and in both modes it is about ONE spacing of the bf16 logits themselves: the two runs agree up to the last bit of the
    bf16 output (different fp32 accumulation orders inside the GEMMs flip a final rounding now and then)."""
    zb = gold("g3_nav_bf16.npz")
    m = build(tiny_cfg("bf16"))
    l16 = T(zb["fuse_logits"])
    res = {}
    for mode in (False, True):
        m.attn_hf_rounding = mode
        with torch.no_grad():
            _, out, _ = _nav_forward(m, zb)
        res[mode] = (maxerr(out["fuse_logits"], l16), bf16_ulps_at_scale(out["fuse_logits"], l16))
    m.attn_hf_rounding = False
    print(f"[g3 rounding] |hip-ref_bf16|: product kernel {res[False][0]:.5f} = {res[False][1]:.2f} ulps, HF rounding points "
          f"{res[True][0]:.5f} = {res[True][1]:.2f} ulps")
    assert res[False][1] <= 2.5 and res[True][1] <= 2.5
    # mid-size vs the bf16 oracle (same rounding points as HF)
    from navillm_amd.params import synth_state_dict
    from navillm_amd.synthetic import SyntheticEpisodes
    from navillm_amd.nav_model import NavModel
    O = load_oracle()
    cfg = _mid_cfg()
    mm = NavModel(nav_config=cfg, device=torch.device(DEV), seed=5)
    mm.eval()
    P16 = synth_state_dict(cfg, 5)
    with torch.no_grad():
        mm.load_reference_state_dict(P16)
    ep = SyntheticEpisodes(cfg, 4, seed=77, instr_len=200, device=torch.device(DEV))
    pin = ep.panorama_inputs()
    with torch.no_grad():
        pano = mm("panorama", pin)
    ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
    nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
    nav["input_ids"], nav["attention_mask"] = ep.tokenise(nav, "<cls_1>")
    cpu = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in nav.items()}
    torch.manual_seed(100)
    with torch.no_grad():
        o16 = O.navigation(P16, cfg, cpu, nav["input_ids"], nav["attention_mask"])
    gaps = {}
    for mode in (False, True):
        mm.attn_hf_rounding = mode
        torch.manual_seed(100)
        with torch.no_grad():
            lg = mm("navigation", nav)["fuse_logits"]
        gaps[mode] = (maxerr(lg, o16["fuse_logits"]), bf16_ulps_at_scale(lg, o16["fuse_logits"]))
    print(f"[mid rounding] |hip-oracle_bf16|: product kernel {gaps[False][0]:.5f} = {gaps[False][1]:.2f} ulps, HF rounding points "
          f"{gaps[True][0]:.5f} = {gaps[True][1]:.2f} ulps (logit scale {float(o16['fuse_logits'][torch.isfinite(o16['fuse_logits'])].abs().max()):.2f})")
    assert gaps[False][1] <= 2.5 and gaps[True][1] <= 2.5


# ------------------------------------------------------------------------------------------------ optimizer semantics
def test_adamw_updates_only_parameters_that_ever_had_a_gradient():
    """torch.optim.AdamW in the reference (tools/optims.py:43-45) skips parameters whose .grad is None and counts steps per
    parameter: og_head never moves, lm_head only after the first LM-loss backward, with its OWN bias-correction step."""
    from navillm_amd.optim import FlatAdamW
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    O = load_oracle()
    cfg = tiny_cfg("bf16")
    m = build(cfg)
    m.train()
    opt = FlatAdamW(m, lr=1e-3)
    st = m.store
    before = {g: t.clone() for g, t in st.param.items()}
    ep = SyntheticEpisodes(cfg, 2, seed=3, instr_len=30, n_views=8, device=torch.device(DEV))
    nav_step(m, CrossEntropyLoss(), ep, train=True, last=True)
    opt.clip_grad_norm_(40.0)
    opt.step()
    opt.zero_grad()
    torch.cuda.synchronize()

    def changed(n):
        o, k = st.offsets[n], st.sizes[n]
        return not torch.equal(st.param[st.group_of[n]][o:o + k], before[st.group_of[n]][o:o + k])

    assert changed("out_head.0.weight") and changed("lang_model.model.layers.0.mlp.up_proj.weight") and changed("img_embeddings.mapper.weight")
    for n in ("lang_model.lm_head.weight", "og_head.0.weight", "og_head.0.bias", "img_embeddings.obj_projector.0.weight",
              "obj_pos_embeddings.0.weight"):
        assert not changed(n), f"{n} moved (weight decay / update) although it never had a gradient"
    # second optimizer step after an LM-loss backward: lm_head joins with step count 1
    q = gold("g5_qa_bf16.npz")
    feats = [dev(q["features"])[i, :int(n)] for i, n in enumerate(q["feat_lens"])]
    m.eval()
    out = m("3dqa", dict(features=feats, question=["q"] * 3, input_ids=T(q["input_ids"]), attention_mask=T(q["attention_mask"]),
                         token_type_ids=T(q["token_type_ids"])), training=True)
    out.loss.backward()
    torch.cuda.synchronize()
    n = "lang_model.lm_head.weight"
    p0, g0 = st.p(n).detach().clone().cpu(), st.g(n).detach().clone().cpu()
    opt.step()                                         # no clip this time
    torch.cuda.synchronize()
    assert opt.born[n] == 1 and opt.step_count == 2
    want = p0.clone()
    O.adamw_step_(want, g0, torch.zeros_like(p0), torch.zeros_like(p0), 1, lr=1e-3)
    got = st.p(n).detach().cpu()
    assert torch.equal(got.float(), want.float()), (got.float() - want.float()).abs().max()


# ------------------------------------------------------------------------------------------------ the reference's constructor call
def _hf_dir(tmp_path, cfg, seed):
    from safetensors.torch import save_file
    from navillm_amd.params import synth_state_dict
    d = tmp_path / "tiny_vicuna"
    shutil.copytree(os.path.join(GOLD, "tiny_llama"), d)
    sd = synth_state_dict(cfg, seed)
    hf = {k[len("lang_model."):]: (v[:cfg.base_vocab_size] if k.endswith(("embed_tokens.weight", "lm_head.weight")) else v).contiguous()
          for k, v in sd.items() if k.startswith("lang_model.")}
    save_file(hf, str(d / "model.safetensors"))
    return str(d), sd


def test_reference_constructor_call_loads_pretrained_lm(tmp_path):
    """`NavModel(args, logger, model_config)` exactly as train.py:227 calls it: config + tokenizer + the PRETRAINED LM weights
    from the HF directory (it used to train a random LM silently), torch-default init for the rest; from_scratch and a
    missing checkpoint behave like the reference."""
    import logging
    from navillm_amd.nav_model import NavModel
    cfg = tiny_cfg("bf16")
    path, sd = _hf_dir(tmp_path, cfg, 21)
    args = types.SimpleNamespace(precision="amp_bf16", pretrained_model_name_or_path=path, image_feat_size=cfg.image_feat_size,
                                 angle_feat_size=4, obj_feat_size=cfg.obj_feat_size, resume_from_checkpoint=None, from_scratch=False,
                                 enable_og=True, fuse_obj=False, feat_dropout=0.4)
    mc = types.SimpleNamespace(num_pano_layers=2)
    torch.cuda.set_device(0)
    m = NavModel(args, logging.getLogger("t"), mc)
    for n in ("lang_model.model.layers.1.mlp.down_proj.weight", "lang_model.model.norm.weight"):
        assert torch.equal(m.P(n).detach().cpu(), sd[n])
    for n in ("lang_model.model.embed_tokens.weight", "lang_model.lm_head.weight"):
        w = m.P(n).detach().cpu()
        assert torch.equal(w[:cfg.base_vocab_size], sd[n][:cfg.base_vocab_size])
        extra = w[cfg.base_vocab_size:].float()
        assert extra.shape[0] == 6 and 0.005 < float(extra.std()) < 0.04        # resize_token_embeddings: N(0, initializer_range)
    assert torch.equal(m.P("img_embeddings.img_layer_norm.weight").detach().cpu(), torch.ones(m.cfg.enc_hidden_size))
    assert float(m.P("img_embeddings.pano_encoder.layers.0.self_attn.in_proj_bias").abs().max()) == 0.0
    # the attached tokenizer reproduces the reference's ids, and the string-prompt path runs end to end
    z = gold("g3_nav_bf16.npz")
    m.eval()
    with torch.no_grad():
        pano = m("panorama", pano_batch(z))
        from util import nav_batch_from_gold
        batch, meta = nav_batch_from_gold(z, pano["pano_embeds"])
        batch["hist_vis"] = [[v.to(DEV) for v in vis] for vis in batch["hist_vis"]]
        out = m("navigation", batch)
    lg = out["fuse_logits"].float().cpu()
    assert torch.equal(torch.isfinite(lg), torch.isfinite(T(z["fuse_logits"])))
    tok = m.lang_model.tokenize(meta["prompts"])
    assert torch.equal(tok["input_ids"], T(z["input_ids"]))
    # from_scratch: from-config LM (normal(0, 0.02)), no checkpoint read
    args.from_scratch = True
    ms = NavModel(args, logging.getLogger("t"), mc)
    w = ms.P("lang_model.model.layers.0.self_attn.q_proj.weight").float()
    assert abs(float(w.std()) - 0.02) < 3e-3 and not torch.equal(w.cpu(), sd["lang_model.model.layers.0.self_attn.q_proj.weight"].float())
    assert torch.equal(ms.P("lang_model.model.norm.weight").detach().cpu().float(), torch.ones(cfg.hidden_size))
    # no weights in the directory: loud failure, never synthetic weights
    os.remove(os.path.join(path, "model.safetensors"))
    args.from_scratch = False
    with pytest.raises(FileNotFoundError):
        NavModel(args, logging.getLogger("t"), mc)


# ------------------------------------------------------------------------------------------------ mixed tasks (config 3)
def test_mixed_task_meta_steps_run_every_backward_of_the_rollout():
    """BASELINE config 3 as training steps (VERDICT r1 'missing' 5): for each task of the mix one episode through the synthetic
    rollout -- per-step navigation backward, fine-grained-R2R LM loss on non-last steps, object grounding and summarization with
    their own backward on the last step -- then clip + AdamW; and a ScanQA batch.  Checks which parameters each task gives a
    gradient (the reference's mode-dependent parameter usage), finiteness, and that an optimizer step follows."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.optim import FlatAdamW
    from navillm_amd.synthetic import SyntheticEpisodes, mixed_task_episode, qa_step
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=9)
    m.train()
    opt = FlatAdamW(m, lr=1e-4)
    crit = CrossEntropyLoss()
    st = m.store

    def nz(name):
        return float(st.g(name).float().abs().max()) > 0

    for i, task in enumerate(("r2r", "reverie", "soon", "cvdn")):
        ep = SyntheticEpisodes(cfg, 3, seed=50 + i, instr_len=120, device=torch.device(DEV), task=task)
        m.zero_grad()
        st.touched.clear()
        losses = mixed_task_episode(m, crit, ep, steps=3)
        torch.cuda.synchronize()
        flat = [l for l in losses["nav"] + losses["fgr2r"] + [losses["og"], losses["sum"]] if l is not None]
        assert all(np.isfinite(float(l.detach())) for l in flat), (task, losses)
        assert (losses["og"] is not None) == (task in ("reverie", "soon"))
        assert (losses["sum"] is not None) == (task != "cvdn")
        assert (len(losses["fgr2r"]) > 0) == (task == "r2r")
        has_lm_loss = task != "cvdn"
        assert ("lang_model.lm_head.weight" in st.touched) == has_lm_loss and nz("lang_model.lm_head.weight") == has_lm_loss
        og = task in ("reverie", "soon")
        assert ("img_embeddings.obj_projector.0.weight" in st.touched) == og and nz("obj_pos_embeddings.0.weight") == og
        assert nz("out_head.0.weight") and nz("img_embeddings.img_linear.weight") and nz("lang_model.model.layers.0.mlp.up_proj.weight")
        assert not nz("og_head.0.weight")
        for g in st.grad.values():
            assert torch.isfinite(g.float()).all()
        before = st.p("lang_model.model.layers.1.self_attn.o_proj.weight").clone()
        opt.clip_grad_norm_(40.0)
        opt.step()
        torch.cuda.synchronize()
        assert not torch.equal(before, st.p("lang_model.model.layers.1.self_attn.o_proj.weight"))
    # ScanQA batch (llava.py:19-42), single-<cand> prompt
    m.zero_grad()
    st.touched.clear()
    loss, _ = qa_step(m, SyntheticEpisodes(cfg, 3, seed=60, instr_len=20, device=torch.device(DEV)))
    torch.cuda.synchronize()
    assert np.isfinite(float(loss.detach())) and nz("lang_model.lm_head.weight") and nz("img_embeddings.mapper.weight")
    assert "out_head.0.weight" not in st.touched and not nz("out_head.0.weight")


# ------------------------------------------------------------------------------------------------ in-kernel dropout
def test_philox_dropout_kernel_statistics_and_backward_mask():
    """nv_dropout_f32: keep rate, scaling, reproducibility from (seed, offset), independent streams for different offsets, and the
    backward regenerating exactly the forward's mask (dropout masks came from torch.rand + torch elementwise ops)."""
    from navillm_amd import ops, functions as Fn
    n = 1 << 20
    x = torch.ones(n, device=DEV)
    for p in (0.1, 0.4):
        y = ops.dropout_f32(x, p, seed=1234, offset=0)
        keep = (y != 0)
        rate = keep.float().mean().item()
        assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4, (p, rate)
        assert torch.allclose(y[keep], torch.full_like(y[keep], 1 / (1 - p)))
        assert torch.equal(y, ops.dropout_f32(x, p, 1234, 0))                          # same key -> same mask
        y2, y3 = ops.dropout_f32(x, p, 1234, n // 4), ops.dropout_f32(x, p, 99, 0)
        for other in (y2, y3):                                                          # other counters / other seed: independent
            agree = ((other != 0) == keep).float().mean().item()
            assert abs(agree - (p * p + (1 - p) * (1 - p))) < 5e-3, agree
        # no structure along the 4-element counter groups or in blocks
        k4 = keep.view(-1, 4).float().mean(0)
        assert (k4 - (1 - p)).abs().max().item() < 5e-3
        assert (keep.view(256, -1).float().mean(1) - (1 - p)).abs().max().item() < 3.5e-2     # 4096 draws per block: sigma 0.008
    # autograd: d(out)/dx is the same mask, regenerated in the backward
    xg = torch.randn(3, 36, 768, device=DEV, requires_grad=True)
    torch.manual_seed(7)
    out = Fn.dropout(xg, 0.4, True)
    out.backward(torch.ones_like(out))
    assert torch.equal(xg.grad != 0, out != 0) and torch.allclose(xg.grad[out != 0], torch.full_like(xg.grad[out != 0], 1 / 0.6))
    o2 = Fn.dropout(xg, 0.4, True)
    assert not torch.equal(o2 != 0, out != 0)                                           # the stream advances from call to call
    torch.manual_seed(7)
    assert torch.equal(Fn.dropout(xg, 0.4, True) != 0, out != 0)                        # torch.manual_seed re-keys it
    assert Fn.dropout(xg, 0.4, False) is xg and n % 4 == 0
    t = torch.ones(1001, device=DEV)                                                    # n % 4 != 0 tail
    assert ops.dropout_f32(t, 0.5, 1, 0).shape == t.shape


def test_fast_maps_equal_the_per_node_maps():
    """SyntheticEpisodes(fast_maps=True) -- node embeddings in one device matrix per batch (index_copy / index_add / gather) and
    visited flags in host sets -- against the per-node form of mp3d_agent.py:304-371 / graph_utils.py:119-142 (one device tensor
    per node, one side-car call per visited() query): bit-identical map embeddings, masks, step ids and pose features at every
    step of a 12-step rollout with a saturating frontier, and the same walk"""
    from navillm_amd import config as nvcfg
    from navillm_amd.nav_model import NavModel
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.synthetic import SyntheticEpisodes
    cfg = nvcfg.NavConfig(hidden_size=256, num_layers=1, num_heads=2, intermediate_size=512, base_vocab_size=500, enc_hidden_size=128,
                          enc_num_heads=4, enc_intermediate_size=256, image_feat_size=768)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=2)
    m.eval()
    runs = {}
    for fast in (False, True):
        ep = SyntheticEpisodes(cfg, 4, seed=77, instr_len=30, device=torch.device(DEV), max_frontier=6, fast_maps=fast)
        torch.manual_seed(5)
        rec = []
        with torch.no_grad():
            for t in range(12):
                pin = ep.panorama_inputs()
                pano = m("panorama", pin)
                ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
                nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
                ids, am = ep.tokenise(nav, m.lang_model.cls_token[0])
                nav["input_ids"], nav["attention_mask"] = ids, am
                out = m("navigation", nav)
                rec.append({k: nav[k].clone().cpu() for k in ("gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_visited_masks", "gmap_masks")})
                rec[-1]["vpids"] = [list(v) for v in nav["gmap_vpids"]]
                rec[-1]["logits"] = out["fuse_logits"].float().cpu()
                ep.teacher_targets(nav, False)
                ep.advance(nav, out["fuse_logits"].float().argmax(1).cpu(), out["fuse_embeds"])
        runs[fast] = (rec, list(ep.cur))
    assert runs[False][1] == runs[True][1]
    for a, b in zip(runs[False][0], runs[True][0]):
        assert a["vpids"] == b["vpids"]
        for k in ("gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_visited_masks", "gmap_masks", "logits"):
            assert torch.equal(a[k], b[k]), k
    assert runs[True][0][-1]["gmap_img_embeds"].shape[1] > 12          # the maps did grow
