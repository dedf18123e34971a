"""GPU: the HIP product path (navillm_amd.NavModel, every op through the C ABI) against
  (1) the golden vectors produced by the reference itself (tests/golden/*.npz), and
  (2) the CPU oracle on the same seeded inputs at a larger, multi-tile size.

Tolerances (stated per north_star): fp32 stages (scene encoder, fusion) 1e-3 relative to the
tensor scale -- measured ~1e-6; bf16 LM outputs are compared bf16-vs-bf16 with the same rounding
points: the allowed gap is a few bf16 ulps of the value range, and additionally the HIP result
must be as close to the reference's FP32 result as the reference's own bf16 run is (x1.5);
action/object selection must be argmax-exact wherever the reference's top-2 margin exceeds the
measured gap.
"""
import json
import numpy as np
import pytest
import torch

from util import (gold, T, tiny_cfg, meta_of, hist_lists, nav_batch_from_gold, load_oracle, GOLDEN_SEED, bf16_ulps_at_scale,
                  episode_step_batch, grad_fixture_errors)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# End-to-end bf16 logits vs the reference's (or the bf16 oracle's) bf16 logits, in units of the bf16 spacing at the logits'
# own magnitude.  Measured on MI355X (round 2): 0.8-1.5 at every size from the tiny fixtures to the 7B-shaped layer, i.e. the
# two runs differ in the LAST BIT of the bf16 output -- not by a rounding-point deviation inside the model: switching the
# attention kernel to HF's rounding points does not move it (tests/test_round2_gpu.py).  Asserted: measured x 1.5, rounded up.
ULPS_LOGITS = 2.5
ULPS_FRAME = 1.5      # what evaluating RoPE in the sample frame instead of the batch frame may add (measured 1.0 on G12: see the G12 episode test)


def build(cfg, seed=GOLDEN_SEED):
    from navillm_amd.nav_model import NavModel
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=seed)
    m.eval()
    return m


def relerr(a, b):
    a, b = a.float().cpu(), torch.as_tensor(b).float()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def maxerr(a, b):
    a, b = a.float().cpu(), torch.as_tensor(b).float()
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), "inf pattern differs"
    return (a[fin] - b[fin]).abs().max().item()


def dev(x):
    return T(x).to(DEV)


def pano_batch(z, with_obj=False):
    b = dict(view_img_fts=dev(z["view_img_fts"]), view_lens=dev(z["view_lens"]), loc_fts=dev(z["loc_fts"]),
             nav_types=dev(z["nav_types"]))
    if with_obj:
        b.update(obj_img_fts=dev(z["obj_img_fts"]), obj_lens=dev(z["obj_lens"]), obj_loc_fts=dev(z["obj_loc_fts"]))
    return b


def test_g1_scene_encoder_vs_reference():
    z = gold("g1_encoder.npz")
    m = build(tiny_cfg("bf16"))
    with torch.no_grad():
        out = m("panorama", pano_batch(z, True))
        nop = m.forward_panorama_per_step(dev(z["view_img_fts"]), dev(z["view_lens"]))
    scale = np.abs(z["pano_embeds"]).max()
    assert maxerr(out["pano_embeds"], z["pano_embeds"]) < 1e-3 * scale
    assert maxerr(out["pano_embeds"], z["pano_embeds"]) < 2e-5, "fp32 MFMA path should be ~1e-6"
    assert np.array_equal(out["pano_masks"].cpu().numpy(), z["pano_masks"])
    assert maxerr(out["obj_embeds"], z["obj_embeds"]) < 2e-5
    assert np.array_equal(out["obj_masks"].cpu().numpy(), z["obj_masks"])
    assert maxerr(nop["pano_embeds"], z["nopose_pano_embeds"]) < 2e-5


def test_g1_scene_encoder_fuse_obj_vs_reference():
    z = gold("g1_encoder_fuseobj.npz")
    m = build(tiny_cfg("bf16", fuse_obj=True))
    with torch.no_grad():
        out = m("panorama", pano_batch(z, True))
    assert maxerr(out["pano_embeds"], z["pano_embeds"]) < 2e-5
    assert maxerr(out["obj_embeds"], z["obj_embeds"]) < 2e-5


def test_g2_visual_token_lm_vs_reference():
    zb, zf = gold("g2_lm_bf16.npz"), gold("g2_lm_fp32.npz")
    m = build(tiny_cfg("bf16"))
    ids, am = T(zb["input_ids"]), T(zb["attention_mask"])
    with torch.no_grad():
        Hs = m._lm(ids, am, cand_vis=dev(zb["cand_vis"]), hist_vis=dev(zb["hist_vis"]))
        loss = m._lm_loss(Hs, ids, T(zb["labels"]))
    real = am.bool().view(-1)
    B, S = ids.shape
    got = Hs.float().cpu() if m._row_map is not None else Hs.float().cpu()[real]      # packed rows: the real tokens only
    assert got.shape[0] == int(real.sum())
    ref16 = T(zb["hidden_states"]).view(B * S, -1)[real]
    ref32 = T(zf["hidden_states"]).view(B * S, -1)[real]
    e_hip = (got - ref32).abs().max().item()
    e_ref = (ref16 - ref32).abs().max().item()
    gap = (got - ref16).abs().max().item()
    print(f"[g2] max|hip-ref_bf16|={gap:.4f}  max|hip-ref_fp32|={e_hip:.4f}  max|ref_bf16-ref_fp32|={e_ref:.4f}")
    assert gap <= 4 * 2 ** -8 * ref16.abs().max().item() + 1e-3      # a few bf16 ulps of the value range
    assert e_hip <= 1.5 * e_ref + 1e-3
    # the reference returns the loss as a bf16 scalar: 1 ulp at ~5.5 is 0.03125
    assert abs(float(loss) - float(zb["loss"])) <= 0.04 and abs(float(loss) - float(zf["loss"])) <= 0.06


def _nav_forward(m, z):
    pano = m("panorama", pano_batch(z))
    batch, meta = nav_batch_from_gold(z, pano["pano_embeds"])
    for k in ("gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_visited_masks", "gmap_masks", "pano_masks", "vp_pos_fts"):
        batch[k] = batch[k].to(DEV)
    batch["hist_vis"] = [[v.to(DEV) for v in vis] for vis in batch["hist_vis"]]
    batch["input_ids"], batch["attention_mask"] = T(z["input_ids"]), T(z["attention_mask"])
    torch.manual_seed(meta["seed_before_nav"])
    out = m("navigation", batch)
    return pano, out, meta


def test_g3_g4_navigation_loss_grads_vs_reference():
    from navillm_amd.losses import CrossEntropyLoss
    zb, zf = gold("g3_nav_bf16.npz"), gold("g3_nav_fp32.npz")
    m = build(tiny_cfg("bf16"))
    m.zero_grad()
    pano, out, meta = _nav_forward(m, zb)
    assert maxerr(pano["pano_embeds"], zb["pano_embeds"]) < 2e-5
    assert maxerr(out["fuse_embeds"], zb["fuse_embeds"]) < 2e-5
    lg, l16, l32 = out["fuse_logits"], T(zb["fuse_logits"]), T(zf["fuse_logits"])
    gap, e_hip, e_ref = maxerr(lg, l16), maxerr(lg, l32), maxerr(l16, l32)
    ulps = bf16_ulps_at_scale(lg, l16)
    print(f"[g3] logits max|hip-ref_bf16|={gap:.5f} = {ulps:.2f} bf16 ulps of the logit scale; |hip-ref_fp32|={e_hip:.5f} "
          f"|ref_bf16-ref_fp32|={e_ref:.5f}")
    # the logits are bf16 numbers: the distance to the reference's bf16 run is counted in their own last bits (measured 0.8)
    assert ulps <= ULPS_LOGITS and e_hip <= 1.5 * e_ref + 2e-3
    # argmax-exact where the reference margin exceeds the gap
    top2 = torch.topk(l16.masked_fill(~torch.isfinite(l16), -1e9), 2, dim=1).values
    for b in range(l16.shape[0]):
        if (top2[b, 0] - top2[b, 1]).item() > 2 * gap:
            assert int(lg[b].float().argmax()) == int(l16[b].argmax())
    targets = torch.tensor(meta["targets"], device=DEV)
    loss = CrossEntropyLoss()(lg, targets) * 1.0 / len(meta["targets"]) / 1
    assert abs(float(loss.detach()) - float(zb["loss"])) < 1e-2
    loss.backward()
    torch.cuda.synchronize()
    worst = {}
    for k in zb:
        if not k.startswith("grad/"):
            continue
        n = k[5:]
        g = m.store.g(n)
        r16, r32, base = relerr(g, zb[k]), relerr(g, zf[k]), relerr(T(zb[k]), zf[k])
        worst[n] = (r16, r32)
        # bf16-vs-bf16 (same rounding points): measured <= 0.014 relative on MI355X (x1.5); and as close to the fp32
        # reference as the reference's own bf16 run is (its relative error `base`), x1.5
        assert r16 < 2.1e-2 and r32 < 1.5 * base + 2e-2, (n, r16, r32, base)
    print("[g4] grad rel errs (vs ref bf16, vs ref fp32):", {k: (round(a, 4), round(b, 4)) for k, (a, b) in worst.items()})
    # embedding-table gradient row norms
    gn = m.store.g("lang_model.model.embed_tokens.weight").float().norm(dim=1).cpu()
    assert relerr(gn, zb["gradnorm/lang_model.model.embed_tokens.weight"]) < 5e-2
    # parameters the reference leaves without gradient stay exactly zero here (og_head, obj_*)
    with_grad = set(str(s) for s in zb["grad_names_with_grad"])
    for n in m.store.offsets:
        if n not in with_grad:
            assert float(m.store.g(n).float().abs().max()) == 0.0, n


def _g12_prefix_ids(z, meta):
    """each sample's static prompt prefix (everything up to "### History:", the same at every step) as token ids, cut out of
    the reference tokenizer's left-padded `input_ids` at the length the fixture recorded"""
    ids, am = T(z["s0/input_ids"]), T(z["s0/attention_mask"])
    out = []
    for b, n in enumerate(meta["steps"][0]["prefix_lens"]):
        real = ids[b][am[b].bool()].tolist()
        out.append(real[:n])
    for t, ms in enumerate(meta["steps"]):            # the same prefix at every step (what makes the reuse legal)
        ids_t, am_t = T(z[f"s{t}/input_ids"]), T(z[f"s{t}/attention_mask"])
        for b, n in enumerate(ms["prefix_lens"]):
            assert ids_t[b][am_t[b].bool()].tolist()[:n] == out[b]
    return out


@pytest.mark.parametrize("mode", ["recompute", "prefix_reuse", "auto"])
def test_g12_episode_accumulated_gradients_vs_reference(mode):
    """VERDICT r2 missing #3: a reference-pinned MULTI-STEP episode.  Three nav steps the way the rollout runs them
    (mp3d_agent.py:659-778): panorama -> navigation with the history THIS model's previous steps produced -> CE * train_ml / B /
    accum -> backward() at once, gradients accumulating; compared with the reference's own run of the episode (G12): logits and
    loss per step, the history rows, the accumulated gradients.  `recompute` = the default path (whole prompt every step, like
    the reference); `prefix_reuse` = navillm_amd/episode.py (static prefix forward once, suffix rows per step, one deferred
    prefix backward) -- the same tolerances for both."""
    from navillm_amd.losses import CrossEntropyLoss
    zb, zf = gold("g12_episode_bf16.npz"), gold("g12_episode_fp32.npz")
    meta = meta_of(zb)
    B = meta["B"]
    if mode == "auto":
        # round 6: the AUTOMATIC episode -- no begin_episode / finish_episode: the model opens the episode on the first training-mode
        # navigation call (prefix = the lengths the batch carries) and `model.parameters()` (train.py:87's clip) closes it.  Automatic
        # episodes exist in training mode only; the fixture is the reference in eval(), so the dropout rates are zero instead
        m = build(tiny_cfg("bf16", feat_dropout=0.0, enc_dropout=0.0))
        m.train()
        m.auto_episode = True
    else:
        m = build(tiny_cfg("bf16"))
    m.zero_grad()
    m.store.touched.clear()
    crit = CrossEntropyLoss()
    if mode == "prefix_reuse":
        m.begin_episode(_g12_prefix_ids(zb, meta))
    hist = [[] for _ in range(B)]
    worst_ulps = 0.0
    for t in range(len(meta["steps"])):
        pre = f"s{t}/"
        zt = {k[len(pre):]: v for k, v in zb.items() if k.startswith(pre)}
        pano = m("panorama", pano_batch(zt))
        assert maxerr(pano["pano_embeds"], zt["pano_embeds"]) < 2e-5
        batch, ms = episode_step_batch(zb, meta, t, pano["pano_embeds"], hist)
        for k in ("gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_visited_masks", "gmap_masks", "pano_masks", "vp_pos_fts"):
            batch[k] = batch[k].to(DEV)
        batch["input_ids"], batch["attention_mask"] = T(zt["input_ids"]), T(zt["attention_mask"])
        if mode == "auto":
            batch["prefix_lens"] = list(ms["prefix_lens"])
        torch.manual_seed(ms["seed_before_nav"])
        out = m("navigation", batch)
        if mode == "auto":
            assert m._auto_open and m.auto_stats["opened"] == 1 and m.episode.prefix is not None
        assert maxerr(out["fuse_embeds"], zt["fuse_embeds"]) < 3e-5
        lg, l16, l32 = out["fuse_logits"], T(zt["fuse_logits"]), T(zf[pre + "fuse_logits"])
        if mode == "auto":
            from navillm_amd.losses import LazyLogits
            assert isinstance(lg, LazyLogits)
            lg = lg.force()                          # this test LOOKS at every step's logits: the handle runs the step and hands out a live tensor
        ulps = bf16_ulps_at_scale(lg, l16)
        worst_ulps = max(worst_ulps, ulps)
        gap, e_hip, e_ref = maxerr(lg, l16), maxerr(lg, l32), maxerr(l16, l32)
        print(f"[g12 {mode} step {t}] logits |hip-ref_bf16|={gap:.5f} = {ulps:.2f} bf16 ulps; |hip-ref_fp32|={e_hip:.5f} |ref_bf16-ref_fp32|={e_ref:.5f}")
        # prefix_reuse evaluates RoPE in the SAMPLE frame (positions from each sample's first token; the reference: arange(S) over the
        # batch's left padding).  That alone moves the recompute path from 2.0 to 3.0 spacings on this fixture, and the prefix-reuse
        # logits are then BIT-IDENTICAL to the recompute path run in that frame (tests/test_parity_r4_gpu.py::
        # test_rope_frame_isolated_on_the_reference_episode_g12, profiles/r04_parity_rope_frame.txt) -- so its bound is the batch-frame
        # bound + the measured frame effect (1.0) with the same x1.5 margin
        assert ulps <= ULPS_LOGITS + (ULPS_FRAME if mode != "recompute" else 0.0) and e_hip <= 1.5 * e_ref + 4e-3
        top2 = torch.topk(l16.masked_fill(~torch.isfinite(l16), -1e9), 2, dim=1).values
        for b in range(B):
            if (top2[b, 0] - top2[b, 1]).item() > 2 * gap:
                assert int(lg[b].float().argmax()) == int(l16[b].argmax())
        tg = torch.tensor(ms["targets"], device=DEV)
        loss = crit(lg, tg) * meta["train_ml"] / B / meta["accum"]
        assert abs(float(loss.detach()) - float(zt["loss"])) < 1e-2
        loss.backward()
        for b in range(B):
            if ms["targets"][b] != -100:
                hist[b].append(out["fuse_embeds"][b][ms["targets"][b]].detach())
    if mode == "prefix_reuse":
        stats = dict(m.episode.stats)
        m.finish_episode()
        assert m.episode.prefix is None
        print(f"[g12 prefix_reuse] token rows: prefix {stats['prefix_rows']} once + suffixes {stats['suffix_rows']}")
    if mode == "auto":
        assert m.episode.has_pending_gradients()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1e9)          # train.py:87 (a bound that never clips): hands the episode over
        assert not m._auto_open and m.episode.prefix is None and m.auto_stats["closed_by"] == {"parameters": 1}
    torch.cuda.synchronize()
    assert [len(h) for h in hist] == meta["hist_final"]
    assert maxerr(torch.stack([v for h in hist for v in h], 0), zb["hist_final_flat"]) < 3e-5
    e16 = grad_fixture_errors(zb, "acc", m.store.g)
    e32 = grad_fixture_errors(zf, "acc", m.store.g)
    base = grad_fixture_errors(zf, "acc", lambda n: _acc_grad(zb, n, m.store.g(n)))
    print(f"[g12 {mode}] accumulated-gradient rel errs vs the reference's bf16 run:", {k: round(v, 4) for k, v in e16.items()})
    for k in e16:
        # the G4 / G10 single-step tolerances: bf16-vs-bf16 2.1 %, and as close to the reference's fp32 gradients as the reference's own
        # bf16 gradients are (x1.5 + 2 %)
        assert e16[k] < (2.1e-2 if not k.startswith("rownorm/") else 5e-2), (mode, k, e16[k])
        assert e32[k] < 1.5 * base[k] + 2e-2, (mode, k, e32[k], base[k])
    with_grad = set(str(s_) for s_ in zb["acc/grad_names_with_grad"])
    for n in m.store.offsets:
        if n not in with_grad:
            assert float(m.store.g(n).float().abs().max()) == 0.0, n


def _acc_grad(z, name, like):
    """the reference's bf16-run gradient `name` in the shape grad_fixture_errors expects (whole tensor rebuilt from what the
    fixture stores: `grad/` whole, `gradsub/` the [::3, ::5] sub-block scattered into zeros, `rownorm/` -> a one-column matrix)"""
    if "acc/grad/" + name in z:
        return T(z["acc/grad/" + name])
    if "acc/gradsub/" + name in z:
        full = torch.zeros(like.shape)
        full[::3, ::5] = T(z["acc/gradsub/" + name])
        return full
    rn = T(z["acc/rownorm/" + name])
    full = torch.zeros(like.shape)
    full[:, 0] = rn
    return full


def test_g5_object_grounding_and_qa_vs_reference():
    z = gold("g5_og_bf16.npz")
    m = build(tiny_cfg("bf16"))
    meta = meta_of(z)
    with torch.no_grad():
        po = m("panorama", pano_batch(z, True))
        assert maxerr(po["obj_embeds"], z["obj_embeds"]) < 2e-5
        hv = hist_lists(dev(z["hist_vis_flat"]), meta["hist_t"])
        b = dict(obj_embeds=po["obj_embeds"], obj_masks=po["obj_masks"], obj_loc_fts=po["obj_loc_fts"], hist_vis=hv,
                 input_ids=T(z["input_ids"]), attention_mask=T(z["attention_mask"]), prompts=meta["prompts"])
        oo = m("object_grounding", b)
        m.enable_kv_cache(len(hv))                      # the same call through the K/V-cache path (inference rollouts)
        oc = m("object_grounding", b)
        oc2 = m("object_grounding", b)                  # second call: everything up to the object tokens is reused
        assert max(m.kv.last_stats["prefix"]) > 0
        m.kv = None
    u = [bf16_ulps_at_scale(o["obj_logits"], z["obj_logits"]) for o in (oo, oc, oc2)]
    print(f"[g5 og] obj_logits vs ref bf16, in bf16 ulps of the logit scale: plain {u[0]:.2f}, K/V cache {u[1]:.2f}, cached again {u[2]:.2f}")
    assert max(u) <= ULPS_LOGITS
    q = gold("g5_qa_bf16.npz")
    feats = [dev(q["features"])[i, :int(n)] for i, n in enumerate(q["feat_lens"])]
    with torch.no_grad():
        out = m("3dqa", dict(features=feats, question=["q"] * len(feats), input_ids=T(q["input_ids"]),
                             attention_mask=T(q["attention_mask"]), token_type_ids=T(q["token_type_ids"])), training=True)
    assert abs(float(out.loss) - float(q["loss"])) <= 0.04


def test_g5_summarization_and_fgr2r_losses_vs_reference():
    z = gold("g5_sum_bf16.npz")
    m = build(tiny_cfg("bf16"))
    meta = meta_of(z)
    B = len(meta["hist_t"])
    with torch.no_grad():
        ps = m("panorama", pano_batch(z))
        vp = torch.cat([torch.zeros_like(ps["pano_embeds"][:, :1]), ps["pano_embeds"]], 1)
        hv = hist_lists(dev(z["hist_vis_flat"]), meta["hist_t"])
        common = dict(vp_img_embeds=vp, vp_nav_masks=T(z["vp_nav_masks"]), instruction=["x"] * B, answer=["y"] * B)
        o1 = m("summarization", dict(common, hist_vis=hv, data_type=["r2r"] * B, input_ids=T(z["sum_input_ids"]),
                                     attention_mask=T(z["sum_attention_mask"]), token_type_ids=T(z["sum_token_type_ids"])),
               training=True)
        o2 = m("embodied_qa", dict(common, hist_vis=[[] for _ in range(B)], data_type=["fgr2r"] * B,
                                   input_ids=T(z["qa_input_ids"]), attention_mask=T(z["qa_attention_mask"]),
                                   token_type_ids=T(z["qa_token_type_ids"])), training=True)
    assert abs(float(o1["loss"]) - float(z["sum_loss"])) <= 0.04      # bf16 scalar: 1 ulp at ~5.5 = 0.03125
    assert abs(float(o2["loss"]) - float(z["qa_loss"])) <= 0.04


def test_g9_generation_vs_reference_generate():
    """model('3dqa' | 'summarization', batch, training=False[, trie=...]) against the token ids the reference's own
    generate() calls returned (fixture G9; bf16): K/V-cache greedy decoding, special-id mask, eos bookkeeping, trie."""
    from test_oracle_golden import g9_inputs
    z = gold("g9_generate_bf16.npz")
    m = build(tiny_cfg("bf16"))
    import types
    meta, feats_cpu, trie = g9_inputs(z)
    m.lang_model.tokenizer = types.SimpleNamespace(eos_token_id=meta["eos"], unk_token_id=meta["pad"])   # ids only: prompts are pre-tokenised
    feats = [f.to(DEV) for f in feats_cpu]
    B = len(feats)
    with torch.no_grad():
        qa = m("3dqa", dict(features=feats, question=["q"] * B, input_ids=T(z["qa_input_ids"]), attention_mask=T(z["qa_attention_mask"])),
               training=False, do_sample=False, max_new_tokens=6)
        pz = {k[4:]: v for k, v in z.items() if k.startswith("sum_")}
        ps = m("panorama", pano_batch(pz))
        vp = torch.cat([torch.zeros_like(ps["pano_embeds"][:, :1]), ps["pano_embeds"]], 1)
        hv = hist_lists(dev(z["sum_hist_vis_flat"]), meta["hist_t"])
        sm = m("summarization", dict(vp_img_embeds=vp, vp_nav_masks=T(z["sum_vp_nav_masks"]), instruction=["x"] * B, answer=[""] * B,
                                     hist_vis=hv, data_type=["r2r"] * B, input_ids=T(z["sum_input_ids"]),
                                     attention_mask=T(z["sum_attention_mask"])), training=False, trie=trie)
    assert qa["generated_ids"] == z["qa_new_ids"].tolist(), (qa["generated_ids"], z["qa_new_ids"].tolist())
    assert sm["generated_ids"] == z["sum_new_ids"].tolist(), (sm["generated_ids"], z["sum_new_ids"].tolist())


def test_tokenizer_path_matches_fixture_ids():
    """drop-in tokenisation: the attached LlamaTokenizer reproduces the reference's ids."""
    import os
    from navillm_amd.nav_model import load_tokenizer
    from util import GOLD
    z = gold("g3_nav_bf16.npz")
    cfg = tiny_cfg("bf16")
    tok = load_tokenizer(os.path.join(GOLD, "tiny_llama"), cfg)
    from navillm_amd.nav_model import LangModelShell
    sh = LangModelShell(cfg, tok)
    t = sh.tokenize(meta_of(z)["prompts"])
    assert torch.equal(t["input_ids"], T(z["input_ids"])) and torch.equal(t["attention_mask"], T(z["attention_mask"]))


class _LazyF32(dict):
    """fp32 view of a bf16 weight dict, one tensor at a time (the oracle only indexes `P[name]`): a second, widened copy of
    Vicuna-7B would be 27 GB of host memory for nothing"""

    def __init__(self, P16):
        super().__init__()
        self._p = P16

    def __getitem__(self, k):
        return self._p[k].float()

    def __contains__(self, k):
        return k in self._p


def _nav_vs_oracle(cfg, B, S_instr, steps, tag, expect_S=None, lazy32=False, argmax_exact=False):
    """HIP vs the CPU oracle run in bf16 and in fp32 on the same seeded weights and inputs, through the synthetic
    episode driver (panorama -> map update -> navigation per step)."""
    from navillm_amd import config as nvcfg
    from navillm_amd.nav_model import NavModel
    from navillm_amd.params import synth_state_dict
    from navillm_amd.synthetic import SyntheticEpisodes
    O = load_oracle()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=5)
    m.eval()
    P16 = synth_state_dict(cfg, 5)
    with torch.no_grad():
        assert m.load_reference_state_dict(P16) == len(P16)      # (large models draw their own weights on the device)
    cfg32 = nvcfg.NavConfig(**{**cfg.__dict__, "precision": "fp32"})
    P32 = _LazyF32(P16) if lazy32 else {k: v.float() for k, v in P16.items()}
    ep = SyntheticEpisodes(cfg, B, seed=77, instr_len=S_instr, device=torch.device(DEV))
    for step in range(steps):
        pin = ep.panorama_inputs()
        with torch.no_grad():
            pano = m("panorama", pin)
            ref_p = O.scene_encoder(P32, cfg32, pin["view_img_fts"].cpu(), pin["view_lens"].cpu(), pin["loc_fts"].cpu(),
                                    pin["nav_types"].cpu())
        assert maxerr(pano["pano_embeds"], ref_p["pano_embeds"]) < 1e-3 * ref_p["pano_embeds"].abs().max().item()
        ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        ids, am = ep.tokenise(nav, "<cls_1>")
        if expect_S is not None:
            assert ids.shape[1] == expect_S, ids.shape
        nav["input_ids"], nav["attention_mask"] = ids, am
        torch.manual_seed(100 + step)
        with torch.no_grad():
            out = m("navigation", nav)
        cpu = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in nav.items()}
        cpu["hist_vis"] = [[v.cpu() for v in vis] for vis in nav["hist_vis"]]
        outs = {}
        for prec, P, c in (("bf16", P16, cfg), ("fp32", P32, cfg32)):
            torch.manual_seed(100 + step)
            with torch.no_grad():
                outs[prec] = O.navigation(P, c, cpu, ids, am)
        assert maxerr(out["fuse_embeds"], outs["fp32"]["fuse_embeds"]) < 1e-4
        lg = out["fuse_logits"]
        gap, e_hip, e_ref = maxerr(lg, outs["bf16"]["fuse_logits"]), maxerr(lg, outs["fp32"]["fuse_logits"]), \
            maxerr(outs["bf16"]["fuse_logits"], outs["fp32"]["fuse_logits"])
        ref16 = outs["bf16"]["fuse_logits"]
        scale = float(ref16[torch.isfinite(ref16)].abs().max())
        ulp = 2.0 ** (int(np.floor(np.log2(scale))) - 7)                      # spacing of the bf16 logits at their own magnitude
        print(f"[{tag} step {step}] S={ids.shape[1]} logits: |hip-orc32|={e_hip:.5f} vs |orc16-orc32|={e_ref:.5f} (ratio {e_hip / e_ref:.2f}); "
              f"|hip-orc16|={gap:.5f} = {gap / ulp:.2f} bf16 spacings at the logit scale {scale:.2f}")
        # The HIP path is an independent bf16 evaluation of the same function as the bf16 oracle (= the reference's rounding
        # points): what can be asserted is that it is as close to the fp32 TRUTH as the oracle's bf16 run is.  Measured on MI355X
        # over tiny / mid / 1024-token / 7B- and 13B-shaped cases: ratio 0.58 .. 1.07 -> asserted 1.25 (+ one output spacing).
        # Its distance to the bf16 oracle itself then follows from the triangle inequality (measured 1 - 3.5 output spacings).
        assert e_hip <= 1.25 * e_ref + ulp, (e_hip, e_ref, ulp)
        if argmax_exact:
            # action selection (north_star: "argmax-exact"): wherever the bf16 oracle's top-2 margin exceeds twice the measured
            # distance between the two bf16 evaluations, both must pick the same map slot
            top2 = torch.topk(ref16.masked_fill(~torch.isfinite(ref16), -1e9), 2, dim=1).values
            decided = 0
            for b in range(ref16.shape[0]):
                if (top2[b, 0] - top2[b, 1]).item() > 2 * gap:
                    decided += 1
                    assert int(lg[b].float().argmax()) == int(ref16[b].argmax()), (b, lg[b], ref16[b])
            agree = sum(int(lg[b].float().argmax()) == int(ref16[b].argmax()) for b in range(ref16.shape[0]))
            print(f"[{tag} step {step}] argmax: {agree}/{ref16.shape[0]} rows agree with the bf16 oracle, {decided} rows have a margin > 2 x gap")
        targets = ep.teacher_targets(nav, last=False)
        ep.advance(nav, targets, out["fuse_embeds"])
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B,S_instr", [(4, 200)])
def test_midsize_navigation_vs_oracle(B, S_instr):
    """multi-tile sizes (d=512, 4 heads, ff=1408, 3 layers, 36 views, S~300)."""
    from navillm_amd import config as nvcfg
    cfg = nvcfg.NavConfig(hidden_size=512, num_layers=3, num_heads=4, intermediate_size=1408, base_vocab_size=1000,
                          enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    _nav_vs_oracle(cfg, B, S_instr, 3, "mid")


def test_long_horizon_truncation_limit_vs_oracle():
    """SURVEY.md §8d config 4 (long-horizon, S at the 1024-token left-truncation limit of modified_lm.py:77-87): the
    prompt is longer than 1024 tokens, so the tokeniser side truncates from the left and every attention tile /
    split path runs at the maximum sequence length."""
    from navillm_amd import config as nvcfg
    cfg = nvcfg.NavConfig(hidden_size=512, num_layers=2, num_heads=4, intermediate_size=1408, base_vocab_size=1000,
                          enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    _nav_vs_oracle(cfg, 2, 1000, 2, "long", expect_S=1024)


def test_full_depth_vicuna_7b_navigation_vs_oracle():
    """VERDICT r2 missing #2 / BASELINE config 2: the FULL 32-layer Vicuna-7B (d=4096, 32 heads, ff=11008, 32 064-token vocabulary),
    36 x 768-d views, 512-token instructions, B=2, one navigation step forward: the HIP logits against the CPU oracle run in
    bf16 (= the reference's rounding points) and in fp32 on the same weights (nav_model.py:232-242).  Asserted: the HIP result
    is as close to the fp32 truth as the bf16 oracle is (x1.25 + one output spacing), and picks the same action wherever the
    oracle's top-2 margin exceeds twice the gap; the distance in bf16 spacings is printed."""
    from navillm_amd import config as nvcfg
    cfg = nvcfg.vicuna_7b(image_feat_size=768)
    _nav_vs_oracle(cfg, 2, 512, 1, "7b-full-depth", lazy32=True, argmax_exact=True)


def test_13b_shaped_layer_vs_oracle():
    """SURVEY.md §8d config 5 shapes (Vicuna-13B: d=5120, 40 heads, ff=13824) on one decoder layer: tile counts that are
    not powers of two (20 / 54 / 108 column tiles), 40 heads in the attention grid."""
    from navillm_amd import config as nvcfg
    cfg = nvcfg.NavConfig(hidden_size=5120, num_layers=1, num_heads=40, intermediate_size=13824, base_vocab_size=1000,
                          enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    _nav_vs_oracle(cfg, 2, 150, 1, "13b-layer")


def test_pruned_last_layer_matches_full_path():
    """navigation reads only the <cls_1> rows: the pruned last layer (B rows through o_proj/MLP/final norm, attention
    queries of the last 128-block only) must give the same logits and the same gradients as the full computation.
    S > 256 so that q_row_min > 0 is exercised in forward AND backward."""
    from navillm_amd import config as nvcfg
    from navillm_amd.nav_model import NavModel
    from navillm_amd.synthetic import SyntheticEpisodes
    from navillm_amd.losses import CrossEntropyLoss
    cfg = nvcfg.NavConfig(hidden_size=512, num_layers=2, num_heads=4, intermediate_size=1408, base_vocab_size=1000,
                          enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=6)
    m.eval()
    crit = CrossEntropyLoss()
    res = {}
    for prune in (True, False):
        m.prune_last_layer = prune
        m.zero_grad()
        ep = SyntheticEpisodes(cfg, 3, seed=21, instr_len=260, device=torch.device(DEV))
        pin = ep.panorama_inputs()
        pano = m("panorama", pin)
        ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
        nav["input_ids"], nav["attention_mask"] = ep.tokenise(nav, "<cls_1>")
        assert nav["input_ids"].shape[1] > 300
        torch.manual_seed(3)
        out = m("navigation", nav)
        tg = ep.teacher_targets(nav, last=False)
        (crit(out["fuse_logits"], tg.to(DEV)) / 3).backward()
        torch.cuda.synchronize()
        res[prune] = (out["fuse_logits"].detach().float().cpu(), {g: t.detach().float().cpu().clone() for g, t in m.store.grad.items()})
    lp, lf = res[True][0], res[False][0]
    fin = torch.isfinite(lf)
    assert torch.equal(torch.isfinite(lp), fin) and (lp[fin] - lf[fin]).abs().max().item() < 4e-3
    for g in ("lm", "f32"):
        a, b = res[True][1][g], res[False][1][g]
        rel = ((a - b).norm() / (b.norm() + 1e-20)).item()
        assert rel < 2e-2, (g, rel)
    # per-tensor check on the tensors the pruning touches most
    st = m.store
    for n in ("lang_model.model.layers.1.mlp.down_proj.weight", "lang_model.model.layers.1.self_attn.o_proj.weight",
              "lang_model.model.layers.1.self_attn.q_proj.weight", "lang_model.model.layers.1.self_attn.k_proj.weight",
              "lang_model.model.layers.0.mlp.gate_proj.weight", "lang_model.model.embed_tokens.weight"):
        o, k = st.offsets[n], st.sizes[n]
        a, b = res[True][1]["lm"][o:o + k], res[False][1]["lm"][o:o + k]
        rel = ((a - b).norm() / (b.norm() + 1e-20)).item()
        assert rel < 3e-2, (n, rel)


def test_packed_rows_match_padded_layout():
    """the LM over the real tokens only (pack_rows, default) vs the reference's padded [B, S] layout: same navigation
    logits and the same gradients (padding rows feed nothing), with prompts of clearly different lengths, B = 1 included."""
    from navillm_amd import config as nvcfg
    from navillm_amd.nav_model import NavModel
    from navillm_amd.synthetic import SyntheticEpisodes
    from navillm_amd.losses import CrossEntropyLoss
    cfg = nvcfg.NavConfig(hidden_size=512, num_layers=2, num_heads=4, intermediate_size=1408, base_vocab_size=1000,
                          enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=8)
    m.eval()
    crit = CrossEntropyLoss()
    for B in (3, 1):
        res = {}
        for pack in (True, False):
            m.pack_rows = pack
            m.zero_grad()
            ep = SyntheticEpisodes(cfg, B, seed=41, instr_len=150, device=torch.device(DEV))
            for b in range(B):                                   # very different prompt lengths: 150, 110, 70 instruction tokens
                ep.instr[b] = ep.instr[b][: 150 - 40 * b]
            pin = ep.panorama_inputs()
            pano = m("panorama", pin)
            ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
            nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
            nav["input_ids"], nav["attention_mask"] = ep.tokenise(nav, "<cls_1>")
            if B > 1:
                assert int(nav["attention_mask"].sum()) < nav["attention_mask"].numel() - 60
            torch.manual_seed(3)
            out = m("navigation", nav)
            tg = ep.teacher_targets(nav, last=False)
            (crit(out["fuse_logits"], tg.to(DEV)) / B).backward()
            torch.cuda.synchronize()
            res[pack] = (out["fuse_logits"].detach().float().cpu(), {g: t.detach().float().cpu().clone() for g, t in m.store.grad.items()})
        lp, lf = res[True][0], res[False][0]
        fin = torch.isfinite(lf)
        assert torch.equal(torch.isfinite(lp), fin) and (lp[fin] - lf[fin]).abs().max().item() < 8e-3
        for g in ("lm", "f32"):
            a, b_ = res[True][1][g], res[False][1][g]
            rel = ((a - b_).norm() / (b_.norm() + 1e-20)).item()
            assert rel < 2e-2, (B, g, rel)
    m.pack_rows = True
