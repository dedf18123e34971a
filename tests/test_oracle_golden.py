"""CPU: the oracle (oracle/navillm_oracle.py) against the golden vectors produced by the
reference itself (tests/golden/make_golden.py). This is what pins the oracle."""
import json
import os
import numpy as np
import pytest
import torch

from util import (load_oracle, gold, T, tiny_cfg, tiny_weights, meta_of, hist_lists, nav_batch_from_gold, GOLD)

O = load_oracle()


def close(a, b, atol, rtol=0.0, what=""):
    a = torch.as_tensor(a).float()
    b = torch.as_tensor(b).float()
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), f"{what}: inf pattern differs"
    err = (a[fin] - b[fin]).abs()
    tol = atol + rtol * b[fin].abs()
    assert bool((err <= tol).all()), f"{what}: max err {err.max().item():.3e} (tol {atol}+{rtol}*|ref|)"


def test_param_inventory_matches_reference_state_dict():
    from navillm_amd.params import param_specs
    meta = json.load(open(os.path.join(GOLD, "g_meta.json")))
    for prec, tag in (("fp32", "fp32"), ("amp_bf16", "bf16")):
        cfg = tiny_cfg(tag)
        ref = meta[prec]
        mine = {n: (list(s), "torch.bfloat16" if (g == "lm" and cfg.lm_is_bf16) else "torch.float32")
                for n, s, g in param_specs(cfg)}
        assert set(mine) == set(ref)
        for k in ref:
            assert mine[k][0] == ref[k][0] and mine[k][1] == ref[k][1], k


def test_g1_scene_encoder():
    z = gold("g1_encoder.npz")
    cfg, P = tiny_weights("fp32")
    with torch.no_grad():
        out = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]), T(z["loc_fts"]), T(z["nav_types"]),
                              T(z["obj_img_fts"]), T(z["obj_lens"]), T(z["obj_loc_fts"]))
        nop = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]))
    close(out["pano_embeds"], z["pano_embeds"], 2e-5, what="pano_embeds")
    assert np.array_equal(out["pano_masks"].numpy(), z["pano_masks"])
    close(out["obj_embeds"], z["obj_embeds"], 2e-5, what="obj_embeds")
    assert np.array_equal(out["obj_masks"].numpy(), z["obj_masks"])
    close(nop["pano_embeds"], z["nopose_pano_embeds"], 2e-5, what="nopose")


def test_g1_scene_encoder_fuse_obj():
    z = gold("g1_encoder_fuseobj.npz")
    cfg, P = tiny_weights("fp32", fuse_obj=True)
    with torch.no_grad():
        out = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]), T(z["loc_fts"]), T(z["nav_types"]),
                              T(z["obj_img_fts"]), T(z["obj_lens"]), T(z["obj_loc_fts"]))
    close(out["pano_embeds"], z["pano_embeds"], 2e-5, what="pano_embeds(fuse_obj)")


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_g2_visual_token_lm(tag):
    z = gold(f"g2_lm_{tag}.npz")
    cfg, P = tiny_weights(tag)
    ids, am = T(z["input_ids"]), T(z["attention_mask"])
    with torch.no_grad():
        loss, logits, Hs = O.lm_forward(P, cfg, ids, am, labels=T(z["labels"]), cand_vis=T(z["cand_vis"]),
                                        hist_vis=T(z["hist_vis"]))
    real = am.bool()
    # bf16: same rounding points as HF -> only accumulate-order noise (1 bf16 ulp of ~|x|<=4)
    atol = 2e-5 if tag == "fp32" else 3e-2
    close(Hs[real], T(z["hidden_states"])[real], atol, what="hidden_states")
    close(logits[:, -10:][real[:, -10:]], T(z["logits"])[real[:, -10:]], atol, what="logits")
    close(loss, z["loss"], 1e-5 if tag == "fp32" else 2e-2, what="lm loss")


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_g3_g4_navigation_loss_grads(tag):
    z = gold(f"g3_nav_{tag}.npz")
    cfg, P = tiny_weights(tag)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    m = meta_of(z)
    pano = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]), T(z["loc_fts"]), T(z["nav_types"]))
    close(pano["pano_embeds"], z["pano_embeds"], 2e-5, what="pano_embeds")
    batch, m = nav_batch_from_gold(z, pano["pano_embeds"])
    torch.manual_seed(m["seed_before_nav"])
    out = O.navigation(P, cfg, batch, T(z["input_ids"]), T(z["attention_mask"]))
    assert [p.tolist() for p in out["perms"]] == m["perms"]
    close(out["fuse_embeds"], z["fuse_embeds"], 2e-5, what="fuse_embeds")
    atol = 2e-5 if tag == "fp32" else 1e-2
    close(out["fuse_logits"], z["fuse_logits"], atol, what="fuse_logits")
    loss = O.action_loss(out["fuse_logits"], torch.tensor(m["targets"])) / len(m["targets"])
    close(loss, z["loss"], atol, what="loss")
    loss.backward()
    names = [k[5:] for k in z if k.startswith("grad/")]
    for n in names:
        gref = T(z["grad/" + n]).float()
        g = P[n].grad.float()
        rel = (g - gref).norm() / (gref.norm() + 1e-12)
        assert rel < (1e-4 if tag == "fp32" else 6e-2), (n, rel.item())
    with_grad = sorted(k for k, v in P.items() if v.grad is not None and bool((v.grad != 0).any()))
    ref_with_grad = [str(s) for s in z["grad_names_with_grad"]]
    # params the reference leaves without grad must also be grad-free here
    assert set(with_grad) <= set(ref_with_grad)


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_g5_object_grounding_and_qa(tag):
    z = gold(f"g5_og_{tag}.npz")
    cfg, P = tiny_weights(tag)
    m = meta_of(z)
    with torch.no_grad():
        po = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]), T(z["loc_fts"]), T(z["nav_types"]),
                             T(z["obj_img_fts"]), T(z["obj_lens"]), T(z["obj_loc_fts"]))
        close(po["obj_embeds"], z["obj_embeds"], 2e-5, what="obj_embeds")
        b = dict(obj_embeds=po["obj_embeds"], obj_masks=po["obj_masks"], obj_loc_fts=po["obj_loc_fts"],
                 hist_vis=hist_lists(T(z["hist_vis_flat"]), m["hist_t"]))
        oo = O.object_grounding(P, cfg, b, T(z["input_ids"]), T(z["attention_mask"]))
    close(oo["obj_logits"], z["obj_logits"], 2e-5 if tag == "fp32" else 1e-2, what="obj_logits")

    q = gold(f"g5_qa_{tag}.npz")
    feats = [T(q["features"])[i, :int(n)] for i, n in enumerate(q["feat_lens"])]
    with torch.no_grad():
        loss = O.qa_3d_loss(P, cfg, feats, T(q["input_ids"]), T(q["attention_mask"]), T(q["token_type_ids"]))
    close(loss, q["loss"], 1e-5 if tag == "fp32" else 2e-2, what="3dqa loss")


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_g5_summarization_and_fgr2r_losses(tag):
    z = gold(f"g5_sum_{tag}.npz")
    cfg, P = tiny_weights(tag)
    m = meta_of(z)
    with torch.no_grad():
        ps = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]), T(z["loc_fts"]), T(z["nav_types"]))
        vp = torch.cat([torch.zeros_like(ps["pano_embeds"][:, :1]), ps["pano_embeds"]], 1)
        hv = hist_lists(T(z["hist_vis_flat"]), m["hist_t"])
        ls = O.summarization_loss(P, cfg, vp, T(z["vp_nav_masks"]), hv, T(z["sum_input_ids"]), T(z["sum_attention_mask"]),
                                  T(z["sum_token_type_ids"]))
        lq = O.summarization_loss(P, cfg, vp, T(z["vp_nav_masks"]), [[] for _ in m["hist_t"]], T(z["qa_input_ids"]),
                                  T(z["qa_attention_mask"]), T(z["qa_token_type_ids"]))
    tol = 1e-5 if tag == "fp32" else 4e-2
    close(ls, z["sum_loss"], tol, what="summarization loss")
    close(lq, z["qa_loss"], tol, what="fgr2r loss")


class _GoldTrie:
    """tools/trie.py protocol over the fixture's word list (no defaultdict side effects)"""

    class _N:
        def __init__(self):
            self.child = {}

    def __init__(self, words, eos):
        self.root, self.eos = self._N(), eos
        for w in words:
            cur = self.root
            for c in w:
                cur = cur.child.setdefault(int(c), self._N())

    def get_child_index(self, cur):
        return [self.eos] if not cur.child else list(cur.child.keys())

    def get_next_node(self, cur, w):
        return cur if not cur.child else cur.child.setdefault(int(w), self._N())


def g9_inputs(z):
    m = meta_of(z)
    feats = [T(z["qa_features"])[i, :int(n)] for i, n in enumerate(z["qa_feat_lens"])]
    words = [[int(c) for c in row if c >= 0] for row in z["trie_words"]]
    return m, feats, _GoldTrie(words, m["eos"])


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_g9_generation_matches_reference_generate(tag):
    """the oracle's cache-free greedy loop reproduces the token ids the reference's own generate() calls returned:
    3dqa (free decoding, 6 new tokens) and summarization (trie-constrained, eos-terminated)."""
    z = gold(f"g9_generate_{tag}.npz")
    cfg, P = tiny_weights(tag)
    m, feats, trie = g9_inputs(z)
    with torch.no_grad():
        qa = O.qa_3d_generate(P, cfg, feats, T(z["qa_input_ids"]), T(z["qa_attention_mask"]), max_new_tokens=6,
                              eos_token_id=m["eos"], pad_token_id=m["pad"])
        ps = O.scene_encoder(P, cfg, T(z["sum_view_img_fts"]), T(z["sum_view_lens"]), T(z["sum_loc_fts"]), T(z["sum_nav_types"]))
        vp = torch.cat([torch.zeros_like(ps["pano_embeds"][:, :1]), ps["pano_embeds"]], 1)
        hv = hist_lists(T(z["sum_hist_vis_flat"]), m["hist_t"])
        sm = O.summarization_generate(P, cfg, vp, T(z["sum_vp_nav_masks"]), hv, T(z["sum_input_ids"]), T(z["sum_attention_mask"]),
                                      max_new_tokens=50, eos_token_id=m["eos"], pad_token_id=m["pad"], trie=trie)
    assert qa == z["qa_new_ids"].tolist(), (qa, z["qa_new_ids"].tolist())
    assert sm == z["sum_new_ids"].tolist(), (sm, z["sum_new_ids"].tolist())


def test_g8_clip_adamw():
    z = gold("g8_adamw.npz")
    ps = [T(z[f"p0_{i}"]) for i in range(3)]
    ps = [ps[0].bfloat16(), ps[1].bfloat16(), ps[2]]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for s in range(3):
        gs = [T(z[f"g{s}_{i}"]).to(ps[i].dtype) for i in range(3)]
        total = O.clip_grad_norm_(gs, 40.0)
        assert abs(float(total) - float(z["norms"][s])) <= 1e-5 * float(z["norms"][s]) + 1e-6
        for i in range(3):
            O.adamw_step_(ps[i], gs[i], ms[i], vs[i], s + 1, lr=1e-3)
            ref = T(z[f"p{s + 1}_{i}"])
            assert torch.equal(ps[i].float(), ref.float()), (s, i, (ps[i].float() - ref.float()).abs().max())


def test_g1_scene_encoder_real_size():
    """the encoder at its real size (h=1024, 16 heads x 64, ff=4096, 36 ragged views; F=1024 with objects, F=768)."""
    for F_ in (1024, 768):
        z = gold(f"g1_encoder_real_F{F_}.npz")
        cfg, P = tiny_weights("fp32", enc_hidden_size=1024, enc_num_heads=16, enc_intermediate_size=4096, image_feat_size=F_,
                              obj_feat_size=768)
        with torch.no_grad():
            if F_ == 1024:
                out = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]), T(z["loc_fts"]), T(z["nav_types"]),
                                      T(z["obj_img_fts"]), T(z["obj_lens"]), T(z["obj_loc_fts"]))
                close(out["obj_embeds"], z["obj_embeds"], 2e-5, what="obj_embeds")
            else:
                out = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]), T(z["loc_fts"]), T(z["nav_types"]))
        close(out["pano_embeds"], z["pano_embeds"], 5e-5, what=f"pano_embeds F={F_}")
        assert np.array_equal(out["pano_masks"].numpy(), z["pano_masks"])


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_g10_training_step_gradients_of_every_mode(tag):
    """object_grounding / summarization / fgr2r / 3dqa: loss AND parameter gradients of one training step each, against
    the reference's own backward (fixture G10; mp3d_agent.py:788-909, llava.py:38-42)."""
    from util import grad_fixture_errors
    z10 = gold(f"g10_grads_{tag}.npz")
    cfg, P0 = tiny_weights(tag)
    B = 3
    ltol, gtol = (1e-5, 2e-4) if tag == "fp32" else (3e-2, 8e-2)

    def fresh():
        return {k: v.clone().requires_grad_(True) for k, v in P0.items()}

    def check(prefix, P, loss):
        close(loss, z10[prefix + "/loss"], ltol, what=prefix + " loss")
        loss.backward()
        errs = grad_fixture_errors(z10, prefix, lambda n: P[n].grad)
        assert errs and max(errs.values()) < gtol, (prefix, {k: round(v, 5) for k, v in errs.items() if v >= gtol})
        with_grad = {k for k, v in P.items() if v.grad is not None and bool((v.grad != 0).any())}
        assert with_grad <= {str(s) for s in z10[prefix + "/grad_names_with_grad"]}

    # object grounding
    z = gold(f"g5_og_{tag}.npz")
    m = meta_of(z)
    P = fresh()
    po = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]), T(z["loc_fts"]), T(z["nav_types"]),
                         T(z["obj_img_fts"]), T(z["obj_lens"]), T(z["obj_loc_fts"]))
    b = dict(obj_embeds=po["obj_embeds"], obj_masks=po["obj_masks"], obj_loc_fts=po["obj_loc_fts"],
             hist_vis=hist_lists(T(z["hist_vis_flat"]), m["hist_t"]))
    oo = O.object_grounding(P, cfg, b, T(z["input_ids"]), T(z["attention_mask"]))
    check("og", P, O.action_loss(oo["obj_logits"], T(z10["og/targets"])) * 0.5 / B)
    # summarization + fgr2r
    z = gold(f"g5_sum_{tag}.npz")
    m = meta_of(z)
    for prefix, key, hv in (("sum", "sum", hist_lists(T(z["hist_vis_flat"]), m["hist_t"])), ("fgr2r", "qa", [[] for _ in range(B)])):
        P = fresh()
        ps = O.scene_encoder(P, cfg, T(z["view_img_fts"]), T(z["view_lens"]), T(z["loc_fts"]), T(z["nav_types"]))
        vp = torch.cat([torch.zeros_like(ps["pano_embeds"][:, :1]), ps["pano_embeds"]], 1)
        l_ = O.summarization_loss(P, cfg, vp, T(z["vp_nav_masks"]), hv, T(z[key + "_input_ids"]), T(z[key + "_attention_mask"]),
                                  T(z[key + "_token_type_ids"]))
        check(prefix, P, l_ * float(z10[prefix + "/coef"]) / B)
    # 3dqa
    q = gold(f"g5_qa_{tag}.npz")
    feats = [T(q["features"])[i, :int(n)] for i, n in enumerate(q["feat_lens"])]
    P = fresh()
    l_ = O.qa_3d_loss(P, cfg, feats, T(q["input_ids"]), T(q["attention_mask"]), T(q["token_type_ids"]))
    check("qa", P, l_ * float(z10["qa/coef"]))


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_g11_fp8_weight_only(tag):
    """weight-only fp8: the oracle's quantiser against torch's float8_e4m3fn vector (bit-exact codes and scales), and the
    oracle navigation on de-quantised weights against the reference run on de-quantised weights."""
    z = gold(f"g11_fp8_{tag}.npz")
    q, s = O.fp8_quantize_rows(T(z["quant_w"]).bfloat16())
    assert np.array_equal(q.view(torch.uint8).numpy(), z["quant_codes"])
    assert np.array_equal(s.numpy(), z["quant_scales"])
    assert torch.equal(O.fp8_dequantize(q, s).float(), T(z["quant_dequant"]).float())
    assert float(s[5]) == 1.0 and int(q.view(torch.uint8)[5].max()) == 0                 # all-zero row
    # navigation on the de-quantised weights (G3 inputs)
    z3 = gold(f"g3_nav_{tag}.npz")
    cfg, P = tiny_weights(tag)
    Pq = O.fp8_weight_only_state_dict(P)
    assert not torch.equal(Pq["lang_model.model.layers.0.self_attn.q_proj.weight"], P["lang_model.model.layers.0.self_attn.q_proj.weight"])
    assert Pq["lang_model.lm_head.weight"] is P["lang_model.lm_head.weight"]
    with torch.no_grad():
        pano = O.scene_encoder(Pq, cfg, T(z3["view_img_fts"]), T(z3["view_lens"]), T(z3["loc_fts"]), T(z3["nav_types"]))
        batch, m = nav_batch_from_gold(z3, pano["pano_embeds"])
        torch.manual_seed(m["seed_before_nav"])
        out = O.navigation(Pq, cfg, batch, T(z3["input_ids"]), T(z3["attention_mask"]))
    close(out["fuse_logits"], z["fuse_logits"], 2e-5 if tag == "fp32" else 1e-2, what="fp8 fuse_logits")
    # and the quantisation really changes the result (the fixture is not the unquantised G3 again)
    fin = np.isfinite(z["fuse_logits"])
    assert np.abs(z["fuse_logits"][fin] - z3["fuse_logits"][fin]).max() > 5e-3


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_g12_episode_accumulated_gradients(tag):
    """the oracle through a whole 3-step episode (panorama -> navigation -> CE -> backward per step, history appended from its
    OWN fuse_embeds, gradients accumulated) against the reference's run of the same episode (mp3d_agent.py:659-778)"""
    from util import episode_step_batch, grad_fixture_errors
    z = gold(f"g12_episode_{tag}.npz")
    meta = meta_of(z)
    cfg, P = tiny_weights(tag)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    B = meta["B"]
    hist = [[] for _ in range(B)]
    atol = 2e-5 if tag == "fp32" else 1.6e-2
    for t in range(len(meta["steps"])):
        pre = f"s{t}/"
        pano = O.scene_encoder(P, cfg, T(z[pre + "view_img_fts"]), T(z[pre + "view_lens"]), T(z[pre + "loc_fts"]), T(z[pre + "nav_types"]))
        close(pano["pano_embeds"], z[pre + "pano_embeds"], 2e-5, what=f"step {t} pano_embeds")
        batch, ms = episode_step_batch(z, meta, t, pano["pano_embeds"], hist)
        torch.manual_seed(ms["seed_before_nav"])
        out = O.navigation(P, cfg, batch, T(z[pre + "input_ids"]), T(z[pre + "attention_mask"]))
        assert [p.tolist() for p in out["perms"]] == ms["perms"]
        close(out["fuse_embeds"], z[pre + "fuse_embeds"], 2e-5, what=f"step {t} fuse_embeds")
        close(out["fuse_logits"], z[pre + "fuse_logits"], atol, what=f"step {t} fuse_logits")
        tg = torch.tensor(ms["targets"])
        loss = O.action_loss(out["fuse_logits"], tg) * meta["train_ml"] / B / meta["accum"]
        close(loss, z[pre + "loss"], atol, what=f"step {t} loss")
        loss.backward()
        for b in range(B):
            if ms["targets"][b] != -100:
                hist[b].append(out["fuse_embeds"][b][ms["targets"][b]].detach())
    assert [len(h) for h in hist] == meta["hist_final"]
    close(torch.stack([v for h in hist for v in h], 0), z["hist_final_flat"], 2e-5, what="history rows")
    errs = grad_fixture_errors(z, "acc", lambda n: P[n].grad)
    assert errs and max(errs.values()) < (1e-4 if tag == "fp32" else 6e-2), errs
    ref_with_grad = [str(s) for s in z["acc/grad_names_with_grad"]]
    with_grad = sorted(k for k, v in P.items() if v.grad is not None and bool((v.grad != 0).any()))
    assert set(with_grad) <= set(ref_with_grad)


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_g14_training_mode_step_with_the_references_dropout_masks(tag):
    """The TRAINING-mode path (VERDICT r4 missing #1/#2): the reference ran in .train() with every dropout mask recorded
    (fixture G14: drop_env on views and objects, the embedding dropout, and per encoder layer the attention-PROBABILITY dropout
    of nn.MultiheadAttention + dropout1 / dropout / dropout2; nav_model.py:91,99-102, image_embedding.py:73-74,
    detr_transformer.py:138,141,146-147,170-182); the oracle consumes the same masks: forward values, both losses, and the
    gradients after the navigation backward and after the accumulated object-grounding backward."""
    from util import grad_fixture_errors, dropout_masks_from_gold
    z = gold(f"g14_train_{tag}.npz")
    cfg, P = tiny_weights(tag)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    m = meta_of(z)
    dm = dropout_masks_from_gold(z)
    assert sorted(dm) == sorted(m["mask_order"]) and len(dm) == 3 + 4 * cfg.num_pano_layers
    pin = dict(view_img_fts=T(z["view_img_fts"]), view_lens=T(z["view_lens"]), loc_fts=T(z["loc_fts"]), nav_types=T(z["nav_types"]),
               obj_img_fts=T(z["obj_img_fts"]), obj_lens=T(z["obj_lens"]), obj_loc_fts=T(z["obj_loc_fts"]))
    pano = O.panorama(P, cfg, pin, training=True, dmasks=dm)
    close(pano["pano_embeds"], z["pano_embeds"], 2e-5, what="pano_embeds (train)")
    close(pano["obj_embeds"], z["obj_embeds"], 2e-5, what="obj_embeds (train)")
    # the masks matter: the eval-mode encoder is far away from the fixture
    with torch.no_grad():
        ev = O.panorama(P, cfg, pin, training=False)
    assert (ev["pano_embeds"] - T(z["pano_embeds"])).abs().max() > 1e-2
    # ... and so does the attention-probability mask alone
    with torch.no_grad():
        noattn = O.panorama(P, cfg, pin, training=True, dmasks={k: (torch.ones_like(v) if k.endswith(".attn") else v) for k, v in dm.items()})
    assert (noattn["pano_embeds"] - T(z["pano_embeds"])).abs().max() > 1e-3
    batch, m = nav_batch_from_gold(z, pano["pano_embeds"])
    torch.manual_seed(m["seed_before_nav"])
    out = O.navigation(P, cfg, batch, T(z["input_ids"]), T(z["attention_mask"]))
    assert [p.tolist() for p in out["perms"]] == m["perms"]
    close(out["fuse_embeds"], z["fuse_embeds"], 2e-5, what="fuse_embeds")
    atol, gtol = (2e-5, 2e-4) if tag == "fp32" else (1e-2, 8e-2)
    close(out["fuse_logits"], z["fuse_logits"], atol, what="fuse_logits")
    B = len(m["targets"])
    loss = O.action_loss(out["fuse_logits"], torch.tensor(m["targets"])) * m["nav_coef"] / B
    close(loss, z["loss"], atol, what="nav loss")
    loss.backward(retain_graph=True)
    errs = grad_fixture_errors(z, "nav", lambda n: P[n].grad)
    assert len(errs) >= 20 and max(errs.values()) < gtol, {k: round(v, 5) for k, v in errs.items() if v >= gtol}
    with_grad = {k for k, v in P.items() if v.grad is not None and bool((v.grad != 0).any())}
    assert with_grad <= {str(s) for s in z["nav/grad_names_with_grad"]}
    ob = dict(obj_embeds=pano["obj_embeds"], obj_masks=pano["obj_masks"], obj_loc_fts=pano["obj_loc_fts"], hist_vis=batch["hist_vis"])
    oo = O.object_grounding(P, cfg, ob, T(z["og_input_ids"]), T(z["og_attention_mask"]))
    close(oo["obj_logits"], z["obj_logits"], atol, what="obj_logits")
    og_loss = O.action_loss(oo["obj_logits"], torch.tensor(m["og_targets"])) * m["og_coef"] / B
    close(og_loss, z["og_loss"], atol, what="og loss")
    og_loss.backward()
    errs = grad_fixture_errors(z, "acc", lambda n: P[n].grad)
    assert len(errs) >= 24 and max(errs.values()) < gtol, {k: round(v, 5) for k, v in errs.items() if v >= gtol}
