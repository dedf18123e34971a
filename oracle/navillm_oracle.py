"""CPU ORACLE for the NaviLLM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this module; `navillm_amd/` never does (the product path fails loudly
when the HIP library is missing -- see navillm_amd/lib.py).

What it is: a plain-PyTorch-CPU restatement, with explicit weights (a name->tensor
dict in the reference's state_dict layout) and no `transformers` dependency, of

  * the scene encoder           models/image_embedding.py:51-121, models/ops.py:6-41,
                                models/detr_transformer.py:71-89,170-182
  * the navigation fusion       models/nav_model.py:129-247
  * the visual-token causal LM  models/modified_lm.py:89-146 -> HF LlamaModel
                                (third-party: transformers, pinned 4.28.0 in the
                                reference's requirements.txt:21; restated from the
                                installed 5.15 source modeling_llama.py:53-67,130-160,
                                174-176,191-213 which SURVEY.md §8c found identical in math)
  * heads and losses            models/nav_model.py:234-242,407-451, train.py:229,
                                models/modified_lm.py:126-137
  * grad clip + AdamW           train.py:86-89, tools/optims.py:43-45
  * greedy generation           models/nav_model.py:324-341,388-402, models/modified_lm.py:10-30,184-199 -> HF
                                `generate(do_sample=False)` (third-party; restated as a cache-free recompute loop:
                                `greedy_generate`), pinned by fixture G9 = the reference's own generate() calls run
                                here with two library-side signature shims (tests/golden/make_golden.py)

Parity pinning: the reference ships no tests, so this oracle is pinned by golden
vectors produced HERE by importing the reference itself (tests/golden/make_golden.py,
fixtures tests/golden/*.npz); tests/test_oracle_golden.py checks every function below
against them.
"""
import math
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- helpers
def gen_seq_masks(seq_lens, max_len=None):
    """models/ops.py:33-41"""
    if max_len is None:
        max_len = int(max(seq_lens))
    return torch.arange(max_len).unsqueeze(0) < torch.as_tensor(seq_lens).unsqueeze(1)


def pad_tensors_wgrad(tensors):
    """models/ops.py:44-66"""
    max_len = max(t.size(0) for t in tensors)
    out = []
    for t in tensors:
        if t.size(0) < max_len:
            t = torch.cat([t, torch.zeros([max_len - t.size(0)] + list(t.shape[1:]), dtype=t.dtype)], 0)
        out.append(t)
    return torch.stack(out, 0)


def _ln(x, P, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), P[prefix + ".weight"], P[prefix + ".bias"], eps)


def _lin(x, P, prefix):
    return F.linear(x, P[prefix + ".weight"], P[prefix + ".bias"])


def _dropout(x, p, training, masks, key):
    """Dropout with an optional injected keep-mask (GPU dropout cannot be bit-matched,
    SURVEY.md §7 'RNG parity'); masks[key] holds 0/1 keep flags."""
    if not training or p == 0.0:
        return x
    if masks is not None and key in masks:
        return x * (masks[key].to(x.dtype) / (1.0 - p))      # torch: noise = bernoulli(1-p) / (1-p); x * noise
    return F.dropout(x, p, True)


# --------------------------------------------------------------------------- scene encoder
def mha_self(x, key_pad, P, prefix, num_heads, drop_p=0.0, training=False, dmasks=None, key=None):
    """nn.MultiheadAttention (packed in_proj, key_padding_mask, dropout=drop_p ON THE ATTENTION PROBABILITIES --
    detr_transformer.py:138 constructs it with dropout=0.1, torch applies it to the softmax output before P.V) on batch-first
    x [B,N,h].  detr_transformer.py:173-175.  dmasks[key]: injected keep flags [B,heads,N,N]."""
    B, N, h = x.shape
    hd = h // num_heads
    qkv = F.linear(x, P[prefix + ".in_proj_weight"], P[prefix + ".in_proj_bias"])
    q, k, v = qkv.split(h, dim=-1)
    q = q.view(B, N, num_heads, hd).transpose(1, 2)
    k = k.view(B, N, num_heads, hd).transpose(1, 2)
    v = v.view(B, N, num_heads, hd).transpose(1, 2)
    s = (q * (1.0 / math.sqrt(hd))) @ k.transpose(-1, -2)
    s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = _dropout(p, drop_p, training, dmasks, key)
    o = (p @ v).transpose(1, 2).reshape(B, N, h)
    return F.linear(o, P[prefix + ".out_proj.weight"], P[prefix + ".out_proj.bias"])


def pano_encoder(x, masks, P, cfg, prefix="img_embeddings.pano_encoder", training=False, dmasks=None):
    """TransformerEncoder of pre-norm layers + final LN(1e-12).
    detr_transformer.py:71-89,170-182 ; ops.py:6-18"""
    key_pad = masks.logical_not()
    for i in range(cfg.num_pano_layers):
        p = f"{prefix}.layers.{i}"
        y = _ln(x, P, p + ".norm1", 1e-5)
        x = x + _dropout(mha_self(y, key_pad, P, p + ".self_attn", cfg.enc_num_heads, cfg.enc_dropout, training, dmasks,
                                  f"l{i}.attn"),
                         cfg.enc_dropout, training, dmasks, f"l{i}.drop1")
        y = _ln(x, P, p + ".norm2", 1e-5)
        y = F.gelu(_lin(y, P, p + ".linear1"))
        y = _dropout(y, cfg.enc_dropout, training, dmasks, f"l{i}.drop")
        x = x + _dropout(_lin(y, P, p + ".linear2"), cfg.enc_dropout, training, dmasks, f"l{i}.drop2")
    return _ln(x, P, prefix + ".norm", 1e-12)


def scene_encoder(P, cfg, view_img_fts, view_lens, loc_fts=None, nav_types=None,
                  obj_img_fts=None, obj_lens=None, obj_loc_fts=None, training=False, dmasks=None):
    """ImageEmbeddings.forward_panorama_per_step, image_embedding.py:51-121."""
    e = "img_embeddings"
    ret = {}
    B = view_img_fts.shape[0]
    x = _ln(_lin(view_img_fts, P, e + ".img_linear"), P, e + ".img_layer_norm", 1e-12)
    if loc_fts is None:
        loc_fts = torch.zeros(x.shape[:2] + (7,), dtype=torch.float)
    x = x + _ln(_lin(loc_fts, P, e + ".loc_linear"), P, e + ".loc_layer_norm", 1e-12)
    if nav_types is None:
        nav_types = torch.ones(x.shape[:2], dtype=torch.long)
    x = x + P[e + ".nav_type_embedding.weight"][nav_types.long()]
    x = _ln(x, P, e + ".layer_norm", 1e-12)
    x = _dropout(x, cfg.enc_dropout, training, dmasks, "emb.drop")
    pano_masks = gen_seq_masks(view_lens)
    if cfg.num_pano_layers > 0:
        if cfg.fuse_obj:
            o = _ln(_lin(obj_img_fts, P, e + ".obj_linear.0"), P, e + ".obj_linear.1", 1e-12) \
                + _ln(_lin(obj_loc_fts, P, e + ".loc_linear"), P, e + ".loc_layer_norm", 1e-12) \
                + P[e + ".nav_type_embedding.weight"][2]
            fuse = pad_tensors_wgrad([torch.cat([x[b, :view_lens[b]], o[b, :obj_lens[b]]], 0) for b in range(B)])
            fmask = gen_seq_masks(torch.as_tensor(view_lens) + torch.as_tensor(obj_lens))
            fuse = pano_encoder(fuse, fmask, P, cfg, training=training, dmasks=dmasks)
            x = pad_tensors_wgrad([fuse[b, :view_lens[b]] for b in range(B)])
        else:
            x = pano_encoder(x, pano_masks, P, cfg, training=training, dmasks=dmasks)
    x = _lin(x, P, e + ".mapper")
    x = x.masked_fill(pano_masks.logical_not().unsqueeze(-1), 0)
    ret.update(pano_embeds=x, pano_masks=pano_masks)
    if obj_img_fts is not None and obj_img_fts.shape[1] > 0:
        oe = _ln(_lin(obj_img_fts, P, e + ".obj_projector.0"), P, e + ".obj_projector.1", 1e-12)
        assert oe.shape[:2] == obj_loc_fts.shape[:2]
        ret.update(obj_embeds=oe, obj_loc_fts=obj_loc_fts, obj_masks=gen_seq_masks(obj_lens))
    return ret


def panorama(P, cfg, batch, training=False, dmasks=None):
    """NavModel.forward(mode='panorama'), nav_model.py:96-111: `drop_env` (nn.Dropout(feat_dropout), :91) on the raw view
    features, and on the object features when the batch carries them (:99-102), then the scene encoder.  Dropout keys (the
    order the reference calls them in one training panorama): drop_env.view, drop_env.obj, emb.drop, then per encoder layer
    l{i}.attn (attention probabilities), l{i}.drop1, l{i}.drop (FFN), l{i}.drop2."""
    v = _dropout(batch["view_img_fts"], cfg.feat_dropout, training, dmasks, "drop_env.view")
    o = batch.get("obj_img_fts")
    if o is not None:
        o = _dropout(o, cfg.feat_dropout, training, dmasks, "drop_env.obj")
    return scene_encoder(P, cfg, v, batch["view_lens"], batch.get("loc_fts"), batch.get("nav_types"), o,
                         batch.get("obj_lens"), batch.get("obj_loc_fts"), training=training, dmasks=dmasks)


# --------------------------------------------------------------------------- Llama
def rms_norm(x, w, eps):
    """HF LlamaRMSNorm: fp32 statistics, cast back, then weight multiply."""
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(x.dtype)


def rope_tables(S, hd, theta, dtype):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    fr = torch.outer(torch.arange(S).float(), inv)
    emb = torch.cat([fr, fr], -1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def llama_decoder(P, cfg, inputs_embeds, attention_mask, prefix="lang_model.model", layers=None, final_norm=True, collect=None):
    """LlamaModel.forward with inputs_embeds + attention_mask only: positions are
    arange(S) INCLUDING left padding, causal + key-padding additive mask,
    eager attention with fp32 softmax.
    Test instruments: `layers` (iterable of layer indices, default all) runs a sub-range, `final_norm=False` returns the last
    layer's output before the final RMSNorm, `collect` (a dict) receives every executed layer's INPUT under its index."""
    x = inputs_embeds
    B, S, d = x.shape
    H, hd = cfg.num_heads, cfg.head_dim
    dt = x.dtype
    cos, sin = rope_tables(S, hd, cfg.rope_theta, dt)
    neg = torch.finfo(dt).min
    allowed = torch.tril(torch.ones(S, S, dtype=torch.bool))[None] & attention_mask.bool()[:, None, :]
    amask = torch.zeros(B, 1, S, S, dtype=dt).masked_fill(allowed.logical_not()[:, None], neg)
    scaling = hd ** -0.5
    for i in (range(cfg.num_layers) if layers is None else layers):
        p = f"{prefix}.layers.{i}"
        if collect is not None:
            collect[i] = x
        n = rms_norm(x, P[p + ".input_layernorm.weight"], cfg.rms_norm_eps)
        q = F.linear(n, P[p + ".self_attn.q_proj.weight"]).view(B, S, H, hd).transpose(1, 2)
        k = F.linear(n, P[p + ".self_attn.k_proj.weight"]).view(B, S, H, hd).transpose(1, 2)
        v = F.linear(n, P[p + ".self_attn.v_proj.weight"]).view(B, S, H, hd).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        w = torch.matmul(q, k.transpose(2, 3)) * scaling + amask
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(dt)
        a = torch.matmul(w, v).transpose(1, 2).reshape(B, S, d)
        x = x + F.linear(a, P[p + ".self_attn.o_proj.weight"])
        n = rms_norm(x, P[p + ".post_attention_layernorm.weight"], cfg.rms_norm_eps)
        g = F.linear(n, P[p + ".mlp.gate_proj.weight"])
        u = F.linear(n, P[p + ".mlp.up_proj.weight"])
        x = x + F.linear(F.silu(g) * u, P[p + ".mlp.down_proj.weight"])
    if not final_norm:
        return x
    return rms_norm(x, P[prefix + ".norm.weight"], cfg.rms_norm_eps)


def inject_visual_tokens(E, input_ids, cfg, cand_vis=None, hist_vis=None, obj_vis=None):
    """modified_lm.py:100-110: E[ids==tok] += vis, rows in (batch, position) order;
    one rounding: lm_dtype(f32(E) + f32(vis)) (SURVEY.md §7 probe)."""
    E = E.clone()
    for tok, vis in ((cfg.cand_token_id, cand_vis), (cfg.hist_token_id, hist_vis), (cfg.obj_token_id, obj_vis)):
        loc = input_ids == tok
        if int(loc.sum()) != 0:
            E[loc] = (E[loc].float() + vis.float()).to(E.dtype)
    return E


def lm_forward(P, cfg, input_ids, attention_mask, labels=None, cand_vis=None, hist_vis=None,
               obj_vis=None, need_logits=True):
    """ModifiedLM.forward, modified_lm.py:89-146. Returns (loss, logits, hidden_states)."""
    E = P["lang_model.model.embed_tokens.weight"][input_ids]
    E = inject_visual_tokens(E, input_ids, cfg, cand_vis, hist_vis, obj_vis)
    Hs = llama_decoder(P, cfg, E, attention_mask)
    logits = loss = None
    if need_logits or labels is not None:
        logits = F.linear(Hs, P["lang_model.lm_head.weight"])
        m = torch.zeros(logits.shape[-1], dtype=torch.bool)
        m[list(cfg.special_token_ids)] = True
        logits = logits.masked_fill(m, float("-inf"))
    if labels is not None:
        sl = logits[..., :-1, :].contiguous().view(-1, cfg.vocab_size)
        loss = F.cross_entropy(sl, labels[..., 1:].contiguous().view(-1))
    return loss, logits, Hs


def greedy_generate(P, cfg, input_ids, attention_mask, cand_vis=None, hist_vis=None, max_new_tokens=20, eos_token_id=2,
                    pad_token_id=0, trie=None):
    """`self.lang_model.generate(do_sample=False, ...)` as nav_model.py:324-341,388-402 calls it, restated WITHOUT a
    K/V cache: every step re-runs `lm_forward` on the whole left-padded batch (quadratic, small cases only).
    HF greedy bookkeeping (transformers 4.28 `GenerationMixin.greedy_search`): logits of the last position -> logits
    processors (TrieLogitsProcessor, modified_lm.py:10-30) -> argmax; finished rows emit pad_token_id; a row finishes
    when it emits eos_token_id; stop when all rows are finished or after max_new_tokens.
    Parity note: HF's generation path numbers positions per sample (cumsum of the mask) while this recompute numbers them
    over the padding like the training path does; RoPE depends on differences only, so the two agree up to bf16
    rounding.  Pinned by fixture G9 (tests/golden/g9_generate_*.npz): the reference's own generate() calls, run with two
    library-side signature shims (tests/golden/make_golden.py::install_generation_shims).  Returns (B lists of new
    tokens, last-step logits)."""
    ids, am = input_ids.clone(), attention_mask.clone()
    B = ids.shape[0]
    out = [[] for _ in range(B)]
    unfinished = [True] * B
    nodes = [trie.root for _ in range(B)] if trie is not None else None
    for step in range(max_new_tokens):
        _, logits, _ = lm_forward(P, cfg, ids, am, cand_vis=cand_vis, hist_vis=hist_vis)
        lg = logits[:, -1, :].float()
        if trie is not None:
            if step > 0:
                nodes = [trie.get_next_node(nodes[b], int(ids[b, -1])) for b in range(B)]
            allow = torch.zeros_like(lg, dtype=torch.bool)
            for b in range(B):
                allow[b, trie.get_child_index(nodes[b])] = True
            lg = lg.masked_fill(~allow, float("-inf"))
        nxt = lg.argmax(-1).tolist()
        col = []
        for b in range(B):
            t = nxt[b] if unfinished[b] else pad_token_id
            out[b].append(t)
            col.append(t)
            if unfinished[b] and t == eos_token_id:
                unfinished[b] = False
        ids = torch.cat([ids, torch.tensor(col, dtype=ids.dtype)[:, None]], 1)
        am = torch.cat([am, torch.ones((B, 1), dtype=am.dtype)], 1)
        if not any(unfinished):
            break
    return out, lg


# --------------------------------------------------------------------------- navigation
def _seq2(x, P, prefix):
    """nn.Sequential(Linear, LayerNorm(1e-12))"""
    return _ln(_lin(x, P, prefix + ".0"), P, prefix + ".1", 1e-12)


def navigation_fusion(P, cfg, batch):
    """nav_model.py:141-197 (gmap/vp fusion). Returns fuse_embeds [B,G,d] fp32, cand_masks."""
    g_img, g_step, g_pos = batch["gmap_img_embeds"], batch["gmap_step_ids"], batch["gmap_pos_fts"]
    g_masks, g_vis, g_vpids = batch["gmap_masks"], batch["gmap_visited_masks"], batch["gmap_vpids"]
    vp_img, vp_pos, vp_cand_vpids = batch["vp_img_embeds"], batch["vp_pos_fts"], batch["vp_cand_vpids"]
    pano_masks = batch["pano_masks"]
    B = vp_img.size(0)
    gmap = g_img + P["gmap_step_embeddings.weight"][g_step] + _seq2(g_pos, P, "gmap_pos_embeddings")
    vp = vp_img + _seq2(vp_pos, P, "vp_pos_embeddings")
    gmap = gmap.masked_fill(g_vis.unsqueeze(-1), 0.).masked_fill(g_masks.logical_not().unsqueeze(-1), 0.)
    vp = vp.masked_fill(pano_masks.logical_not().unsqueeze(-1), 0.)
    ttype = torch.zeros(gmap.shape[:2], dtype=torch.long)
    rows = []
    for i in range(B):
        visited = set(v for v, m in zip(g_vpids[i], g_vis[i]) if m)
        tmp = {}
        for j, cv in enumerate(vp_cand_vpids[i]):
            if j > 0 and cv not in visited:
                tmp[cv] = vp[i, j]
        row = []
        for j, v in enumerate(g_vpids[i]):
            r = gmap[i, j]
            if j > 0 and v not in visited:
                if v in tmp:
                    r = r + tmp[v]
                else:
                    ttype[i, j] = 1
            row.append(r)
        row += [gmap[i, j] for j in range(len(g_vpids[i]), gmap.shape[1])]
        rows.append(torch.stack(row, 0))
    fuse = torch.stack(rows, 0) + P["token_type_embeddings.weight"][ttype]
    fuse = fuse.masked_fill(g_vis.unsqueeze(-1), 0.).masked_fill(g_masks.logical_not().unsqueeze(-1), 0.)
    cand_masks = g_masks & g_vis.logical_not()
    return fuse, cand_masks


def navigation(P, cfg, batch, input_ids, attention_mask, perms=None):
    """NavModel.forward_navigation (nav_model.py:129-247) after tokenisation.
    `perms`: optional list of per-sample candidate permutations; if None they are drawn
    with torch.randperm in the reference's call order (nav_model.py:216-223)."""
    fuse, cand_masks = navigation_fusion(P, cfg, batch)
    B = fuse.shape[0]
    cand_nums = cand_masks.sum(-1)
    hv = [v for vis in batch["hist_vis"] for v in vis]
    hist_vis = torch.stack(hv, 0) if hv else None
    cand_embeds, inv_perms, used = [], [], []
    for b in range(B):
        ce = fuse[b][cand_masks[b]][1:]
        rp = torch.randperm(ce.shape[0]) if perms is None else perms[b]
        ip = torch.arange(ce.shape[0])
        ip[rp] = torch.arange(ce.shape[0])
        inv_perms.append(ip)
        used.append(rp)
        cand_embeds.append(ce[rp])
    cand_embeds = torch.cat(cand_embeds, 0)
    _, _, Hs = lm_forward(P, cfg, input_ids, attention_mask, cand_vis=cand_embeds, hist_vis=hist_vis,
                          need_logits=False)
    lm_dt = Hs.dtype
    pred = F.linear(Hs[input_ids == cfg.cls_token_ids[0]], P["out_head.0.weight"], P["out_head.0.bias"])
    logits = torch.zeros(fuse.shape[:2], dtype=lm_dt)
    rows = []
    for i in range(B):
        n = int(cand_nums[i])
        vals = torch.cat([pred[i, 0:1], pred[i, 1:n][inv_perms[i]]], 0)
        rows.append(torch.zeros(fuse.shape[1], dtype=lm_dt).masked_scatter(cand_masks[i], vals))
    logits = torch.stack(rows, 0).masked_fill(cand_masks.logical_not(), -float("inf"))
    return {"fuse_embeds": fuse.detach(), "fuse_logits": logits, "perms": used, "hidden_states": Hs}


def object_grounding(P, cfg, batch, input_ids, attention_mask):
    """NavModel.forward_object_grounding, nav_model.py:407-451."""
    oe = batch["obj_embeds"] + _seq2(batch["obj_loc_fts"], P, "obj_pos_embeddings")
    om = batch["obj_masks"].bool()
    cand_nums = om.sum(1) + 1
    hv = [v for vis in batch["hist_vis"] for v in vis]
    hist_vis = torch.stack(hv, 0) if hv else None
    _, _, Hs = lm_forward(P, cfg, input_ids, attention_mask, cand_vis=oe[om], hist_vis=hist_vis, need_logits=False)
    pred = F.linear(Hs[input_ids == cfg.cls_token_ids[0]], P["out_head.0.weight"], P["out_head.0.bias"])
    col = torch.arange(pred.shape[1])[None]
    return {"obj_logits": pred.masked_fill(col >= cand_nums[:, None], float("-inf"))}


def qa_3d_loss(P, cfg, features, input_ids, attention_mask, token_type_ids):
    """NavModel.forward_3dqa training branch, nav_model.py:346-385."""
    vf = pad_tensors_wgrad(list(features))
    vl = torch.tensor([f.shape[0] for f in features])
    out = scene_encoder(P, cfg, vf, vl)
    pe, pm = out["pano_embeds"], out["pano_masks"]
    pe = pe + _seq2(torch.zeros(pe.shape[:2] + (14,)), P, "vp_pos_embeddings")
    pe = pe + P["token_type_embeddings.weight"][0]
    labels = input_ids.clone()
    labels[token_type_ids == 0] = -100
    loss, _, _ = lm_forward(P, cfg, input_ids, attention_mask, labels=labels, cand_vis=pe[pm])
    return loss


def summarization_loss(P, cfg, vp_img_embeds, vp_nav_masks, hist_vis, input_ids, attention_mask, token_type_ids):
    """NavModel.forward_summarization training branch (also mode 'embodied_qa'), nav_model.py:251-319."""
    x = vp_img_embeds[:, 1:, :]
    nm = vp_nav_masks[:, 1:].bool()
    x = x + _seq2(torch.zeros(x.shape[:2] + (14,)), P, "vp_pos_embeddings")
    x = x + P["token_type_embeddings.weight"][0]
    hv = [v for vis in hist_vis for v in vis]
    hist = torch.stack(hv, 0) if hv else None
    labels = input_ids.clone()
    labels[token_type_ids == 0] = -100
    loss, _, _ = lm_forward(P, cfg, input_ids, attention_mask, labels=labels, cand_vis=x[nm], hist_vis=hist)
    return loss


def qa_3d_generate(P, cfg, features, input_ids, attention_mask, **gen):
    """NavModel.forward_3dqa inference branch, nav_model.py:386-404 -> new token ids per sample."""
    vf = pad_tensors_wgrad(list(features))
    vl = torch.tensor([f.shape[0] for f in features])
    out = scene_encoder(P, cfg, vf, vl)
    pe, pm = out["pano_embeds"], out["pano_masks"]
    pe = pe + _seq2(torch.zeros(pe.shape[:2] + (14,)), P, "vp_pos_embeddings")
    pe = pe + P["token_type_embeddings.weight"][0]
    return greedy_generate(P, cfg, input_ids, attention_mask, cand_vis=pe[pm], **gen)[0]


def summarization_generate(P, cfg, vp_img_embeds, vp_nav_masks, hist_vis, input_ids, attention_mask, **gen):
    """NavModel.forward_summarization inference branch, nav_model.py:320-343 (max_new_tokens=50, optional trie)."""
    x = vp_img_embeds[:, 1:, :]
    nm = vp_nav_masks[:, 1:].bool()
    x = x + _seq2(torch.zeros(x.shape[:2] + (14,)), P, "vp_pos_embeddings")
    x = x + P["token_type_embeddings.weight"][0]
    hv = [v for vis in hist_vis for v in vis]
    hist = torch.stack(hv, 0) if hv else None
    return greedy_generate(P, cfg, input_ids, attention_mask, cand_vis=x[nm], hist_vis=hist, **gen)[0]


def action_loss(logits, targets):
    """train.py:229 criterion: CrossEntropyLoss(ignore_index=-100, reduction='sum')."""
    return F.cross_entropy(logits, targets, ignore_index=-100, reduction="sum")


# --------------------------------------------------------------------------- optimizer
def clip_grad_norm_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, 40.) (train.py:87) on a list of grads."""
    norms = [torch.linalg.vector_norm(g, 2) for g in grads]
    total = torch.linalg.vector_norm(torch.stack([n.float() for n in norms]), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)   # fp32 0-dim coefficient: product formed in fp32, rounded once to g.dtype
    return total


def adamw_step_(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor update (tools/optims.py:43-45 uses defaults);
    every intermediate is rounded to the tensor dtype, as torch does for bf16 params."""
    p.mul_(1 - lr * wd)
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


# --------------------------------------------------------------------------- weight-only fp8
# No reference counterpart: the reference has no quantised path (models/nav_model.py:40-47 builds a bf16/fp32 LM).  The
# semantics are "the reference run on de-quantised weights", pinned by tests/golden/g11_fp8_*.npz (the reference model
# loaded with weights that make_golden.py fake-quantises with torch's own float8_e4m3fn conversion).
FP8_MAX = 448.0      # largest finite OCP e4m3fn value


def fp8_quantize_rows(W):
    """per-OUTPUT-channel (row) symmetric quantisation: s[n] = max|W[n,:]| / 448 (1 for an all-zero row),
    q = e4m3fn(W / s) round-to-nearest-even -> (q as torch.float8_e4m3fn [N,K], s fp32 [N])"""
    Wf = W.float()
    s = Wf.abs().amax(1) / FP8_MAX
    s = torch.where(s > 0, s, torch.ones_like(s))
    return (Wf / s[:, None]).to(torch.float8_e4m3fn), s


def fp8_dequantize(q, s, dtype=torch.bfloat16):
    """the weight every GEMM of the fp8 path multiplies with: dtype(s[n] * q[n,k])"""
    return (q.float() * s[:, None]).to(dtype)


def is_fp8_weight(name):
    """the decoder's seven Linear weights per layer (12.7 of Vicuna-13B's 13.0 B parameters); embeddings, norms, lm_head and
    the fp32 encoder stay as they are"""
    return name.startswith("lang_model.model.layers.") and name.endswith("_proj.weight")


def fp8_weight_only_state_dict(P):
    """state dict with every decoder Linear weight replaced by its de-quantised fp8 version"""
    out = {}
    for k, v in P.items():
        if is_fp8_weight(k):
            q, s = fp8_quantize_rows(v)
            out[k] = fp8_dequantize(q, s, v.dtype)
        else:
            out[k] = v
    return out
