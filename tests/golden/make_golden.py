#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to the
GPU box).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs (all small, committed):
    tiny_llama/{config.json,tokenizer.model,...}  tiny Llama dir the reference can load
    g_meta.json        state_dict key -> shape/dtype of the reference model (fp32 & bf16)
    g1_encoder*.npz    scene encoder          
    g2_lm_*.npz        visual-token LM        (G2)
    g3_nav_*.npz       navigation mode        (G3) + g4 loss/grads (G4)
    g5_*.npz           object grounding, 3dqa (G5)
    g6_prompts.json    prompt strings         (G6)
    g7_graph.npz       graph_utils            (G7)
    g8_adamw.npz       clip + AdamW on bf16   (G8)
    g9_generate_*.npz  greedy generation through the reference's own generate() calls (3dqa free, summarization trie)
    g10_grads_*.npz    loss + parameter gradients of object_grounding / summarization / fgr2r / 3dqa TRAINING steps on the
                       G5 inputs (round 2: the backward of every mode, not only navigation)
    g11_fp8_*.npz      weight-only fp8: navigation logits of the reference run on de-quantised weights; a quantisation vector
                       (weights -> scales, e4m3fn codes, de-quantised values) and the 256-entry decode table, all from torch's float8_e4m3fn
    g1_encoder_real_F{1024,768}.npz   scene encoder at its real size (h=1024, 16 heads, ff=4096, 36 ragged views)
    g12_episode_*.npz  (round 3) a 3-step R2R EPISODE through the reference: panorama -> navigation -> CE -> backward() at every
                       step, the chosen slot's fuse_embeds appended to the history (mp3d_agent.py:683-778), gradients
                       ACCUMULATED over the steps (train.py:86-89 steps the optimizer only afterwards): per-step logits,
                       losses, history rows and the accumulated parameter gradients
    g14_train_*.npz    (round 5) the TRAINING-MODE path: the reference model in .train(), torch.nn.functional.dropout replaced
                       (from outside the reference tree) by a recorder that draws every keep mask from a seeded CPU generator and
                       stores it -- all dropout sites of a training step: drop_env on view / object features (nav_model.py:91,
                       99-102), the embedding dropout (image_embedding.py:73-74), and per encoder layer the ATTENTION-PROBABILITY
                       dropout of nn.MultiheadAttention (detr_transformer.py:138) + dropout1 / dropout / dropout2 (:141,146-147,
                       170-182): panorama (with objects) -> navigation -> CE -> backward, then object_grounding -> CE -> backward
    g13_optimizer_bf16.npz  (round 3) the `optimizer` entry of a reference checkpoint (torch.optim.AdamW.state_dict(), tools/optims.py:73)
                       after two optimizer steps (navigation, then object grounding) + the named_parameters() order its keys index

Three shims, all outside the reference tree (the third -- fp32 RoPE
frequencies, see build_reference -- undoes a transformers 4.28 -> 5.15 drift): the bert-large-uncased
config lookup is answered locally, and the LM is a from-config tiny Llama with a
locally trained SentencePiece tokenizer.  Weights are NOT the reference's random
init: they come from navillm_amd.params.synth_state_dict (seeded, per-name), are loaded
into the reference module with load_state_dict(strict=True), and are regenerated from
the seed by the tests -- so fixtures hold inputs + expected outputs only.

G9 needs two more version-drift shims, both on the LIBRARY side (install_generation_shims): the reference calls
`LlamaForCausalLM.prepare_inputs_for_generation(self, input_ids, past_key_values, attention_mask, inputs_embeds)`
positionally (modified_lm.py:187-193, the 4.28 signature) while 5.15 put `next_sequence_length` second; and it tests
`if not past_key_values` (modified_lm.py:194) for "this is the prefill step", where 4.28 passed None and 5.15 passes an
empty DynamicCache object.  The shims re-map the positionals to keywords and make an EMPTY cache falsy; nothing in the
reference tree is touched.
"""
import os, sys, json, types, logging, random, io
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from transformers import PretrainedConfig, BertConfig, LlamaConfig  # noqa: E402
from navillm_amd import config as nvcfg  # noqa: E402
from navillm_amd.params import param_specs, synth_state_dict  # noqa: E402

TINY_DIR = os.path.join(HERE, "tiny_llama")
_ENC = {}


def _install_bert_shim():
    orig = PretrainedConfig.from_pretrained.__func__

    def shim(cls, name, *a, **k):
        if name == "bert-large-uncased":
            return BertConfig(hidden_size=_ENC["h"], num_hidden_layers=24,
                              num_attention_heads=_ENC["heads"], intermediate_size=_ENC["ff"])
        return orig(cls, name, *a, **k)
    PretrainedConfig.from_pretrained = classmethod(shim)


def make_tiny_llama_dir(cfg):
    import sentencepiece as spm
    os.makedirs(TINY_DIR, exist_ok=True)
    random.seed(0)
    words = ("navigate following the instruction history which contains visual information of your "
             "previous decisions candidate several directions you can go to at current position is stop "
             "compare and infer progress then select correct direction from candidates target location output "
             "walk past table turn left right door kitchen stairs bedroom hallway exit enter wait near sofa "
             "chair window go straight up down answer question what color object exist select").split()
    corpus = os.path.join("/tmp", "nv_corpus.txt")
    with open(corpus, "w") as f:
        for _ in range(3000):
            f.write(" ".join(random.choices(words, k=random.randint(5, 20))) + "\n")
        for _ in range(50):
            f.write("### Instruction: Navigate following the instruction. ### History: (0) (1) (2) (3) (4) "
                    "### Candidate: (0) stop (1) (2) (5) (6) (7) (8) (9) ### Output: ### Answer: ### Question:\n")
    spm.SentencePieceTrainer.train(
        input=corpus, model_prefix=os.path.join(TINY_DIR, "tokenizer"), vocab_size=cfg.base_vocab_size,
        model_type="bpe", character_coverage=1.0, unk_id=0, bos_id=1, eos_id=2, pad_id=-1,
        minloglevel=2)
    os.remove(os.path.join(TINY_DIR, "tokenizer.vocab"))
    hf = LlamaConfig(vocab_size=cfg.base_vocab_size, hidden_size=cfg.hidden_size,
                     intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_layers,
                     num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_heads,
                     rms_norm_eps=cfg.rms_norm_eps, max_position_embeddings=2048)
    hf._attn_implementation = "eager"
    hf.save_pretrained(TINY_DIR)


def install_generation_shims():
    from transformers import LlamaForCausalLM
    from transformers.cache_utils import DynamicCache
    if getattr(LlamaForCausalLM, "_nv_shimmed", False):
        return
    orig = LlamaForCausalLM.prepare_inputs_for_generation

    def prep(self, input_ids, *args, **kw):
        if args and not isinstance(args[0], int):          # 4.28 calling convention
            for n, v in zip(["past_key_values", "attention_mask", "inputs_embeds"], args):
                kw[n] = v
            args = ()
        return orig(self, input_ids, *args, **kw)

    LlamaForCausalLM.prepare_inputs_for_generation = prep
    DynamicCache.__bool__ = lambda self: self.get_seq_length() > 0
    LlamaForCausalLM._nv_shimmed = True


def gen_generation(model, cfg, tag, seed=23):
    """G9: the reference's inference branches (nav_model.py:320-343, 386-404) run through ITS generate() calls;
    the ids HF returns are captured by wrapping lang_model.generate."""
    from tasks.agents.r2r import R2RAgent
    from tools.trie import Trie
    install_generation_shims()
    lm = model.lang_model
    g = torch.Generator().manual_seed(seed)
    B, N = 3, 4
    captured = []
    real_generate = lm.generate

    def spy(*a, **k):
        out = real_generate(*a, **k)
        captured.append((k["input_ids"].clone(), k["attention_mask"].clone(), out.clone()))
        return out

    lm.generate = spy
    try:
        with torch.no_grad():
            # ---- 3dqa, unconstrained, max_new_tokens=6 (llava.py:57-63 passes do_sample/temperature/max_new_tokens)
            feats = [torch.randn(n, cfg.image_feat_size, generator=g) for n in (5, 3, 4)]
            qprompts = [" ".join(["<cand>"] * f.shape[0]) + " ### Question: " + q + " ### Answer: "
                        for f, q in zip(feats, ("what color is the chair ?", "what is near the window ?", "exist table ?"))]
            qb = dict(question=qprompts, prompts=qprompts, answers=[["x"]] * B, features=feats, data_type=["scanqa"] * B)
            qo = model("3dqa", qb, training=False, do_sample=False, max_new_tokens=6)
            q_in, q_am, q_out = captured[-1]
            # ---- summarization with the trie constraint (mp3d_agent.py:541-581 builds it from candidate instructions)
            pin_s, _ = pano_inputs(cfg, g, B, N)
            ps = model("panorama", dict(pin_s))
            vp_img = torch.cat([torch.zeros_like(ps["pano_embeds"][:, :1]), ps["pano_embeds"]], 1)
            nav_masks = torch.cat([torch.ones(B, 1, dtype=torch.bool), pin_s["nav_types"] == 1], 1)
            hist_ts = [1, 0, 2]
            hvs = [[torch.randn(cfg.hidden_size, generator=g) for _ in range(hist_ts[b])] for b in range(B)]
            cn = nav_masks[:, 1:].sum(1)
            prompts_s = [R2RAgent.get_summarization_prompt(None, INSTR[b], hist_ts[b], int(cn[b])) for b in range(B)]
            tok = lm.tokenizer
            words = ["walk to the kitchen", "walk to the sofa and stop", "turn left at the door", "go up the stairs"]
            trie = Trie(tok.bos_token_id, tok.eos_token_id)
            word_ids = []
            for w in words:
                ids = tok(w, add_special_tokens=False)["input_ids"] + [tok.eos_token_id]
                trie.insert(ids)
                word_ids.append(ids)
            sb = dict(vp_img_embeds=vp_img.clone(), vp_pos_fts=torch.zeros(B, N + 1, 14), vp_nav_masks=nav_masks,
                      vp_cand_vpids=[[None]] * B, instruction=INSTR, answer=[""] * B, history=[["<hist>"] * t for t in hist_ts],
                      hist_vis=hvs, data_type=["r2r"] * B, prompts=prompts_s)
            so = model("summarization", sb, training=False, trie=trie)
            s_in, s_am, s_out = captured[-1]
    finally:
        lm.generate = real_generate
    L = max(len(w) for w in word_ids)
    words_arr = np.full((len(word_ids), L), -1, dtype=np.int64)
    for i, w in enumerate(word_ids):
        words_arr[i, :len(w)] = w
    hv_flat = torch.stack([v for vis in hvs for v in vis], 0)
    save(f"g9_generate_{tag}.npz", qa_features=pad(feats), qa_feat_lens=torch.tensor([f.shape[0] for f in feats]),
         qa_input_ids=q_in, qa_attention_mask=q_am, qa_new_ids=q_out[:, q_in.shape[1]:],
         **{"sum_" + k: v for k, v in pin_s.items() if torch.is_tensor(v)}, sum_vp_nav_masks=nav_masks, sum_hist_vis_flat=hv_flat,
         sum_input_ids=s_in, sum_attention_mask=s_am, sum_new_ids=s_out[:, s_in.shape[1]:], trie_words=words_arr,
         meta=np.array(json.dumps(dict(hist_t=hist_ts, qa_prompts=qprompts, sum_prompts=prompts_s, trie_words=words,
                                       qa_sentences=qo["generated_sentences"], sum_sentences=so["generated_sentences"],
                                       eos=tok.eos_token_id, pad=tok.unk_token_id, bos=tok.bos_token_id))))


def build_reference(cfg, seed):
    """Reference NavModel holding synth_state_dict(cfg, seed)."""
    import models.nav_model as nm
    _ENC.update(h=cfg.enc_hidden_size, heads=cfg.enc_num_heads, ff=cfg.enc_intermediate_size)
    args = types.SimpleNamespace(
        precision=cfg.precision, pretrained_model_name_or_path=TINY_DIR,
        image_feat_size=cfg.image_feat_size, angle_feat_size=cfg.angle_feat_size,
        obj_feat_size=cfg.obj_feat_size, resume_from_checkpoint=None, from_scratch=True,
        enable_og=cfg.enable_og, fuse_obj=cfg.fuse_obj, feat_dropout=cfg.feat_dropout)
    model = nm.NavModel(args, logging.getLogger("golden"), types.SimpleNamespace(num_pano_layers=cfg.num_pano_layers))
    model.lang_model.config._attn_implementation = "eager"
    ref_sd = model.state_dict()
    mine = synth_state_dict(cfg, seed)
    assert set(ref_sd) == set(mine), (sorted(set(ref_sd) ^ set(mine)))
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(mine[k].shape) and ref_sd[k].dtype == mine[k].dtype, \
            (k, ref_sd[k].shape, ref_sd[k].dtype, mine[k].shape, mine[k].dtype)
    model.load_state_dict(mine, strict=True)
    # Version-drift shim (third one): the reference pins transformers 4.28, whose
    # LlamaRotaryEmbedding caches cos/sin computed in fp32 at construction; `.to(bf16)`
    # (modified_lm.py:47) then rounds the TABLES.  The installed 5.15 instead keeps
    # `inv_freq` as a buffer, so the same `.to(bf16)` rounds the FREQUENCIES (a different,
    # position-dependent error).  Restore fp32 inv_freq so the fixture has the pinned
    # behaviour: cos/sin = lm_dtype(cos(fp32 angle)).
    rot = model.lang_model.model.rotary_emb
    hd = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    rot.inv_freq = torch.nn.Buffer(inv, persistent=False)
    rot.original_inv_freq = torch.nn.Buffer(inv.clone(), persistent=False)
    model.eval()
    return model


def npf(t):
    """tensor -> numpy (bf16 widened to fp32, exact)."""
    if isinstance(t, torch.Tensor):
        t = t.detach()
        if t.dtype == torch.bfloat16:
            t = t.float()
        return t.cpu().numpy()
    return np.asarray(t)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: npf(v) for k, v in arrs.items()})
    print(f"  wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


# ------------------------------------------------------------------ synthetic inputs
def pano_inputs(cfg, g, B, N, with_obj=False, O=5):
    lens = torch.tensor([N, N - 2, N - 1][:B] + [N] * max(0, B - 3))
    x = torch.randn(B, N, cfg.image_feat_size, generator=g)
    loc = torch.randn(B, N, 7, generator=g)
    nav = torch.zeros(B, N, dtype=torch.long)
    cand_k = [3, 2, 4][:B] + [3] * max(0, B - 3)
    for b in range(B):
        nav[b, :cand_k[b]] = 1
        x[b, lens[b]:] = 0
        loc[b, lens[b]:] = 0
    d = dict(view_img_fts=x, view_lens=lens, loc_fts=loc, nav_types=nav)
    if with_obj:
        ol = torch.tensor([O, O - 3, 1][:B] + [O] * max(0, B - 3))
        of = torch.randn(B, O, cfg.obj_feat_size, generator=g)
        olf = torch.randn(B, O, 7, generator=g)
        for b in range(B):
            of[b, ol[b]:] = 0
            olf[b, ol[b]:] = 0
        d.update(obj_img_fts=of, obj_lens=ol, obj_loc_fts=olf)
    return d, cand_k


def nav_inputs(cfg, g, pano_embeds, pano_masks, cand_k, hist_t):
    """A toy topological map per sample: slot 0 = stop, some visited nodes, frontier nodes;
    candidates of the current panorama point at a subset of the frontier (+ one visited)."""
    B, N, d = pano_embeds.shape
    vp_img = torch.cat([torch.zeros_like(pano_embeds[:, :1]), pano_embeds], 1)
    pm = torch.cat([torch.ones_like(pano_masks[:, :1]), pano_masks], 1)
    vp_pos = torch.randn(B, N + 1, 14, generator=g)
    gv, gvis, gstep, cand_vp = [], [], [], []
    for b in range(B):
        K = cand_k[b]
        visited = [f"v{b}_{i}" for i in range(1 + b)]
        frontier = [f"f{b}_{i}" for i in range(K + 1 - (1 if b == 1 else 0))]
        vpids = [None] + visited + frontier
        gv.append(vpids)
        gvis.append([0] + [1] * len(visited) + [0] * len(frontier))
        gstep.append([0] + list(range(1, len(visited) + 1)) + [0] * len(frontier))
        cands = frontier[:K - 1] + [visited[0]] if b != 2 else frontier[:K]
        cand_vp.append([None] + cands)
    G = max(len(v) for v in gv)
    gmask = torch.zeros(B, G, dtype=torch.bool)
    gvm = torch.zeros(B, G, dtype=torch.bool)
    gst = torch.zeros(B, G, dtype=torch.long)
    gimg = torch.randn(B, G, d, generator=g)
    gpos = torch.randn(B, G, 7, generator=g)
    for b in range(B):
        n = len(gv[b])
        gmask[b, :n] = True
        gvm[b, :n] = torch.tensor(gvis[b]).bool()
        gst[b, :n] = torch.tensor(gstep[b])
        gimg[b, 0] = 0
        gimg[b, n:] = 0
        gpos[b, n:] = 0
    hist_vis = [[torch.randn(d, generator=g) for _ in range(hist_t[b])] for b in range(B)]
    return dict(
        gmap_vpids=gv, gmap_img_embeds=gimg, gmap_step_ids=gst, gmap_pos_fts=gpos,
        gmap_visited_masks=gvm, gmap_masks=gmask,
        gmap_pair_dists=torch.zeros(B, G, G),
        vp_img_embeds=vp_img, pano_masks=pm, vp_pos_fts=vp_pos,
        vp_nav_masks=torch.ones(B, N + 1, dtype=torch.bool), vp_cand_vpids=cand_vp,
        hist_vis=hist_vis, history=[["<hist>"] * hist_t[b] for b in range(B)],
        data_type=["r2r"] * B,
    )


INSTR = ["walk past the table and turn left at the door then wait near the sofa",
         "go straight down the hallway and enter the bedroom",
         "exit the kitchen turn right and go up the stairs then stop near the window chair"]


def gen_precision(prec, seed=11):
    from tasks.agents.r2r import R2RAgent
    from tasks.agents.reverie import REVERIEAgent
    cfg = nvcfg.tiny(precision=prec)
    tag = "bf16" if cfg.lm_is_bf16 else "fp32"
    print(f"[{tag}] building reference model")
    model = build_reference(cfg, seed)
    lm = model.lang_model
    assert lm.cand_token_id == [cfg.cand_token_id] and lm.hist_token_id == [cfg.hist_token_id]
    assert lm.obj_token_id == [cfg.obj_token_id] and lm.cls_token_id == list(cfg.cls_token_ids)
    assert lm.tokenizer.pad_token_id == cfg.pad_token_id and len(lm.tokenizer) == cfg.vocab_size
    g = torch.Generator().manual_seed(1234)
    B, N = 3, 8

    # ---- G1 encoder (fp32 regardless of precision -> only emitted once)
    if tag == "fp32":
        pin, cand_k = pano_inputs(cfg, g, B, N, with_obj=True)
        with torch.no_grad():
            out = model("panorama", dict(pin))
            out_nopose = model.img_embeddings.forward_panorama_per_step(pin["view_img_fts"], pin["view_lens"])
        save("g1_encoder.npz", **pin, pano_embeds=out["pano_embeds"], pano_masks=out["pano_masks"],
             obj_embeds=out["obj_embeds"], obj_masks=out["obj_masks"],
             nopose_pano_embeds=out_nopose["pano_embeds"])
        # fuse_obj variant needs its own model (extra obj_linear params)
        cfg_f = nvcfg.tiny(precision=prec, fuse_obj=True)
        mf = build_reference(cfg_f, seed)
        with torch.no_grad():
            outf = mf("panorama", dict(pin))
        save("g1_encoder_fuseobj.npz", **pin, pano_embeds=outf["pano_embeds"], pano_masks=outf["pano_masks"],
             obj_embeds=outf["obj_embeds"])
        del mf

    # ---- G2 LM: left-padded ids with <cand>/<hist>, labels on the tail
    g2 = torch.Generator().manual_seed(77)
    S = 40
    ids = torch.randint(3, cfg.base_vocab_size, (B, S), generator=g2)
    am = torch.ones(B, S, dtype=torch.long)
    pads = [0, 7, 3]
    ncand, nhist = 0, 0
    for b in range(B):
        ids[b, :pads[b]] = cfg.pad_token_id
        am[b, :pads[b]] = 0
        for p_ in (12, 15 + b):
            ids[b, p_] = cfg.hist_token_id
            nhist += 1
        for p_ in (20, 22, 25 + b):
            ids[b, p_] = cfg.cand_token_id
            ncand += 1
        ids[b, S - 9] = cfg.cls_token_ids[0]
    cand_vis = torch.randn(ncand, cfg.hidden_size, generator=g2)
    hist_vis = torch.randn(nhist, cfg.hidden_size, generator=g2)
    labels = ids.clone()
    labels[:, :S - 8] = -100
    with torch.no_grad():
        o = lm(input_ids=ids, attention_mask=am, labels=labels, cand_vis=cand_vis, hist_vis=hist_vis)
    save(f"g2_lm_{tag}.npz", input_ids=ids, attention_mask=am, labels=labels, cand_vis=cand_vis,
         hist_vis=hist_vis, hidden_states=o.hidden_states, logits=o.logits[:, -10:], loss=o.loss)

    # ---- G3/G4 navigation + loss + grads
    pin, cand_k = pano_inputs(cfg, g, B, N)
    model.zero_grad()
    pano = model("panorama", dict(pin))
    hist_t = [2, 0, 3]
    nin = nav_inputs(cfg, g, pano["pano_embeds"], pano["pano_masks"], cand_k, hist_t)
    cand_nums = (nin["gmap_masks"] & ~nin["gmap_visited_masks"]).sum(-1)
    prompts = [R2RAgent.get_navigation_prompt(None, INSTR[b], hist_t[b], int(cand_nums[b]), lm.cls_token[0])
               for b in range(B)]
    nin["prompts"] = prompts
    nin["instruction"] = INSTR
    tok = lm.tokenize(prompts)
    torch.manual_seed(4321)
    perms = [torch.randperm(int(cand_nums[b]) - 1) for b in range(B)]
    torch.manual_seed(4321)
    nout = model("navigation", nin)
    targets = torch.tensor([2, -100, 4])   # unvisited map slots of samples 0 and 2; sample 1 ignored
    crit = torch.nn.CrossEntropyLoss(ignore_index=-100, reduction="sum")
    loss = crit(nout["fuse_logits"], targets) * 1.0 / B / 1
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    gnames = ["out_head.0.weight", "out_head.0.bias", "img_embeddings.img_linear.weight",
              "img_embeddings.mapper.weight", "vp_pos_embeddings.0.weight", "vp_pos_embeddings.1.weight",
              "gmap_pos_embeddings.0.weight", "gmap_step_embeddings.weight", "token_type_embeddings.weight",
              "lang_model.model.layers.0.self_attn.q_proj.weight", "lang_model.model.layers.1.mlp.down_proj.weight",
              "lang_model.model.layers.0.input_layernorm.weight", "lang_model.model.norm.weight",
              "img_embeddings.pano_encoder.layers.0.self_attn.in_proj_weight",
              "img_embeddings.pano_encoder.layers.1.linear1.weight", "img_embeddings.layer_norm.weight",
              "img_embeddings.loc_linear.weight", "img_embeddings.nav_type_embedding.weight"]
    gsave = {"grad/" + n: grads[n] for n in gnames}
    # embed_tokens grad is large-ish: keep row norms only
    gsave["gradnorm/lang_model.model.embed_tokens.weight"] = grads["lang_model.model.embed_tokens.weight"].float().norm(dim=1)
    gsave["grad_names_with_grad"] = np.array(sorted(grads.keys()))
    flat = {k: v for k, v in pin.items()}
    for k in ("gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_visited_masks", "gmap_masks",
              "vp_pos_fts"):
        flat[k] = nin[k]
    flat["nav_pano_masks"] = nin["pano_masks"]
    flat["hist_vis_flat"] = torch.stack([v for vis in nin["hist_vis"] for v in vis], 0)
    meta = dict(gmap_vpids=nin["gmap_vpids"], vp_cand_vpids=nin["vp_cand_vpids"], hist_t=hist_t, prompts=prompts,
                targets=targets.tolist(), perms=[p.tolist() for p in perms], seed_before_nav=4321)
    save(f"g3_nav_{tag}.npz", **flat, input_ids=tok["input_ids"], attention_mask=tok["attention_mask"],
         pano_embeds=pano["pano_embeds"], fuse_logits=nout["fuse_logits"], fuse_embeds=nout["fuse_embeds"],
         loss=loss, meta=np.array(json.dumps(meta)), **gsave)

    # ---- G11 (round 2): weight-only fp8.  The reference has no quantised path; what the fixture pins is
    #      "the reference model run on DE-QUANTISED weights": every decoder Linear weight replaced by dtype(s * e4m3fn(W / s)),
    #      s = rowmax|W| / 448, with torch's own float8_e4m3fn conversion as the external definition of the format.
    orig_sd = {k: v.clone() for k, v in model.state_dict().items()}
    fq = {}
    for k, v in orig_sd.items():
        if k.startswith("lang_model.model.layers.") and k.endswith("_proj.weight"):
            sc = v.float().abs().amax(1, keepdim=True) / 448.0
            sc = torch.where(sc > 0, sc, torch.ones_like(sc))
            fq[k] = ((v.float() / sc).to(torch.float8_e4m3fn).float() * sc).to(v.dtype)
        else:
            fq[k] = v
    model.load_state_dict(fq, strict=True)
    with torch.no_grad():
        torch.manual_seed(4321)
        nq = model("navigation", dict(nin, vp_img_embeds=nin["vp_img_embeds"].detach()))
    model.load_state_dict(orig_sd, strict=True)
    gq = torch.Generator().manual_seed(808)
    wq = (torch.randn(48, 128, generator=gq) * torch.logspace(-3, 1, 48)[:, None]).bfloat16()   # rows of very different scales
    wq[5] = 0                                                                                     # an all-zero row
    wq[7, 3] = 300.0                                                                              # an outlier: the rest of the row lands in the subnormals
    sq = wq.float().abs().amax(1, keepdim=True) / 448.0
    sq = torch.where(sq > 0, sq, torch.ones_like(sq))
    cq = (wq.float() / sq).to(torch.float8_e4m3fn)
    save(f"g11_fp8_{tag}.npz", fuse_logits=nq["fuse_logits"], quant_w=wq, quant_scales=sq[:, 0], quant_codes=cq.view(torch.uint8),
         quant_dequant=(cq.float() * sq).bfloat16(), decode_table=torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float())

    # ---- G5 object grounding + 3dqa training loss
    g5 = torch.Generator().manual_seed(99)
    pin_o, _ = pano_inputs(cfg, g5, B, N, with_obj=True)
    with torch.no_grad():
        po = model("panorama", dict(pin_o))
        hist_t5 = [1, 2, 0]
        hv = [[torch.randn(cfg.hidden_size, generator=g5) for _ in range(hist_t5[b])] for b in range(B)]
        ocn = po["obj_masks"].sum(1) + 1
        oprompts = [REVERIEAgent.get_object_grounding_prompt(None, INSTR[b], hist_t5[b], int(ocn[b]), lm.cls_token[0])
                    for b in range(B)]
        ob = dict(obj_embeds=po["obj_embeds"], obj_masks=po["obj_masks"], obj_loc_fts=po["obj_loc_fts"],
                  hist_vis=hv, history=[["<hist>"] * t for t in hist_t5], instruction=INSTR,
                  data_type=["reverie"] * B, prompts=oprompts)
        oo = model("object_grounding", ob)
        otok = lm.tokenize(oprompts)
    save(f"g5_og_{tag}.npz", **pin_o, input_ids=otok["input_ids"], attention_mask=otok["attention_mask"],
         hist_vis_flat=torch.stack([v for vis in hv for v in vis], 0), obj_embeds=po["obj_embeds"],
         obj_logits=oo["obj_logits"], meta=np.array(json.dumps(dict(hist_t=hist_t5, prompts=oprompts))))

    # ---- G10 (round 2): the SAME inputs with autograd on -> loss -> backward, as the rollout does on an episode's last step
    #      (mp3d_agent.py:788-842: panorama again, object_grounding, CE_sum * obj_loss_coef / B / accum, backward)
    crit_sum = torch.nn.CrossEntropyLoss(ignore_index=-100, reduction="sum")
    model.zero_grad()
    po_g = model("panorama", dict(pin_o))
    ob_g = dict(ob, obj_embeds=po_g["obj_embeds"], obj_masks=po_g["obj_masks"], obj_loc_fts=po_g["obj_loc_fts"])
    oo_g = model("object_grounding", ob_g)
    og_targets = torch.tensor([2, -100, 1])
    og_loss = crit_sum(oo_g["obj_logits"], og_targets) * 0.5 / B / 1
    og_loss.backward()
    g10 = {"og/" + k: v for k, v in grad_fixture(model, [n for n in G10_COMMON if not n.startswith(("out_head.0.bias", "img_embeddings.img_linear",
                                                              "img_embeddings.mapper", "img_embeddings.pano_encoder", "img_embeddings.layer_norm"))]
                                                 + ["out_head.0.bias"] + G10_OG).items()}
    g10["og/loss"] = og_loss
    g10["og/targets"] = og_targets

    # ---- G5s summarization / embodied_qa(fgr2r) training losses (nav_model.py:251-319; on the training path
    #      with --enable_summarize / --enable_fgr2r, mp3d_agent.py:845-909)
    with torch.no_grad():
        pin_s, cand_k_s = pano_inputs(cfg, g5, B, N)
        ps = model("panorama", dict(pin_s))
        vp_img = torch.cat([torch.zeros_like(ps["pano_embeds"][:, :1]), ps["pano_embeds"]], 1)
        nav_masks = torch.cat([torch.ones(B, 1, dtype=torch.bool), pin_s["nav_types"] == 1], 1)
        hist_ts = [2, 1, 0]
        hvs = [[torch.randn(cfg.hidden_size, generator=g5) for _ in range(hist_ts[b])] for b in range(B)]
        cn = nav_masks[:, 1:].sum(1)
        out_s = {}
        for mode, dtype_, answers in (("summarization", "r2r", ["", "", ""]),
                                      ("embodied_qa", "fgr2r", ["turn left at the door", "go up the stairs", "wait near the sofa"])):
            instr = INSTR if mode == "summarization" else ["where are we going with direction (1) ?"] * B
            hv_m = hvs if mode == "summarization" else [[] for _ in range(B)]
            hist_n = hist_ts if mode == "summarization" else [0] * B
            fn = R2RAgent.get_summarization_prompt if mode == "summarization" else R2RAgent.get_embodied_qa_prompt
            prompts_s = [fn(None, instr[b], hist_n[b], int(cn[b])) for b in range(B)]
            sb = dict(vp_img_embeds=vp_img.clone(), vp_pos_fts=torch.zeros(B, N + 1, 14), vp_nav_masks=nav_masks,
                      vp_cand_vpids=[[None]] * B, instruction=instr, answer=answers, history=[["<hist>"] * t for t in hist_n],
                      hist_vis=hv_m, data_type=[dtype_] * B, prompts=prompts_s)
            o = model(mode, sb, training=True)
            labels_txt = [(answers[b] if dtype_ in ("eqa", "fgr2r") else instr[b]) + lm.tokenizer.eos_token for b in range(B)]
            tk = lm.tokenize([[prompts_s[b], labels_txt[b]] for b in range(B)])
            out_s[mode] = (o["loss"], tk, prompts_s, labels_txt)
    # G10: summarization and fgr2r with autograd on (mp3d_agent.py:845-909: panorama -> mode -> loss * gen_loss_coef / B / accum
    # -> backward); the encoder receives gradient through vp_img_embeds
    for mode, dtype_, coef in (("summarization", "r2r", 1.0), ("embodied_qa", "fgr2r", 0.8)):
        model.zero_grad()
        ps_g = model("panorama", dict(pin_s))
        vp_g = torch.cat([torch.zeros_like(ps_g["pano_embeds"][:, :1]), ps_g["pano_embeds"]], 1)
        instr = INSTR if mode == "summarization" else ["where are we going with direction (1) ?"] * B
        answers = ["", "", ""] if mode == "summarization" else ["turn left at the door", "go up the stairs", "wait near the sofa"]
        hv_m = hvs if mode == "summarization" else [[] for _ in range(B)]
        hist_n = hist_ts if mode == "summarization" else [0] * B
        sb = dict(vp_img_embeds=vp_g, vp_pos_fts=torch.zeros(B, N + 1, 14), vp_nav_masks=nav_masks, vp_cand_vpids=[[None]] * B,
                  instruction=instr, answer=answers, history=[["<hist>"] * t for t in hist_n], hist_vis=hv_m,
                  data_type=[dtype_] * B, prompts=out_s[mode][2])
        o = model(mode, sb, training=True)
        l_ = o["loss"] * coef / B / 1
        l_.backward()
        key = "sum" if mode == "summarization" else "fgr2r"
        g10.update({f"{key}/" + k: v for k, v in grad_fixture(model, G10_COMMON[2:] + G10_LM).items()})
        g10[f"{key}/loss"] = l_
        g10[f"{key}/coef"] = np.array(coef)

    hv_flat = torch.stack([v for vis in hvs for v in vis], 0)
    save(f"g5_sum_{tag}.npz", **pin_s, vp_nav_masks=nav_masks, hist_vis_flat=hv_flat,
         sum_input_ids=out_s["summarization"][1]["input_ids"], sum_attention_mask=out_s["summarization"][1]["attention_mask"],
         sum_token_type_ids=out_s["summarization"][1]["token_type_ids"], sum_loss=out_s["summarization"][0],
         qa_input_ids=out_s["embodied_qa"][1]["input_ids"], qa_attention_mask=out_s["embodied_qa"][1]["attention_mask"],
         qa_token_type_ids=out_s["embodied_qa"][1]["token_type_ids"], qa_loss=out_s["embodied_qa"][0],
         meta=np.array(json.dumps(dict(hist_t=hist_ts, sum_prompts=out_s["summarization"][2], sum_labels=out_s["summarization"][3],
                                       qa_prompts=out_s["embodied_qa"][2], qa_labels=out_s["embodied_qa"][3]))))

    feats = [torch.randn(n, cfg.image_feat_size, generator=g5) for n in (6, 4, 5)]
    qprompts = ["### Question: what color is the chair ? ### Answer: ",
                "### Question: what is near the window ? ### Answer: ",
                "### Question: exist table ? ### Answer: "]
    # one <cand> per view row (llava.py:13-17 uses a single <cand> with one feature row; the model code
    # only requires #<cand> == #feature rows, nav_model.py:380-385)
    qprompts = [" ".join(["<cand>"] * f.shape[0]) + " " + q for f, q in zip(feats, qprompts)]
    answers = [["left door"], ["sofa"], ["kitchen table stop"]]
    qb = dict(question=qprompts, prompts=qprompts, answers=answers, features=feats, data_type=["scanqa"] * B)
    with torch.no_grad():
        qo = model("3dqa", qb, training=True)
    all_text = [[p, a[0] + lm.tokenizer.eos_token] for p, a in zip(qprompts, answers)]
    qtok = lm.tokenize(all_text)
    save(f"g5_qa_{tag}.npz", features=pad(feats), feat_lens=torch.tensor([f.shape[0] for f in feats]),
         input_ids=qtok["input_ids"], attention_mask=qtok["attention_mask"], token_type_ids=qtok["token_type_ids"],
         loss=qo.loss, meta=np.array(json.dumps(dict(prompts=qprompts, answers=answers))))
    # G10: 3dqa with autograd on (llava.py:38-42: loss * coef / accum -> backward)
    model.zero_grad()
    qo_g = model("3dqa", qb, training=True)
    q_l = qo_g.loss * 0.7 / 1
    q_l.backward()
    g10.update({"qa/" + k: v for k, v in grad_fixture(model, G10_COMMON[2:] + G10_LM).items()})
    g10["qa/loss"] = q_l
    g10["qa/coef"] = np.array(0.7)
    model.zero_grad()
    save(f"g10_grads_{tag}.npz", **g10)
    return model, cfg


def grad_fixture(model, names, big=20000):
    """selected parameter gradients after a backward(): small tensors whole, large 2-D ones as the strided sub-block
    [::3, ::5] plus their Frobenius norm; plus the sorted list of every parameter that has a gradient at all."""
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    out = {}
    for n in names:
        g = grads[n]
        if g.numel() > big and g.dim() == 2:
            out["gradsub/" + n] = g[::3, ::5]
            out["gradnorm/" + n] = g.float().norm()
        else:
            out["grad/" + n] = g
    out["rownorm/lang_model.model.embed_tokens.weight"] = grads["lang_model.model.embed_tokens.weight"].float().norm(dim=1)
    if "lang_model.lm_head.weight" in grads:
        out["rownorm/lang_model.lm_head.weight"] = grads["lang_model.lm_head.weight"].float().norm(dim=1)
    out["grad_names_with_grad"] = np.array(sorted(grads.keys()))
    return out


G10_COMMON = ["out_head.0.weight", "out_head.0.bias", "img_embeddings.img_linear.weight", "img_embeddings.mapper.weight",
              "img_embeddings.pano_encoder.layers.0.self_attn.in_proj_weight", "img_embeddings.layer_norm.weight",
              "lang_model.model.layers.0.self_attn.q_proj.weight", "lang_model.model.layers.1.mlp.down_proj.weight",
              "lang_model.model.layers.1.mlp.gate_proj.weight", "lang_model.model.layers.0.input_layernorm.weight",
              "lang_model.model.norm.weight"]
G10_LM = ["vp_pos_embeddings.0.bias", "vp_pos_embeddings.1.weight", "token_type_embeddings.weight"]
G10_OG = ["obj_pos_embeddings.0.weight", "obj_pos_embeddings.1.bias", "img_embeddings.obj_projector.0.weight",
          "img_embeddings.obj_projector.1.weight"]


G12_TARGETS = [[2, 3, 4], [0, -100, 5], [3, 4, 0]]      # per step: unvisited map slots / stop (0) / ended sample (-100)


def gen_episode(prec, seed=11, T_steps=3):
    """G12: one training episode of the rollout loop (mp3d_agent.py:659-778, teacher forcing) on fabricated inputs: per step
    `model('panorama')`, `model('navigation')` with the history the PREVIOUS steps produced (`hist_vis[b].append(fuse_embeds[b][a_t])`
    unless a_t == -100, :774-778), `criterion(logits, targets) * train_ml / B / accum` and an immediate `backward()` (:750-757);
    no zero_grad in between -- the fixture's gradients are what `optimizer.step()` would see (train.py:86-89)."""
    from tasks.agents.r2r import R2RAgent
    cfg = nvcfg.tiny(precision=prec)
    tag = "bf16" if cfg.lm_is_bf16 else "fp32"
    print(f"[{tag}] G12 episode")
    model = build_reference(cfg, seed)
    lm = model.lang_model
    g = torch.Generator().manual_seed(31337)
    B, N = 3, 8
    train_ml, accum = 0.8, 2
    crit = torch.nn.CrossEntropyLoss(ignore_index=-100, reduction="sum")
    hist_vis = [[] for _ in range(B)]
    model.zero_grad()
    arrs, steps_meta = {}, []
    for t in range(T_steps):
        pin, cand_k = pano_inputs(cfg, g, B, N)
        pano = model("panorama", dict(pin))
        hist_t = [len(h) for h in hist_vis]
        nin = nav_inputs(cfg, g, pano["pano_embeds"], pano["pano_masks"], cand_k, [0] * B)
        nin["hist_vis"] = [list(h) for h in hist_vis]
        nin["history"] = [["<hist>"] * n for n in hist_t]
        cand_nums = (nin["gmap_masks"] & ~nin["gmap_visited_masks"]).sum(-1)
        prompts = [R2RAgent.get_navigation_prompt(None, INSTR[b], hist_t[b], int(cand_nums[b]), lm.cls_token[0]) for b in range(B)]
        nin["prompts"] = prompts
        nin["instruction"] = INSTR
        tok = lm.tokenize(prompts)
        torch.manual_seed(5000 + t)
        perms = [torch.randperm(int(cand_nums[b]) - 1) for b in range(B)]
        torch.manual_seed(5000 + t)
        nout = model("navigation", nin)
        targets = torch.tensor(G12_TARGETS[t])
        loss = crit(nout["fuse_logits"], targets) * train_ml / B / accum
        loss.backward()
        for b in range(B):
            if int(targets[b]) != -100:
                hist_vis[b].append(nout["fuse_embeds"][b][int(targets[b])])
        # static prefix of each prompt (everything up to "### History:") in token ids: the longest common prefix of the two
        # tokenisations, so that a sub-word merge across the cut cannot matter
        plens = []
        for b in range(B):
            full = lm.tokenizer(prompts[b], add_special_tokens=True)["input_ids"]
            cut = prompts[b].index("### History:") + len("### History:")
            head = lm.tokenizer(prompts[b][:cut], add_special_tokens=True)["input_ids"]
            n = 0
            while n < min(len(full), len(head)) and full[n] == head[n]:
                n += 1
            plens.append(n)
        pre = f"s{t}/"
        for k, v in pin.items():
            arrs[pre + k] = v
        for k in ("gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_visited_masks", "gmap_masks", "vp_pos_fts"):
            arrs[pre + k] = nin[k]
        arrs[pre + "nav_pano_masks"] = nin["pano_masks"]
        arrs[pre + "input_ids"] = tok["input_ids"]
        arrs[pre + "attention_mask"] = tok["attention_mask"]
        arrs[pre + "pano_embeds"] = pano["pano_embeds"]
        arrs[pre + "fuse_logits"] = nout["fuse_logits"]
        arrs[pre + "fuse_embeds"] = nout["fuse_embeds"]
        arrs[pre + "loss"] = loss
        steps_meta.append(dict(gmap_vpids=nin["gmap_vpids"], vp_cand_vpids=nin["vp_cand_vpids"], hist_t=hist_t, prompts=prompts,
                               targets=targets.tolist(), perms=[p_.tolist() for p_ in perms], seed_before_nav=5000 + t,
                               prefix_lens=plens))
    gnames = ["out_head.0.weight", "out_head.0.bias", "img_embeddings.img_linear.weight",
              "img_embeddings.mapper.weight", "vp_pos_embeddings.0.weight", "vp_pos_embeddings.1.weight",
              "gmap_pos_embeddings.0.weight", "gmap_step_embeddings.weight", "token_type_embeddings.weight",
              "lang_model.model.layers.0.self_attn.q_proj.weight", "lang_model.model.layers.0.self_attn.k_proj.weight",
              "lang_model.model.layers.0.self_attn.v_proj.weight", "lang_model.model.layers.1.self_attn.o_proj.weight",
              "lang_model.model.layers.1.mlp.down_proj.weight", "lang_model.model.layers.0.mlp.up_proj.weight",
              "lang_model.model.layers.0.input_layernorm.weight", "lang_model.model.layers.1.post_attention_layernorm.weight",
              "lang_model.model.norm.weight", "img_embeddings.pano_encoder.layers.0.self_attn.in_proj_weight",
              "img_embeddings.pano_encoder.layers.1.linear1.weight", "img_embeddings.layer_norm.weight",
              "img_embeddings.loc_linear.weight", "img_embeddings.nav_type_embedding.weight"]
    arrs.update({"acc/" + k: v for k, v in grad_fixture(model, gnames).items()})
    arrs["hist_final_flat"] = torch.stack([v for vis in hist_vis for v in vis], 0)
    meta = dict(steps=steps_meta, train_ml=train_ml, accum=accum, B=B, hist_final=[len(h) for h in hist_vis])
    save(f"g12_episode_{tag}.npz", **arrs, meta=np.array(json.dumps(meta)))


class DropoutRecorder:
    """stand-in for torch.nn.functional.dropout while the reference runs in .train(): same arithmetic as torch's
    (input * (keep / (1 - p))), the keep flags drawn from a seeded CPU generator and recorded in call order."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.rec = []

    def __call__(self, input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        keep = torch.rand(input.shape, generator=self.g) >= p
        self.rec.append((float(p), keep))
        return input * (keep.to(input.dtype) / (1.0 - p))


def gen_train_mode(prec, seed=11):
    """G14 (round 5): one TRAINING-mode step of the rollout's calls (mp3d_agent.py:683,726,750-757,791-824): `model.train()`,
    panorama with objects -> navigation -> CE * w / B -> backward(), then object_grounding on the same panorama's objects ->
    CE * w / B -> backward() (gradients accumulate).  Every dropout mask the reference consumed is stored batch-first under the
    key names navillm_amd.NavModel.injected_dropout / the oracle's `dmasks` use."""
    import torch.nn.functional as F_
    from tasks.agents.r2r import R2RAgent
    from tasks.agents.reverie import REVERIEAgent
    cfg = nvcfg.tiny(precision=prec)
    tag = "bf16" if cfg.lm_is_bf16 else "fp32"
    print(f"[{tag}] G14 training-mode step")
    model = build_reference(cfg, seed)
    model.train()
    lm = model.lang_model
    g = torch.Generator().manual_seed(1414)
    B, N = 3, 8
    heads, h, ff = cfg.enc_num_heads, cfg.enc_hidden_size, cfg.enc_intermediate_size
    crit = torch.nn.CrossEntropyLoss(ignore_index=-100, reduction="sum")
    pin, cand_k = pano_inputs(cfg, g, B, N, with_obj=True)
    O = pin["obj_img_fts"].shape[1]
    rec = DropoutRecorder(9090)
    real_dropout = F_.dropout
    F_.dropout = rec
    try:
        model.zero_grad()
        pano = model("panorama", dict(pin))
        n_pano_calls = len(rec.rec)
        hist_t = [1, 0, 2]
        nin = nav_inputs(cfg, g, pano["pano_embeds"], pano["pano_masks"], cand_k, hist_t)
        cand_nums = (nin["gmap_masks"] & ~nin["gmap_visited_masks"]).sum(-1)
        prompts = [R2RAgent.get_navigation_prompt(None, INSTR[b], hist_t[b], int(cand_nums[b]), lm.cls_token[0]) for b in range(B)]
        nin["prompts"] = prompts
        nin["instruction"] = INSTR
        tok = lm.tokenize(prompts)
        torch.manual_seed(1415)
        perms = [torch.randperm(int(cand_nums[b]) - 1) for b in range(B)]
        torch.manual_seed(1415)
        nout = model("navigation", nin)
        targets = torch.tensor([3, 0, -100])
        loss = crit(nout["fuse_logits"], targets) * 0.7 / B / 1
        loss.backward(retain_graph=True)
        nav_grads = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in grad_fixture(model, G14_NAV).items()}   # .grad keeps accumulating below
        # object grounding on the same panorama's object tokens (the rollout calls panorama again at :791; one panorama keeps
        # the mask bookkeeping to ONE set of encoder masks and still sends a second gradient through drop_env.obj)
        ocn = pano["obj_masks"].sum(1) + 1
        oprompts = [REVERIEAgent.get_object_grounding_prompt(None, INSTR[b], hist_t[b], int(ocn[b]), lm.cls_token[0]) for b in range(B)]
        ob = dict(obj_embeds=pano["obj_embeds"], obj_masks=pano["obj_masks"], obj_loc_fts=pano["obj_loc_fts"],
                  hist_vis=nin["hist_vis"], history=nin["history"], instruction=INSTR, data_type=["reverie"] * B, prompts=oprompts)
        oo = model("object_grounding", ob)
        otok = lm.tokenize(oprompts)
        og_targets = torch.tensor([1, 2, -100])
        og_loss = crit(oo["obj_logits"], og_targets) * 0.5 / B / 1
        og_loss.backward()
        acc_grads = grad_fixture(model, G14_NAV + G10_OG)
    finally:
        F_.dropout = real_dropout
    # ---- the masks, in the order the reference drew them, renamed and laid out batch-first
    assert len(rec.rec) == n_pano_calls, "a dropout fired outside the panorama call"
    names = ["drop_env.view", "drop_env.obj", "emb.drop"]
    for i in range(cfg.num_pano_layers):
        names += [f"l{i}.attn", f"l{i}.drop1", f"l{i}.drop", f"l{i}.drop2"]
    assert len(rec.rec) == len(names), (len(rec.rec), names)
    masks = {}
    for nm, (p_, keep) in zip(names, rec.rec):
        if nm.startswith("drop_env"):
            assert abs(p_ - cfg.feat_dropout) < 1e-9
        else:
            assert abs(p_ - cfg.enc_dropout) < 1e-9
        if nm.endswith(".attn"):
            assert tuple(keep.shape) == (B * heads, N, N), keep.shape      # torch: index = b * heads + head
            keep = keep.view(B, heads, N, N)
        elif nm.startswith("l"):
            assert tuple(keep.shape) in ((N, B, h), (N, B, ff)), keep.shape  # the encoder runs sequence-first (detr_transformer.py:76-77)
            keep = keep.transpose(0, 1).contiguous()
        masks["mask/" + nm] = keep.to(torch.uint8)
    flat = {k: v for k, v in pin.items()}
    for k in ("gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_visited_masks", "gmap_masks", "vp_pos_fts"):
        flat[k] = nin[k]
    flat["nav_pano_masks"] = nin["pano_masks"]
    flat["hist_vis_flat"] = torch.stack([v for vis in nin["hist_vis"] for v in vis], 0)
    meta = dict(gmap_vpids=nin["gmap_vpids"], vp_cand_vpids=nin["vp_cand_vpids"], hist_t=hist_t, prompts=prompts, og_prompts=oprompts,
                targets=targets.tolist(), og_targets=og_targets.tolist(), perms=[p_.tolist() for p_ in perms], seed_before_nav=1415,
                nav_coef=0.7, og_coef=0.5, mask_order=names)
    save(f"g14_train_{tag}.npz", **flat, **masks, input_ids=tok["input_ids"], attention_mask=tok["attention_mask"],
         og_input_ids=otok["input_ids"], og_attention_mask=otok["attention_mask"],
         pano_embeds=pano["pano_embeds"], obj_embeds=pano["obj_embeds"], fuse_logits=nout["fuse_logits"],
         fuse_embeds=nout["fuse_embeds"], loss=loss, obj_logits=oo["obj_logits"], og_loss=og_loss,
         **{"nav/" + k: v for k, v in nav_grads.items()}, **{"acc/" + k: v for k, v in acc_grads.items()},
         meta=np.array(json.dumps(meta)))


G14_NAV = ["out_head.0.weight", "out_head.0.bias", "img_embeddings.img_linear.weight", "img_embeddings.img_linear.bias",
           "img_embeddings.mapper.weight", "vp_pos_embeddings.0.weight", "vp_pos_embeddings.1.weight",
           "gmap_pos_embeddings.0.weight", "gmap_step_embeddings.weight", "token_type_embeddings.weight",
           "lang_model.model.layers.0.self_attn.q_proj.weight", "lang_model.model.layers.1.mlp.down_proj.weight",
           "lang_model.model.layers.0.input_layernorm.weight", "lang_model.model.norm.weight",
           "img_embeddings.pano_encoder.layers.0.self_attn.in_proj_weight", "img_embeddings.pano_encoder.layers.0.self_attn.in_proj_bias",
           "img_embeddings.pano_encoder.layers.0.self_attn.out_proj.weight", "img_embeddings.pano_encoder.layers.1.self_attn.in_proj_weight",
           "img_embeddings.pano_encoder.layers.0.linear2.weight", "img_embeddings.pano_encoder.layers.1.linear1.weight",
           "img_embeddings.pano_encoder.layers.1.norm1.weight", "img_embeddings.pano_encoder.norm.weight",
           "img_embeddings.layer_norm.weight", "img_embeddings.loc_linear.weight", "img_embeddings.nav_type_embedding.weight"]


def gen_optimizer_state(seed=11):
    """G13: the `optimizer` entry of a reference checkpoint (tools/optims.py:65-78) after two optimizer steps of the training loop
    (train.py:86-89: clip_grad_norm_(40) + AdamW.step + zero_grad) on the tiny amp_bf16 model: step 1 after a navigation
    backward, step 2 after an object-grounding backward, so `obj_pos_embeddings` / `obj_projector` enter the state one step late
    and `og_head` / `lm_head` never do (no gradient: optimizer.step skips them).  The optimizer is built as tools/optims.py:43
    builds it.  zero_grad(set_to_none=False) = the pinned torch 1.10's zero_grad."""
    from tasks.agents.r2r import R2RAgent
    from tasks.agents.reverie import REVERIEAgent
    cfg = nvcfg.tiny(precision="amp_bf16")
    print("[bf16] G13 optimizer state")
    model = build_reference(cfg, seed)
    lm = model.lang_model
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert names == [n for n, _, _ in param_specs(cfg)], "params.param_specs is not in named_parameters() order"
    opt = torch.optim.AdamW([p for n, p in model.named_parameters() if p.requires_grad], lr=3e-5)
    crit = torch.nn.CrossEntropyLoss(ignore_index=-100, reduction="sum")
    g = torch.Generator().manual_seed(4242)
    B, N = 3, 8
    # optimizer step 1: one navigation step
    pin, cand_k = pano_inputs(cfg, g, B, N)
    pano = model("panorama", dict(pin))
    nin = nav_inputs(cfg, g, pano["pano_embeds"], pano["pano_masks"], cand_k, [0] * B)
    nin["hist_vis"] = [[] for _ in range(B)]
    nin["history"] = [[] for _ in range(B)]
    cand_nums = (nin["gmap_masks"] & ~nin["gmap_visited_masks"]).sum(-1)
    nin["prompts"] = [R2RAgent.get_navigation_prompt(None, INSTR[b], 0, int(cand_nums[b]), lm.cls_token[0]) for b in range(B)]
    nin["instruction"] = INSTR
    torch.manual_seed(6000)
    nout = model("navigation", nin)
    (crit(nout["fuse_logits"], torch.tensor(G12_TARGETS[0])) * 0.8 / B).backward()
    n1 = torch.nn.utils.clip_grad_norm_(model.parameters(), 40.)
    opt.step()
    opt.zero_grad(set_to_none=False)
    # optimizer step 2: one object-grounding step (mp3d_agent.py:788-842)
    pin_o, _ = pano_inputs(cfg, g, B, N, with_obj=True)
    po = model("panorama", dict(pin_o))
    ocn = po["obj_masks"].sum(1) + 1
    ob = dict(obj_embeds=po["obj_embeds"], obj_masks=po["obj_masks"], obj_loc_fts=po["obj_loc_fts"],
              hist_vis=[[] for _ in range(B)], history=[[] for _ in range(B)], instruction=INSTR, data_type=["reverie"] * B,
              prompts=[REVERIEAgent.get_object_grounding_prompt(None, INSTR[b], 0, int(ocn[b]), lm.cls_token[0]) for b in range(B)])
    oo = model("object_grounding", ob)
    (crit(oo["obj_logits"], torch.tensor([2, -100, 1])) * 0.5 / B).backward()
    n2 = torch.nn.utils.clip_grad_norm_(model.parameters(), 40.)
    opt.step()
    sd = opt.state_dict()
    arrs = {}
    steps = {}
    for k, st in sd["state"].items():
        steps[str(k)] = float(st["step"])
        for key in ("exp_avg", "exp_avg_sq"):
            # the moments are payload here (G8 pins the arithmetic): big matrices keep their [::3, ::5] sub-block only
            t = st[key]
            if t.numel() > 20000:
                arrs[f"{key}_sub/{k}"] = t[::3, ::5]
            else:
                arrs[f"{key}/{k}"] = t
    grp = {k: (list(v) if isinstance(v, tuple) else v) for k, v in sd["param_groups"][0].items()}
    meta = dict(names=names, steps=steps, param_group=grp, grad_norms=[float(n1), float(n2)],
                shapes={str(k): list(st["exp_avg"].shape) for k, st in sd["state"].items()},
                moment_dtypes={str(k): str(st["exp_avg"].dtype) for k, st in sd["state"].items()})
    save("g13_optimizer_bf16.npz", **arrs, meta=np.array(json.dumps(meta)))


def pad(ts):
    m = max(t.shape[0] for t in ts)
    return torch.stack([torch.cat([t, torch.zeros(m - t.shape[0], *t.shape[1:])], 0) for t in ts], 0)


def gen_encoder_real_size(seed=11):
    """G1 at the REAL scene-encoder size: h=1024, 16 heads x 64, ff=4096, 2 layers, 36 views, ragged view
    counts, F=1024 (EVA-CLIP-L, with objects) and F=768 (the BASELINE synthetic width).  Only the mapper's output width is
    tiny (d=256, the fixture LM) -- it is a plain Linear(1024 -> d) -- so the fixture stays < 1 MB."""
    for F_ in (1024, 768):
        cfg = nvcfg.tiny(precision="fp32", enc_hidden_size=1024, enc_num_heads=16, enc_intermediate_size=4096, image_feat_size=F_,
                         obj_feat_size=768)
        model = build_reference(cfg, seed)
        g = torch.Generator().manual_seed(4242 + F_)
        B, N = 3, 36
        x = torch.randn(B, N, F_, generator=g)
        lens = torch.tensor([36, 29, 33])
        loc = torch.randn(B, N, 7, generator=g)
        nav = torch.zeros(B, N, dtype=torch.long)
        for b, k in enumerate((5, 2, 8)):
            nav[b, :k] = 1
            x[b, lens[b]:] = 0
            loc[b, lens[b]:] = 0
        pin = dict(view_img_fts=x, view_lens=lens, loc_fts=loc, nav_types=nav)
        if F_ == 1024:
            O = 6
            ol = torch.tensor([6, 2, 4])
            of = torch.randn(B, O, 768, generator=g)
            olf = torch.randn(B, O, 7, generator=g)
            for b in range(B):
                of[b, ol[b]:] = 0
                olf[b, ol[b]:] = 0
            pin.update(obj_img_fts=of, obj_lens=ol, obj_loc_fts=olf)
        with torch.no_grad():
            out = model("panorama", dict(pin))
        extra = dict(obj_embeds=out["obj_embeds"], obj_masks=out["obj_masks"]) if F_ == 1024 else {}
        save(f"g1_encoder_real_F{F_}.npz", **pin, pano_embeds=out["pano_embeds"], pano_masks=out["pano_masks"], **extra)
        del model


def gen_prompts():
    """G6: byte-exact prompt strings of every agent/mode the synthetic driver uses."""
    from tasks.agents.r2r import R2RAgent
    from tasks.agents.reverie import REVERIEAgent
    from tasks.agents.soon import SOONAgent
    from tasks.agents.cvdn import CVDNAgent
    out = {}
    for cls in (R2RAgent, REVERIEAgent, SOONAgent, CVDNAgent):
        for (h, c) in ((0, 1), (2, 4), (5, 3)):
            for mode in ("navigation", "object_grounding", "summarization", "embodied_qa"):
                fn = getattr(cls, f"get_{mode}_prompt", None)
                if fn is None:
                    continue
                try:
                    if mode in ("navigation", "object_grounding"):
                        s = fn(None, "INSTR", h, c, "<cls_1>")
                    else:
                        s = fn(None, "INSTR", h, c)
                except Exception as e:  # agent does not implement it this way
                    continue
                out[f"{cls.name}/{mode}/{h}/{c}"] = s
    with open(os.path.join(HERE, "g6_prompts.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"  wrote g6_prompts.json: {len(out)} prompts")


def gen_graph():
    """G7: models/graph_utils.py on a toy graph."""
    from models.graph_utils import calculate_vp_rel_pos_fts, get_angle_fts, FloydGraph, GraphMap
    rng = np.random.RandomState(5)
    pos = {f"n{i}": rng.randn(3).astype(np.float64) * 3 for i in range(7)}
    edges = [(0, 1), (1, 2), (2, 3), (1, 4), (4, 5), (5, 6), (3, 6)]
    fg = FloydGraph()
    dists_after = []
    for a, b in edges:
        d = float(np.linalg.norm(pos[f"n{a}"] - pos[f"n{b}"]))
        fg.add_edge(f"n{a}", f"n{b}", d)
    for k in ("n1", "n2", "n4", "n5"):
        fg.update(k)
        dists_after.append([[fg.distance(f"n{i}", f"n{j}") for j in range(7)] for i in range(7)])
    paths = {f"{i}-{j}": fg.path(f"n{i}", f"n{j}") for i in (0, 1, 2) for j in (3, 5, 6) if i != j}
    rel = np.stack([calculate_vp_rel_pos_fts(pos["n0"], pos[f"n{j}"], base_heading=0.3, base_elevation=-0.1)
                    for j in range(1, 7)])
    ang = get_angle_fts(rel[:, 0], rel[:, 1], 4)
    gm = GraphMap("n0")
    gm.node_positions = dict(pos)
    gm.graph = fg
    pf = gm.get_pos_fts("n1", [None, "n0", "n2", "n3", "n5"], 0.3, -0.1)
    dd = np.array(dists_after, dtype=np.float64)
    dd[~np.isfinite(dd)] = -1.0
    save("g7_graph.npz", positions=np.stack([pos[f"n{i}"] for i in range(7)]), edges=np.array(edges),
         dists_after=dd, rel=rel, ang=ang, pos_fts=pf, meta=np.array(json.dumps(dict(paths=paths))))


def gen_adamw():
    """G8: 3 steps of clip_grad_norm_(40) + torch.optim.AdamW on bf16 and fp32 tensors."""
    g = torch.Generator().manual_seed(3)
    p0 = [torch.randn(300, 64, generator=g).mul(0.05).bfloat16(), torch.randn(1000, generator=g).bfloat16(),
          torch.randn(77, 33, generator=g).mul(0.1)]
    params = [torch.nn.Parameter(p.clone()) for p in p0]
    opt = torch.optim.AdamW(params, lr=1e-3)
    grads, snaps, norms = [], [], []
    for step in range(3):
        gs = [(torch.randn(p.shape, generator=g) * (30.0 if step == 1 else 0.3)).to(p.dtype) for p in params]
        grads.append(gs)
        for p, gr in zip(params, gs):
            p.grad = gr.clone()
        norms.append(torch.nn.utils.clip_grad_norm_(params, 40.0))
        opt.step()
        snaps.append([p.detach().clone() for p in params])
    arr = {}
    for i in range(3):
        arr[f"p0_{i}"] = p0[i]
        for s in range(3):
            arr[f"g{s}_{i}"] = grads[s][i]
            arr[f"p{s + 1}_{i}"] = snaps[s][i]
    arr["norms"] = torch.stack([n.float() for n in norms])
    save("g8_adamw.npz", **arr)


def main():
    torch.set_num_threads(8)
    torch.use_deterministic_algorithms(False)
    _install_bert_shim()
    cfg = nvcfg.tiny()
    make_tiny_llama_dir(cfg)
    meta = {}
    if "--only-encoder-real" in sys.argv:        # add the real-size G1 without touching the other fixtures
        gen_encoder_real_size()
        return
    if "--only-episode" in sys.argv:             # add G12 without touching the other fixtures
        for prec in ("fp32", "amp_bf16"):
            gen_episode(prec)
        return
    if "--only-train-mode" in sys.argv:          # add G14 without touching the other fixtures
        for prec in ("fp32", "amp_bf16"):
            gen_train_mode(prec)
        return
    if "--only-optimizer" in sys.argv:           # add G13 without touching the other fixtures
        gen_optimizer_state()
        return
    if "--only-generation" in sys.argv:          # add G9 without touching the other fixtures
        for prec in ("fp32", "amp_bf16"):
            c = nvcfg.tiny(precision=prec)
            gen_generation(build_reference(c, 11), c, "bf16" if c.lm_is_bf16 else "fp32")
        return
    for prec in ("fp32", "amp_bf16"):
        model, c = gen_precision(prec)
        gen_generation(model, c, "bf16" if c.lm_is_bf16 else "fp32")
        meta[prec] = {k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}
    # also record the key inventory for the configs with fuse_obj / no objects
    with open(os.path.join(HERE, "g_meta.json"), "w") as f:
        json.dump(meta, f, indent=0, sort_keys=True)
    gen_encoder_real_size()
    gen_prompts()
    gen_graph()
    gen_adamw()
    for prec in ("fp32", "amp_bf16"):
        gen_episode(prec)
    gen_optimizer_state()
    for prec in ("fp32", "amp_bf16"):
        gen_train_mode(prec)


if __name__ == "__main__":
    main()
