"""SURVEY.md §8f items 1-2 on the GPU: prompt-prefix K/V reuse across navigation steps and greedy generation, each
against full recompute (the HIP training-path forward) and against the CPU oracle."""
import pytest
import torch

from util import load_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Node:
    def __init__(self):
        self.child = {}


class _Trie:
    """tools/trie.py protocol (root / get_child_index / get_next_node) without its defaultdict side effects"""

    def __init__(self, eos):
        self.root, self.eos = _Node(), eos

    def insert(self, word):
        cur = self.root
        for c in word:
            cur = cur.child.setdefault(c, _Node())

    def get_child_index(self, cur):
        return [self.eos] if not cur.child else list(cur.child.keys())

    def get_next_node(self, cur, w):
        if not cur.child:
            return cur
        return cur.child.setdefault(w, _Node())


def _mid_cfg(layers=2):
    from navillm_amd import config as nvcfg
    return nvcfg.NavConfig(hidden_size=512, num_layers=layers, num_heads=4, intermediate_size=1408, base_vocab_size=1000,
                           enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)


def test_prefix_kv_reuse_matches_full_recompute_over_an_episode():
    """6 no-grad navigation steps of 3 lock-step episodes, once through the K/V cache and once through the full
    forward: same logits (bf16 noise), same argmax wherever the margin allows, and from step 1 on only the prompt
    suffix is recomputed."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.synthetic import SyntheticEpisodes
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=9)
    m.eval()
    B = 3
    eps = [SyntheticEpisodes(cfg, B, seed=31, instr_len=300, device=torch.device(DEV)) for _ in range(2)]
    m.enable_kv_cache(B, capacity=1024)
    worst = 0.0
    for step in range(6):
        logits = []
        for use_cache, ep in ((True, eps[0]), (False, eps[1])):
            kv, m.kv = m.kv, (m.kv if use_cache else None)
            with torch.no_grad():
                pin = ep.panorama_inputs()
                pano = m("panorama", pin)
                ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
                nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
                nav["input_ids"], nav["attention_mask"] = ep.tokenise(nav, "<cls_1>")
                torch.manual_seed(500 + step)
                out = m("navigation", nav)
            m.kv = kv
            logits.append(out["fuse_logits"].float().cpu())
            tg = ep.teacher_targets(nav, last=False)
            ep.advance(nav, tg, out["fuse_embeds"])
        a, b = logits
        fin = torch.isfinite(b)
        assert torch.equal(torch.isfinite(a), fin)
        gap = (a[fin] - b[fin]).abs().max().item()
        worst = max(worst, gap)
        st = m.kv.last_stats
        print(f"[kv step {step}] S={nav['input_ids'].shape[1]} prefix={st['prefix']} new={st['new']} |cached-full|={gap:.4f}")
        assert gap < 0.06
        if step > 0:
            assert min(st["prefix"]) > 300 and max(st["new"]) < 120, st     # the instruction + history prefix was reused
        top2 = b.masked_fill(~fin, -1e9).topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 4 * gap + 1e-3
        assert torch.equal(a.argmax(1)[safe], b.argmax(1)[safe])
    print(f"[kv] worst |cached-full| over the episode: {worst:.4f}")


def _gen_case(cfg, B, seed):
    """prompts of different lengths with <cand> and <hist> tokens + their visual rows"""
    g = torch.Generator().manual_seed(seed)
    ids_l, n_c, n_h = [], 0, 0
    for b in range(B):
        L = 40 + 17 * b
        ids = torch.randint(3, cfg.base_vocab_size, (L,), generator=g).tolist()
        ids[0] = 1
        for j in (5, 9 + b):
            ids[j] = cfg.cand_token_id
            n_c += 1
        ids[20] = cfg.hist_token_id
        n_h += 1
        ids_l.append(ids)
    S = max(len(x) for x in ids_l)
    ids_t = torch.full((B, S), cfg.pad_token_id, dtype=torch.int64)
    am = torch.zeros((B, S), dtype=torch.int64)
    for b, x in enumerate(ids_l):
        ids_t[b, S - len(x):] = torch.tensor(x)
        am[b, S - len(x):] = 1
    cand = torch.randn(n_c, cfg.hidden_size, generator=g) * 0.5
    hist = torch.randn(n_h, cfg.hidden_size, generator=g) * 0.5
    return ids_t, am, cand, hist


@pytest.mark.parametrize("with_trie", [False, True])
def test_greedy_generation_matches_oracle_recompute(with_trie):
    """K/V-cache greedy decoding (HIP) vs the oracle's cache-free recompute: identical token sequences up to the first
    step whose top-2 margin is inside the bf16 noise; eos / pad bookkeeping and the trie constraint included."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.params import synth_state_dict
    from navillm_amd.kvcache import KVCacheLM
    O = load_oracle()
    cfg = _mid_cfg(layers=2)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=5)
    m.eval()
    P16 = synth_state_dict(cfg, 5)
    with torch.no_grad():
        assert m.load_reference_state_dict(P16) == len(P16)
    B = 3
    ids_t, am, cand, hist = _gen_case(cfg, B, 123)
    eos, pad = 2, 0
    trie = None
    if with_trie:
        trie = _Trie(eos)
        g = torch.Generator().manual_seed(7)
        for _ in range(40):
            trie.insert(torch.randint(3, cfg.base_vocab_size, (6,), generator=g).tolist() + [eos])
    max_new = 10
    ref_ids, _ = O.greedy_generate(P16, cfg, ids_t, am, cand_vis=cand, hist_vis=hist, max_new_tokens=max_new, eos_token_id=eos,
                                   pad_token_id=pad, trie=trie)
    ids_l, vix_l, vis_all, _ = m._vis_layout(ids_t, am, cand.to(DEV), hist.to(DEV), None)
    kv = KVCacheLM(m, B, capacity=256)
    got = kv.generate(ids_l, vix_l, vis_all, max_new_tokens=max_new, eos_token_id=eos, pad_token_id=pad, trie=trie)
    print("oracle:", ref_ids)
    print("hip   :", got)
    # the first generated token comes from the prefill alone: its logits must agree closely
    for b in range(B):
        agree = 0
        for x, y in zip(got[b], ref_ids[b]):
            if x != y:
                break
            agree += 1
        # a disagreement is only acceptable as a near-tie; re-derive the oracle's margin at that step
        if agree < len(ref_ids[b]):
            seq = torch.tensor(ids_t[b, am[b].bool()].tolist() + ref_ids[b][:agree])[None]
            _, lg, _ = O.lm_forward(P16, cfg, seq, torch.ones_like(seq), cand_vis=cand[2 * b:2 * b + 2], hist_vis=hist[b:b + 1])
            lgl = lg[0, -1].float()
            if trie is not None:
                node = trie.root
                for t in ref_ids[b][:agree]:
                    node = trie.get_next_node(node, t)
                allow = torch.zeros_like(lgl, dtype=torch.bool)
                allow[trie.get_child_index(node)] = True
                lgl = lgl.masked_fill(~allow, float("-inf"))
            margin = (lgl[ref_ids[b][agree]] - lgl[got[b][agree]]).item()
            print(f"sample {b}: diverged at step {agree}, oracle margin {margin:.4f}")
            assert abs(margin) < 0.05, (b, agree, margin)   # a bf16 near-tie (the single-sample re-derivation itself moves by ~1e-2)
        assert agree >= 1 or len(ref_ids[b]) == 0
    if with_trie:
        # every generated sequence is a path of the trie (until eos), whatever the logits were
        for b in range(B):
            node = trie.root
            for t in got[b]:
                assert t in trie.get_child_index(node) or t == pad, (b, t)
                if t == eos:
                    break
                node = trie.get_next_node(node, t)


def test_generation_modes_return_sentences():
    """model('3dqa' | 'summarization', batch, training=False) -> {'generated_sentences'} (nav_model.py:324-343,388-404)"""
    from navillm_amd.nav_model import NavModel
    cfg = _mid_cfg(layers=1)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=3)
    m.eval()
    B = 2
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(5, cfg.image_feat_size, generator=g), torch.randn(3, cfg.image_feat_size, generator=g)]
    ids_l = []
    for b, f in enumerate(feats):
        ids = [1] + torch.randint(3, cfg.base_vocab_size, (12 + b,), generator=g).tolist()
        for j in range(f.shape[0]):
            ids[2 + j] = cfg.cand_token_id
        ids_l.append(ids)
    S = max(len(x) for x in ids_l)
    ids_t = torch.full((B, S), cfg.pad_token_id, dtype=torch.int64)
    am = torch.zeros((B, S), dtype=torch.int64)
    for b, x in enumerate(ids_l):
        ids_t[b, S - len(x):] = torch.tensor(x)
        am[b, S - len(x):] = 1
    out = m("3dqa", {"features": feats, "question": ["q"] * B, "input_ids": ids_t, "attention_mask": am}, training=False,
            max_new_tokens=6, do_sample=False, temperature=1.0)
    assert len(out["generated_sentences"]) == B and all(1 <= len(x) <= 6 for x in out["generated_ids"])
    batch = {"features": feats, "question": ["q"] * B, "input_ids": ids_t, "attention_mask": am}
    # sampling (llava.py:58-62 forwards --do_sample / --temperature): temperature -> 0 reproduces greedy decoding; at T = 1 the draws
    # differ between seeds, stay inside the vocabulary minus the special ids, and are reproducible from torch's generator
    cold = m("3dqa", batch, training=False, max_new_tokens=6, do_sample=True, temperature=1e-4)
    assert cold["generated_ids"] == out["generated_ids"]
    draws = []
    for seed in (1, 1, 2, 3):
        torch.manual_seed(seed)
        draws.append(m("3dqa", batch, training=False, max_new_tokens=6, do_sample=True, temperature=1.0)["generated_ids"])
    assert draws[0] == draws[1] and (draws[0] != draws[2] or draws[0] != draws[3])
    special = set(cfg.special_token_ids)
    assert all(0 <= t < cfg.vocab_size and t not in special for d in draws for row in d for t in row)


def test_nav_step_feedback_modes():
    """mp3d_agent.py:759-772: argmax / sample (Categorical(softmax(logits / T))) / teacher action selection in the rollout driver"""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    cfg = _mid_cfg(layers=1)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=3)
    m.eval()
    crit = CrossEntropyLoss()
    hist = {}
    for tag, kw in (("argmax", {}), ("cold", dict(feedback="sample", temperature=1e-4)), ("hot", dict(feedback="sample", temperature=50.0))):
        ep = SyntheticEpisodes(cfg, 4, seed=5, instr_len=40, device=torch.device(DEV))
        torch.manual_seed(0)
        pos = []
        with torch.no_grad():
            for t in range(4):
                nav_step(m, crit, ep, train=False, **kw)
                pos.append(list(ep.cur))
        hist[tag] = pos
    assert hist["argmax"] == hist["cold"]                 # T -> 0: the sample is the argmax
    assert hist["hot"] != hist["argmax"]                  # T large: (nearly) uniform over the candidates
    with pytest.raises(NotImplementedError):
        nav_step(m, crit, SyntheticEpisodes(cfg, 4, seed=5, instr_len=40, device=torch.device(DEV)), train=False, feedback="beam")


def test_decode_pick_and_advance_kernels_vs_torch():
    """nv_decode_pick_bf16 (masked argmax, first index on ties, HF's finished/pad bookkeeping) and nv_decode_advance (cache
    indices, dyn = {max len + 1, 128-aligned min len}) against the same arithmetic in torch"""
    from navillm_amd import ops, lib
    L = ops._L()
    B, V, Vp, cap, eos, pad = 5, 1006, 1024, 512, 2, 1005
    sp0, nsp = 1000, 5
    g = torch.Generator().manual_seed(3)
    lg = (torch.randn(B, Vp, generator=g) * 2).to(torch.bfloat16)
    lg[0, 1001] = 50.0                       # a special id holds the maximum: must be skipped
    lg[1, 1010] = 60.0                       # beyond the vocabulary
    lg[2, 7] = lg[2, 300] = lg[2, 650] = 40.0   # ties -> smallest id
    lg[3, eos] = 70.0                        # finishes now
    lgd = lg.to(DEV)
    n = L.nv_decode_state_ints(B)
    st = torch.zeros(n, dtype=torch.int32)
    st[B + 4] = 1                            # sample 4 finished earlier -> pad
    lens = torch.tensor([130, 257, 300, 128, 140], dtype=torch.int32)
    st[2 * B:3 * B] = lens
    st[7 * B + 2] = 1                        # second step
    std = st.to(DEV)
    out = torch.full((4, B), -1, dtype=torch.int32, device=DEV)
    lib.check(L.nv_decode_pick_bf16(lgd.data_ptr(), Vp, V, sp0, nsp, std.data_ptr(), out.data_ptr(), 4, B, eos, pad, ops._st()), "pick")
    lib.check(L.nv_decode_advance(std.data_ptr(), B, cap, ops._st()), "advance")
    torch.cuda.synchronize()
    ref = lg.float()[:, :V].clone()
    ref[:, sp0:sp0 + nsp] = float("-inf")
    want = ref.argmax(-1).tolist()
    assert want[2] == 7
    want[4] = pad
    s = std.cpu()
    assert s[:B].tolist() == want and out[1].tolist() == want and out[0].tolist() == [-1] * B
    assert s[B:2 * B].tolist() == [0, 0, 0, 1, 1]
    assert s[2 * B:3 * B].tolist() == (lens + 1).tolist() and s[3 * B:4 * B].tolist() == lens.tolist()
    assert s[4 * B:5 * B].tolist() == [b * cap + int(lens[b]) for b in range(B)] == s[5 * B:6 * B].tolist()
    assert s[6 * B:7 * B].tolist() == list(range(B))
    assert s[7 * B:7 * B + 3].tolist() == [301, 128, 2]
    assert int(s[7 * B + 3]) == 0                               # no cache filled up
    # a FULL cache: the token goes to the junk row, the sample is marked finished and the sticky overflow word is raised
    st2 = torch.zeros(n, dtype=torch.int32)
    st2[2 * B:3 * B] = torch.tensor([cap, 10, cap - 1, 20, 30], dtype=torch.int32)
    std2 = st2.to(DEV)
    lib.check(L.nv_decode_advance(std2.data_ptr(), B, cap, ops._st()), "advance")
    lib.check(L.nv_decode_advance(std2.data_ptr(), B, cap, ops._st()), "advance")      # sample 2 fills up in the first, overflows in the second
    torch.cuda.synchronize()
    s2 = std2.cpu()
    assert int(s2[7 * B + 3]) == 1 and s2[B:2 * B].tolist() == [1, 0, 1, 0, 0]
    assert s2[2 * B:3 * B].tolist() == [cap, 12, cap, 22, 32]
    assert s2[4 * B:5 * B].tolist() == [B * cap, cap + 11, B * cap, 3 * cap + 21, 4 * cap + 31]


def test_attention_with_device_side_lengths_equals_static_launch():
    from navillm_amd import ops, lib
    L = ops._L()
    B, H, hd, cap, S, qmin = 3, 4, 128, 512, 300, 256
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B * cap + 1, 3 * H * hd, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    kv0 = torch.zeros(B, dtype=torch.int32, device=DEV)
    outs = []
    for dyn in (False, True):
        out = torch.zeros(B * cap, H * hd, dtype=torch.bfloat16, device=DEV)
        lse = torch.zeros(B, H, cap, dtype=torch.float32, device=DEV)
        if dyn:
            d = torch.tensor([S, qmin], dtype=torch.int32, device=DEV)
            lib.check(L.nv_attn_fwd_strided_dyn_bf16(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), kv0.data_ptr(), B, cap, H, hd, d.data_ptr(),
                                                     ops._st()), "attn dyn")
        else:
            ops.attn_fwd_strided(qkv, kv0, B, S, cap, H, hd, out, lse, q_row_min=qmin)
        torch.cuda.synchronize()
        outs.append((out.cpu(), lse.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0].view(B, cap, -1)[:, qmin:S].abs().sum() > 0 and outs[0][0].view(B, cap, -1)[:, :qmin].abs().sum() == 0


def test_device_side_greedy_loop_equals_host_loop(monkeypatch):
    """greedy decoding with the choice + bookkeeping on the device and the step replayed from a hipGraph == the host loop
    (token for token, early eos / pad rows included); the captured graph is reused by a second call"""
    import navillm_amd.kvcache as kvm
    from navillm_amd.nav_model import NavModel
    from navillm_amd.kvcache import KVCacheLM
    cfg = _mid_cfg(layers=3)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=11)
    m.eval()
    B = 4
    ids_t, am, cand, hist = _gen_case(cfg, B, 321)
    ids_l, vix_l, vis_all, _ = m._vis_layout(ids_t, am, cand.to(DEV), hist.to(DEV), None)
    pad = 0
    runs = {}
    monkeypatch.setattr(kvm, "DEVICE_GREEDY", False)
    kv = KVCacheLM(m, B, capacity=256)
    free = kv.generate(ids_l, vix_l, vis_all, max_new_tokens=12, eos_token_id=-7, pad_token_id=pad)
    assert all(len(x) == 12 for x in free)
    eos = free[1][2]                          # sample 1 (at least) finishes at its third token
    for tag, dev_loop, graph in (("host", False, False), ("device", True, False), ("graph", True, True)):
        monkeypatch.setattr(kvm, "DEVICE_GREEDY", dev_loop)
        monkeypatch.setattr(kvm, "USE_HIP_GRAPH", graph)
        kv = KVCacheLM(m, B, capacity=256)
        a = kv.generate(ids_l, vix_l, vis_all, max_new_tokens=12, eos_token_id=eos, pad_token_id=pad)
        b = kv.generate(ids_l, vix_l, vis_all, max_new_tokens=12, eos_token_id=eos, pad_token_id=pad)      # graph reuse
        c = kv.generate(ids_l, vix_l, vis_all, max_new_tokens=5, eos_token_id=-7, pad_token_id=pad)
        runs[tag] = (a, b, c)
        assert a == b
        # the cache bookkeeping stays consistent: a following extend() of prompt + generated tokens reuses everything but the last token
        seqs = [list(ids_l[i]) + c[i] for i in range(B)]
        vix = [list(vix_l[i]) + [-1] * len(c[i]) for i in range(B)]
        keys = [("gen", r) for r in range(vis_all.shape[0])]
        kv.extend(seqs, vix, vis_all, keys)
        if dev_loop:
            assert kv.last_stats["new"] == [1] * B, kv.last_stats
    print("host  :", runs["host"][0])
    print("graph :", runs["graph"][0])
    assert runs["host"] == runs["device"] == runs["graph"]
    assert pad in runs["host"][0][1] or len(runs["host"][0][1]) == 3


@pytest.mark.parametrize("fp8w", [False, True])
@pytest.mark.parametrize("M", [1, 8, 13])
def test_gemv_pre_modes_equal_the_unfused_kernels_bit_for_bit(M, fp8w):
    """nv_gemv_pre (RMSNorm / SwiGLU / row gather folded into the weight streamer) == the row kernel followed by the plain GEMV"""
    from navillm_amd import ops, lib, fp8
    L = ops._L()
    d, ff = 512, 1408
    g = torch.Generator().manual_seed(17 + M)
    x = (torch.randn(M, d, generator=g) * 1.3).to(torch.bfloat16).to(DEV)
    nw = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16).to(DEV)
    W1 = (torch.randn(2 * ff, d, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    W2 = (torch.randn(d, ff, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    R = torch.randn(M, d, generator=g).to(torch.bfloat16).to(DEV)
    q1 = fp8.quantize_rows(W1) if fp8w else None
    q2 = fp8.quantize_rows(W2) if fp8w else None

    def pre(A, rows, W, q, N, K, lda, R=None, norm=None, swiglu=False):
        out = torch.empty(M, N // 2 if swiglu else N, dtype=torch.bfloat16, device=DEV)
        Wp, sp = (q[0].data_ptr(), q[1].data_ptr()) if q is not None else (W.data_ptr(), None)
        lib.check(L.nv_gemv_pre(A.data_ptr(), ops._p(rows), Wp, sp, out.data_ptr(), ops._p(R), M, N, K, lda, K, out.shape[1],
                                N if R is not None else 0, int(norm is not None), ops._p(norm), 1e-6, int(swiglu), ops._st()), "nv_gemv_pre")
        return out

    def plain(A, W, q, R=None):
        if q is not None:
            return fp8.gemv_fp8w(A, q[0], q[1], R=R, epilogue=ops.EPI_RESID if R is not None else ops.EPI_STORE)
        return ops.gemm_bf16(ops.NT, A, W, R=R, epilogue=ops.EPI_RESID if R is not None else ops.EPI_STORE)

    def norm_of(t):
        n = ops.rmsnorm_fwd(t, nw, 1e-6)
        return n[0] if isinstance(n, tuple) else n

    # RMSNorm -> gate|up
    gu = plain(norm_of(x), W1, q1)
    got = pre(x, None, W1, q1, 2 * ff, d, d, norm=nw)
    assert torch.equal(got.view(torch.int16), gu.view(torch.int16)), (got.float() - gu.float()).abs().max()
    # RMSNorm -> gate|up -> SwiGLU in the epilogue (gate|up never stored)
    h = ops.swiglu_fwd(gu)
    got_h = pre(x, None, W1, q1, 2 * ff, d, d, norm=nw, swiglu=True)
    assert torch.equal(got_h.view(torch.int16), h.view(torch.int16)), (got_h.float() - h.float()).abs().max()
    # down (+ residual) on it
    assert torch.equal(pre(h, None, W2, q2, d, ff, ff, R=R).view(torch.int16), plain(h, W2, q2, R=R).view(torch.int16))
    # gathered rows (+ RMSNorm over gathered rows)
    big = (torch.randn(40, d, generator=g)).to(torch.bfloat16).to(DEV)
    rows = torch.randperm(40, generator=g)[:M].to(torch.int32).to(DEV)
    sel = ops.gather_rows_bf16(big, rows)
    W3 = W1[:d].contiguous()
    q3 = fp8.quantize_rows(W3) if fp8w else None
    assert torch.equal(pre(big, rows, W3, q3, d, d, d, R=R).view(torch.int16), plain(sel, W3, q3, R=R).view(torch.int16))
    assert torch.equal(pre(big, rows, W3, q3, d, d, d, norm=nw).view(torch.int16), plain(norm_of(sel), W3, q3).view(torch.int16))


def test_rope_scatter_equals_rope_then_scatter():
    from navillm_amd import ops, lib
    L = ops._L()
    M, H, hd, rows_dst = 11, 4, 128, 64
    g = torch.Generator().manual_seed(23)
    qkv = torch.randn(M, 3 * H * hd, generator=g).to(torch.bfloat16).to(DEV)
    pos = torch.randint(0, 500, (M,), generator=g).to(torch.int32).to(DEV)
    rows = torch.randperm(rows_dst, generator=g)[:M].to(torch.int32).to(DEV)
    inv = 1.0 / (10000 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    emb = torch.cat([torch.outer(torch.arange(512).float(), inv)] * 2, -1)
    cos_t, sin_t = emb.cos().to(torch.bfloat16).to(DEV).contiguous(), emb.sin().to(torch.bfloat16).to(DEV).contiguous()
    want = torch.zeros(rows_dst, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    a = qkv.clone()
    ops.rope_rows_(a, cos_t, sin_t, pos, H, hd)
    ops.scatter_rows_bf16_(a, rows, want)
    got = torch.zeros_like(want)
    lib.check(L.nv_rope_scatter_rows_bf16(qkv.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), pos.data_ptr(), rows.data_ptr(), got.data_ptr(),
                                          M, H, hd, 3 * H * hd, ops._st()), "rope_scatter")
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("fp8w", [False, True])
def test_fused_decode_layers_equal_the_unfused_sequence(monkeypatch, fp8w, big):
    """nv_decoder_extend with <= 16 new rows: 6 fused launches per layer == the 11-launch sequence, bit for bit (hidden states of
    all rows and the cache contents), bf16 and weight-only fp8"""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.kvcache import KVCacheLM
    cfg = _mid_cfg(layers=3)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=13)
    m.eval()
    if fp8w:
        m.to_fp8_weight_only()
    B = 4
    g = torch.Generator().manual_seed(2)
    prompt = [torch.randint(3, cfg.base_vocab_size, (50 + 7 * b,), generator=g).tolist() for b in range(B)]
    # big: > 16 new rows (tile GEMMs; only RoPE + scatter are fused there)
    more = [p + torch.randint(3, cfg.base_vocab_size, ((9 if big else 1) + (b % 3),), generator=g).tolist() for b, p in enumerate(prompt)]
    res = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("NV_DECODER_FUSED", fused)
        kv = KVCacheLM(m, B, capacity=128)
        kv.extend(prompt)
        hs_all, _ = kv.extend(more, return_rows="all")
        assert (kv.last_stats["block_rows"] > 16) == big
        torch.cuda.synchronize()
        res[fused] = (hs_all.clone(), [q.clone() for q in kv.qkv])
    assert torch.equal(res["0"][0].view(torch.int16), res["1"][0].view(torch.int16))
    for a, b in zip(res["0"][1], res["1"][1]):
        assert torch.equal(a[:-1].view(torch.int16), b[:-1].view(torch.int16))      # (the last row is the junk row)


@pytest.mark.parametrize("lens", [[1, 5, 64, 65], [300, 129, 511, 512, 17, 256, 640, 1000]])
def test_decode_attention_vs_fp32_reference_and_tile_kernel(lens):
    """nv_attn_decode_bf16 (one query per sample, streaming) against softmax(q k^T / sqrt(d)) v in fp32 on the same bf16 cache, and
    against the tile kernel's row (which rounds P to bf16): both within bf16 output noise"""
    from navillm_amd import ops, lib
    L = ops._L()
    B, H, hd, cap = len(lens), 4, 128, 1024
    g = torch.Generator().manual_seed(41)
    qkv = (torch.randn(B * cap + 1, 3 * H * hd, generator=g)).to(torch.bfloat16)
    qkv[:, H * hd:2 * H * hd] *= 0.6
    qd = qkv.to(DEV)
    pos = torch.tensor([n - 1 for n in lens], dtype=torch.int32).to(DEV)
    crow = torch.tensor([b * cap + lens[b] - 1 for b in range(B)], dtype=torch.int32).to(DEV)
    out = torch.zeros(B, H * hd, dtype=torch.bfloat16, device=DEV)
    lib.check(L.nv_attn_decode_bf16(qd.data_ptr(), crow.data_ptr(), pos.data_ptr(), out.data_ptr(), B, H, hd, cap, ops._st()), "decode attn")
    torch.cuda.synchronize()
    got = out.float().cpu().view(B, H, hd)
    ref = torch.zeros(B, H, hd)
    for b in range(B):
        rows = qkv[b * cap:b * cap + lens[b]].float().view(lens[b], 3, H, hd)
        q = rows[-1, 0]                                           # [H, hd]
        k, v = rows[:, 1], rows[:, 2]                              # [L, H, hd]
        s = torch.einsum("hd,lhd->hl", q, k) / hd ** 0.5
        ref[b] = torch.einsum("hl,lhd->hd", torch.softmax(s, -1), v)
    err = (got - ref).abs().max().item()
    assert err <= 2 ** -8 * max(1.0, ref.abs().max().item()) * 1.01, err
    # the tile kernel on the same cache
    Lmax = max(lens)
    tile = torch.zeros(B * cap, H * hd, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(B, H, cap, dtype=torch.float32, device=DEV)
    ops.attn_fwd_strided(qd, torch.zeros(B, dtype=torch.int32, device=DEV), B, Lmax, cap, H, hd, tile, lse, q_row_min=0)
    torch.cuda.synchronize()
    trow = torch.stack([tile[b * cap + lens[b] - 1] for b in range(B)]).float().cpu().view(B, H, hd)
    assert (got - trow).abs().max().item() <= 2 ** -6 * max(1.0, ref.abs().max().item())


def test_two_batches_in_flight_walk_the_same_trajectories_as_sequential_rollouts():
    """rollout_interleaved (host phase of one batch under the GPU phase of the other, own K/V cache per batch) == the two
    rollouts run one after the other: same viewpoints visited, same final logits"""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.kvcache import KVCacheLM
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step, rollout_interleaved
    cfg = _mid_cfg(layers=2)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=3)
    m.eval()
    crit = CrossEntropyLoss()
    B, T = 3, 7
    mk = lambda k: SyntheticEpisodes(cfg, B, seed=50 + k, instr_len=60, device=torch.device(DEV))
    # reference: the same two batches stepped alternately WITH a sync after every step (the candidate permutation of
    # forward_navigation draws from torch's global RNG, so the order of the model calls must be the same: A0 B0 A1 B1 ...)
    ref_eps = [mk(0), mk(1)]
    ref_kvs = [KVCacheLM(m, B, capacity=512) for _ in ref_eps]
    seq_logits = [None, None]
    torch.manual_seed(7)
    with torch.no_grad():
        for t in range(T):
            for i, ep in enumerate(ref_eps):
                m.kv = ref_kvs[i]
                _, lg = nav_step(m, crit, ep, train=False)
                seq_logits[i] = lg.float().cpu()
    seq_pos = [list(e.cur) for e in ref_eps]
    m.kv = None
    torch.manual_seed(7)
    eps = [mk(0), mk(1)]
    kvs = [KVCacheLM(m, B, capacity=512) for _ in eps]
    last = rollout_interleaved(m, eps, kvs, T)
    assert [list(e.cur) for e in eps] == seq_pos
    for a, b in zip(last, seq_logits):
        assert torch.equal(a.float().cpu(), b)
