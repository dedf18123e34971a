"""GPU: training with the prompt's static prefix computed once per episode (navillm_amd/episode.py) against the default path
that recomputes the whole prompt at every step like the reference (tasks/agents/mp3d_agent.py:726,756): same logits at every
step, same gradients after the episode."""
import numpy as np
import pytest
import torch

from util import bf16_ulps_at_scale
from test_round2_gpu import _mid_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _episode(model, cfg, steps, use_prefix, seed=31, B=3, instr_len=180, teacher_forced=False):
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    from navillm_amd.losses import CrossEntropyLoss
    ep = SyntheticEpisodes(cfg, B, seed=seed, instr_len=instr_len, device=torch.device(DEV))
    for b in range(B):
        ep.instr[b] = ep.instr[b][: instr_len - 23 * b]             # ragged prompts: prefixes of different lengths
    crit = CrossEntropyLoss()
    model.zero_grad()
    model.store.touched.clear()
    if use_prefix:
        model.begin_episode(ep.prefix_ids(), teacher_forced=teacher_forced)
    logits = []
    for t in range(steps):
        torch.manual_seed(500 + t)
        _, lg = nav_step(model, crit, ep, train=True, last=(t == steps - 1))
        logits.append(lg)
    if use_prefix:
        stats = dict(model.episode.stats)
        model.finish_episode()
    torch.cuda.synchronize()
    # (a teacher-forced episode hands out deferred-logits handles: their values exist after finish_episode())
    logits = [(lg.value if hasattr(lg, "value") else lg).detach().float().cpu() for lg in logits]
    grads = {g: t.detach().float().clone() for g, t in model.store.grad.items()}
    return logits, grads, (stats if use_prefix else None)


@pytest.mark.parametrize("size,defer,fuse", [("mid", "all", "1"), ("mid", "all-steps", "1"), ("mid", "wgrad", "1"), ("mid", "none", "1"),
                                             ("mid", "wgrad", "0"), ("7b-width", "all", "1"), ("7b-width", "all-steps", "1"),
                                             ("7b-width", "wgrad", "1")])
def test_prefix_episode_matches_per_step_recompute(size, defer, fuse, monkeypatch):
    """mid: d=512, 3 layers; 7b-width: Vicuna-7B's d=4096 / 32 heads / ff=11008 with two layers (multi-tile GEMMs, split-K tails,
    32 heads in the strided attention backward).  defer: "all" (default) = the steps' whole LM backward batched into finish(),
    "wgrad" = only the weight gradients as ONE GEMM per weight over all token rows at finish(), "none" = everything step by step.  fuse: the prefix rows' K/V gradients accumulated in fp32 by the attention
    backward itself (default) or by the separate nv_kv_grad_accum_f32 pass over the bf16 rows."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd import config as nvcfg
    # "all-steps": mode "all" with one strided attention backward per step (round 3a) instead of nv_attn_bwd_episode_bf16
    monkeypatch.setenv("NAVILLM_EPISODE_ATTN_BWD", "steps" if defer == "all-steps" else "episode")
    defer = defer.split("-")[0]
    monkeypatch.setenv("NAVILLM_EPISODE_DEFER", defer)
    monkeypatch.setenv("NAVILLM_EPISODE_FUSE_KVACC", fuse)
    cfg = _mid_cfg() if size == "mid" else nvcfg.vicuna_7b(image_feat_size=768, num_layers=2, base_vocab_size=2000)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()                                                        # dropout off: both runs see the same encoder outputs
    steps = 4
    # measured on MI355X: mid 1.0-1.5 spacings / 0.9-1.1 % ; 7B width 2.3-2.9 spacings / 1.5-2.4 % (the RoPE position frame differs
    # between the two formulations, on top of the summation order) -> asserted x1.4
    max_ulps, max_rel, max_rel_tensor = (3.0, 2.5e-2, 4e-2) if size == "mid" else (4.0, 3.3e-2, 3.5e-2)
    l_ref, g_ref, _ = _episode(m, cfg, steps, use_prefix=False)
    touched_ref = set(m.store.touched)
    l_pre, g_pre, stats = _episode(m, cfg, steps, use_prefix=True)
    assert set(m.store.touched) == touched_ref
    assert m.episode.prefix is None                                  # finish() closed the episode
    for t in range(steps):
        fin = torch.isfinite(l_ref[t])
        assert torch.equal(torch.isfinite(l_pre[t]), fin)
        u = bf16_ulps_at_scale(l_pre[t], l_ref[t])
        print(f"[episode {size} step {t}] logits prefix-reuse vs recompute: {(l_pre[t][fin] - l_ref[t][fin]).abs().max().item():.5f} = {u:.2f} bf16 ulps")
        assert u <= max_ulps
    st = m.store
    worst = {}
    for g in g_ref:
        rel = ((g_pre[g] - g_ref[g]).norm() / (g_ref[g].norm() + 1e-20)).item()
        worst[g] = rel
        assert rel < max_rel, (g, rel)
    for n in ("lang_model.model.layers.0.self_attn.q_proj.weight", "lang_model.model.layers.0.self_attn.k_proj.weight",
              f"lang_model.model.layers.{cfg.num_layers - 1}.self_attn.v_proj.weight", "lang_model.model.layers.1.mlp.down_proj.weight",
              "lang_model.model.layers.0.input_layernorm.weight", "lang_model.model.embed_tokens.weight", "out_head.0.weight",
              "img_embeddings.mapper.weight"):
        o, k = st.offsets[n], st.sizes[n]
        a, b = g_pre[st.group_of[n]][o:o + k], g_ref[st.group_of[n]][o:o + k]
        rel = ((a - b).norm() / (b.norm() + 1e-20)).item()
        worst[n] = rel
        assert rel < max_rel_tensor, (n, rel)
    print(f"[episode {size}] gradient rel err prefix-reuse vs recompute:", {k: round(v, 4) for k, v in worst.items()})
    rows_ref = steps * sum(180 - 23 * b + 90 for b in range(3))
    print(f"[episode] token rows through the LM: prefix {stats['prefix_rows']} once + suffixes {stats['suffix_rows']} (recompute: ~{rows_ref})")
    assert stats["prefix_rows"] + sum(stats["suffix_rows"]) < 0.6 * rows_ref


@pytest.mark.parametrize("H,lens,ns", [(4, [70, 131, 64], [[5, 9, 1], [40, 70, 65], [17, 3, 30]]),
                                       (2, [1, 300], [[129, 2]]),
                                       (32, [530, 512, 541, 499, 520, 533, 507, 528], [[95, 101, 88, 110, 97, 104, 92, 99]] * 2)])
def test_attn_bwd_episode_kernel_vs_one_strided_backward_per_step(H, lens, ns):
    """nv_attn_bwd_episode_bf16 (all the steps of an episode in one launch per kernel, reading the episode's row buffers in place) against
    the round-3a sequence it replaces: per step scatter into the K/V cache layout -> nv_attn_bwd_strided_kvacc_bf16 -> gather -> RoPE^T.
    Same bf16-rounded probabilities in both; the key tiles are cut at different places, so sums differ in the last bits."""
    from navillm_amd import ops
    dev = torch.device(DEV)
    g = torch.Generator(device="cpu").manual_seed(5)
    B, hd, cap, T = len(lens), 128, 512 if max(lens) < 400 else 1024, len(ns)
    d = H * hd
    BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32
    cu = np.zeros(B + 1, np.int32); cu[1:] = np.cumsum(lens)
    Mp = int(cu[-1])
    Ns = [max(n) for n in ns]
    Ms = [sum(n) for n in ns]                                        # packed step blocks: sample after sample, no padding rows
    offs = [np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int64) for n in ns]
    r0s, R = [], Mp
    for M_ in Ms:
        r0s.append(R); R += M_
    qkv = (torch.randn(R, 3 * d, generator=g) * 0.8).to(BF).to(dev)
    dout = torch.zeros(R, d, dtype=BF, device=dev)
    attn = torch.zeros(R, d, dtype=BF, device=dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(cap).float(), inv)
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(BF).to(dev).contiguous(), emb.sin().to(BF).to(dev).contiguous()
    cache = torch.zeros(B * cap + 1, 3 * d, dtype=BF, device=dev)
    attn_buf = torch.zeros(B * cap + 1, d, dtype=BF, device=dev)
    dout_full = torch.zeros(B * cap + 1, d, dtype=BF, device=dev)
    dqkv_full = torch.zeros(B * cap + 1, 3 * d, dtype=BF, device=dev)
    kv0 = torch.zeros(B, dtype=I32, device=dev)
    lens_dev = torch.tensor(lens, dtype=I32, device=dev)
    pcrow = torch.tensor(np.concatenate([b * cap + np.arange(n) for b, n in enumerate(lens)]), dtype=I32, device=dev)
    ops.scatter_rows_bf16_(qkv[:Mp], pcrow, cache)
    lses, crows, poss = [], [], []
    junk = B * cap
    for t in range(T):
        M = Ms[t]
        crow = np.full(M, junk, np.int32); pos = np.zeros(M, np.int32)
        for b in range(B):
            ar = np.arange(lens[b], lens[b] + ns[t][b], dtype=np.int32)
            o = int(offs[t][b])
            crow[o:o + ns[t][b]] = b * cap + ar
            pos[o:o + ns[t][b]] = ar
        crow_d, pos_d = torch.tensor(crow, dtype=I32, device=dev), torch.tensor(pos, dtype=I32, device=dev)
        crows.append(crow_d); poss.append(pos_d)
        rows = slice(r0s[t], r0s[t] + M)
        live = torch.tensor(crow != junk, device=dev)
        dout[rows] = torch.where(live[:, None], (torch.randn(M, d, generator=g) * 0.5).to(BF).to(dev), torch.zeros((), dtype=BF, device=dev))
        # forward of step t over the cache: its attention outputs and lse (the next step overwrites the same cache rows, as in the episode)
        ops.scatter_rows_bf16_(qkv[rows], crow_d, cache)
        lse = torch.zeros(B, H, cap, dtype=F32, device=dev)
        Lmax = max(lens[b] + ns[t][b] for b in range(B))
        qmin = (min(lens) // 128) * 128
        ops.attn_fwd_strided(cache, kv0, B, Lmax, cap, H, hd, out=attn_buf, lse2=lse, q_row_min=qmin)
        grow = torch.tensor(np.where(crow == junk, 0, crow), dtype=I32, device=dev)
        ops.gather_rows_bf16(attn_buf, grow, out=attn[rows])
        lses.append(lse)
    # ---- reference sequence
    acc_ref = torch.full((B * cap, 2 * d), float("nan"), dtype=F32, device=dev)
    dqkv_ref = torch.zeros(R, 3 * d, dtype=BF, device=dev)
    for t in range(T):
        M = Ms[t]
        rows = slice(r0s[t], r0s[t] + M)
        Lmax = max(lens[b] + ns[t][b] for b in range(B))
        ops.scatter_rows_bf16_(qkv[rows], crows[t], cache)
        ops.scatter_rows_bf16_(attn[rows], crows[t], attn_buf)
        ops.scatter_rows_bf16_(dout[rows], crows[t], dout_full)
        ops.attn_bwd_strided(cache, attn_buf, dout_full, lses[t], kv0, B, Lmax, cap, H, hd, dqkv_full, q_row_min=(min(lens) // 128) * 128,
                             kv_acc=acc_ref, prefix_len_i32=lens_dev, first=(t == 0))
        ops.scatter_rows_bf16_(torch.zeros(M, d, dtype=BF, device=dev), crows[t], dout_full)
        ops.gather_rows_bf16(dqkv_full, crows[t], out=dqkv_ref[rows])
        ops.rope_rows_t_(dqkv_ref[rows], cos, sin, poss[t], H, hd)
    # ---- one call
    acc = torch.full((B * cap, 2 * d), float("nan"), dtype=F32, device=dev)
    dqkv = torch.full((R, 3 * d), float("nan"), dtype=BF, device=dev)
    tab = torch.tensor(np.concatenate([np.concatenate([r0s[t] + offs[t] for t in range(T)]).astype(np.int32), np.array(ns, np.int32).reshape(-1)]),
                       dtype=I32, device=dev)
    ptrs = torch.tensor([l.data_ptr() for l in lses], dtype=torch.int64, device=dev)
    ops.attn_bwd_episode(qkv, attn, dout, dqkv, ptrs, torch.tensor(cu, dtype=I32, device=dev), tab, acc, T, B, H, hd, cap, Mp, max(lens), max(Ns),
                         rope=(cos, sin))
    torch.cuda.synchronize()
    assert torch.isnan(dqkv[:Mp].float()).all()                       # the prefix rows of dqkv are not this kernel's
    got, ref = dqkv[Mp:].float(), dqkv_ref[Mp:].float()
    assert torch.isfinite(got).all()                                 # every row of [Mp, R) was written
    for name, c0 in (("dq", 0), ("dk", d), ("dv", 2 * d)):
        a, r = got[:, c0:c0 + d], ref[:, c0:c0 + d]
        rel = ((a - r).norm() / r.norm()).item()
        worst = ((a - r).abs().max() / r.abs().max()).item()
        print(f"[attn_bwd_episode H={H}] {name}: rel {rel:.2e}, max |diff| / max |ref| {worst:.2e}")
        assert rel < 4e-3 and worst < 2e-2, (name, rel, worst)
    for b in range(B):
        a, r = acc[b * cap:b * cap + lens[b]], acc_ref[b * cap:b * cap + lens[b]]
        assert torch.isfinite(a).all() and torch.isnan(acc[b * cap + lens[b]:(b + 1) * cap]).all()     # only the prefix rows are written
        rel = ((a - r).norm() / r.norm()).item()
        assert rel < 1e-4, (b, rel)


def test_prefix_episode_rejects_foreign_prompts_and_wrong_use():
    from navillm_amd.nav_model import NavModel
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    from navillm_amd.losses import CrossEntropyLoss
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()
    ep = SyntheticEpisodes(cfg, 2, seed=3, instr_len=60, device=torch.device(DEV))
    pre = ep.prefix_ids()
    pre[1][5] += 1                                                   # not a prefix of sample 1's prompts
    m.begin_episode(pre)
    with pytest.raises(AssertionError, match="does not start with the prefix"):
        nav_step(m, CrossEntropyLoss(), ep, train=True, last=True)
    with pytest.raises(AssertionError, match="visual tokens"):
        m.begin_episode([[1, cfg.cand_token_id, 5], [1, 2, 3]])
    # an optimizer step in front of finish_episode() would train on the encoder's gradients alone: refused
    from navillm_amd.optim import FlatAdamW
    opt = FlatAdamW(m, lr=1e-5)
    ep2 = SyntheticEpisodes(cfg, 2, seed=4, instr_len=60, device=torch.device(DEV))
    m.begin_episode(ep2.prefix_ids())
    nav_step(m, CrossEntropyLoss(), ep2, train=True, last=False)
    with pytest.raises(RuntimeError, match="finish_episode"):
        opt.clip_grad_norm_(40.0)
    with pytest.raises(RuntimeError, match="finish_episode"):
        opt.step()
    m.finish_episode()
    opt.clip_grad_norm_(40.0); opt.step(); opt.zero_grad()           # fine now
    m.begin_episode(ep2.prefix_ids())
    nav_step(m, CrossEntropyLoss(), ep2, train=True, last=False)
    m.episode_abort()
    opt.step()                                                       # an aborted episode holds nothing back


def test_episode_buffers_refuse_to_outgrow_the_device(monkeypatch):
    """the deferred forms keep every token row of an episode (131.6 KB per row and layer at 7B): when the rows would not fit the free
    device memory the episode says so and names the way out, instead of dying in the allocator halfway through a rollout"""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.synthetic import SyntheticEpisodes
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    ep = SyntheticEpisodes(cfg, 2, seed=3, instr_len=60, device=torch.device(DEV))
    m.begin_episode(ep.prefix_ids())                                  # fits: builds the episode object and its buffers
    assert m.episode._rows_fit(m.episode._ecap + 1024)
    m.episode_abort()
    real = torch.cuda.mem_get_info
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (0, real()[1]))
    assert not m.episode._rows_fit(10 * m.episode._ecap + (1 << 22))
    m.episode._E, m.episode._ecap = None, 0                           # as if the next episode had to build them afresh
    with pytest.raises(RuntimeError, match="NAVILLM_EPISODE_DEFER=none"):
        m.episode._ensure_rows(1 << 24)


def test_prefix_episode_under_the_data_parallel_wrapper_world1():
    """the episode mode behind NavDataParallel (world of one, exchange forced): the per-layer exchanges are launched from the
    prefix's deferred backward inside final_backward(); gradients must equal the unwrapped episode bit for bit and nothing may
    stay pending."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.parallel import NavDataParallel, RcclComm
    from navillm_amd.synthetic import SyntheticEpisodes, prefix_reuse_episode
    from navillm_amd.losses import CrossEntropyLoss
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()
    crit = CrossEntropyLoss()

    def run(wrapped):
        ep = SyntheticEpisodes(cfg, 3, seed=8, instr_len=120, device=torch.device(DEV))
        m.zero_grad()
        torch.manual_seed(5)
        prefix_reuse_episode(wrapped, crit, ep, 3)
        torch.cuda.synchronize()
        return {g: t.detach().clone() for g, t in m.store.grad.items()}

    base = run(m)
    comm = RcclComm(0, 1)
    ddp = NavDataParallel(m, comm=comm, force_sync=True)
    got = run(ddp)
    assert not ddp._pending and not ddp._queued
    object.__setattr__(m, "_dp", None)
    for g in base:
        assert torch.equal(base[g], got[g]), g
    comm.close()


@pytest.mark.parametrize("task,tf", [("r2r", False), ("reverie", False), ("r2r", True), ("reverie", True)])
def test_mixed_task_episode_with_navigation_over_cached_prefix(task, tf):
    """BASELINE config 3 in prefix-reuse mode: the navigation steps of a multi-task episode run over the cached prompt prefix while
    its sub-tasks (fine-grained R2R, object grounding, summarization: prompts of their own) go through the whole LM; the prefix's
    deferred backward comes last.  Accumulated gradients and every loss must match the all-recompute episode.  tf: the navigation
    steps' forward deferred as well (teacher-forced episode)."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.synthetic import SyntheticEpisodes, mixed_task_episode
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()
    crit = CrossEntropyLoss()

    def run(prefix):
        ep = SyntheticEpisodes(cfg, 3, seed=41, instr_len=150, device=torch.device(DEV), task=task)
        m.zero_grad()
        m.store.touched.clear()
        torch.manual_seed(9)
        losses = mixed_task_episode(m, crit, ep, steps=3, prefix_reuse=prefix, teacher_forced=tf and prefix)
        torch.cuda.synchronize()
        flat = [float(l.detach()) for l in losses["nav"] + losses["fgr2r"] + [losses["og"], losses["sum"]] if l is not None]
        return flat, {g: t.detach().float().clone() for g, t in m.store.grad.items()}, set(m.store.touched)

    l_ref, g_ref, t_ref = run(False)
    l_pre, g_pre, t_pre = run(True)
    assert m.episode.prefix is None and t_pre == t_ref and len(l_ref) == len(l_pre)
    for a, b in zip(l_pre, l_ref):
        assert abs(a - b) <= 2e-2 * max(1.0, abs(b)), (l_pre, l_ref)
    for g in g_ref:
        rel = ((g_pre[g] - g_ref[g]).norm() / (g_ref[g].norm() + 1e-20)).item()
        print(f"[mixed {task}] gradient buffer {g}: prefix-reuse vs recompute rel err {rel:.4f}")
        assert rel < 2.5e-2, (g, rel)


def test_long_episode_flushes_segments_and_matches_recompute(monkeypatch):
    """VERDICT r3 next #7a: a 16-step episode in the default deferred form with the episode buffers capped (NAVILLM_EPISODE_MAX_ROWS), so
    that the steps' deferred backward is flushed in several segments (prefix open throughout, fp32 prefix K/V accumulators stored by the
    first segment and added to by the later ones) -- against the per-step recompute (the reference's formulation) and against the
    unsegmented run of the same mode."""
    from navillm_amd.nav_model import NavModel
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()
    steps = 16
    l_ref, g_ref, _ = _episode(m, cfg, steps, use_prefix=False)
    _episode(m, cfg, steps, use_prefix=True)             # cold: the buffers are sized between episodes, a first long one may run in segments
    l_one, g_one, st_one = _episode(m, cfg, steps, use_prefix=True)
    assert st_one["segments_flushed"] == 0
    rows_per_step = max(st_one["suffix_rows"])
    monkeypatch.setenv("NAVILLM_EPISODE_MAX_ROWS", str(st_one["prefix_rows"] + 3 * rows_per_step))
    l_seg, g_seg, st_seg = _episode(m, cfg, steps, use_prefix=True)
    monkeypatch.delenv("NAVILLM_EPISODE_MAX_ROWS")
    assert st_seg["segments_flushed"] >= 2, st_seg
    assert m.episode.prefix is None
    worst = 0.0
    for t in range(steps):
        assert torch.equal(l_seg[t], l_one[t]), f"step {t}: the forward must not depend on where the segments are cut"
        worst = max(worst, bf16_ulps_at_scale(l_seg[t], l_ref[t]))
    rel = {g: (((g_seg[g] - g_ref[g]).norm() / (g_ref[g].norm() + 1e-20)).item(), ((g_seg[g] - g_one[g]).norm() / (g_one[g].norm() + 1e-20)).item())
           for g in g_ref}
    print(f"[long episode, {steps} steps, {st_seg['segments_flushed']} flushed segments] logits vs recompute: worst {worst:.2f} bf16 spacings; gradient "
          f"rel err (vs recompute, vs the unsegmented run): {rel}")
    assert worst <= 3.0
    for g, (a, b) in rel.items():
        assert a < 2.5e-2 and b < 1.5e-2, (g, a, b)


def test_left_truncated_prompts_fall_back_to_the_reference_formulation():
    """VERDICT r3 next #7b: prompts that reach the tokenizer's 1024-token limit are LEFT-truncated (modified_lm.py:77-87) -- they no
    longer start with the episode's prefix and every hidden state changes, so such a step takes the full `_lm` path inside the open
    episode (gradients straight into `.grad`), the untruncated steps before it stay on the cached prefix, and the episode's result still
    matches the per-step recompute."""
    from navillm_amd.nav_model import NavModel
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()
    steps, kw = 5, dict(seed=31, B=2, instr_len=925)
    l_ref, g_ref, _ = _episode(m, cfg, steps, use_prefix=False, **kw)
    l_pre, g_pre, stats = _episode(m, cfg, steps, use_prefix=True, **kw)
    n_fb = stats["recomputed_steps"]
    print(f"[truncation] prefix rows {stats['prefix_rows']}, suffix rows {stats['suffix_rows']}, steps that fell back to the full prompt: {n_fb}")
    assert 1 <= n_fb < steps and len(stats["suffix_rows"]) == steps - n_fb
    for t in range(steps):
        u = bf16_ulps_at_scale(l_pre[t], l_ref[t])
        if t >= steps - n_fb:
            assert u == 0.0, (t, u)                                   # the same kernels on the same inputs
        else:
            assert u <= 3.0, (t, u)
    for g in g_ref:
        rel = ((g_pre[g] - g_ref[g]).norm() / (g_ref[g].norm() + 1e-20)).item()
        assert rel < 2.5e-2, (g, rel)


@pytest.mark.parametrize("size", ["mid", "7b-width"])
def test_teacher_forced_episode_batches_the_forward_and_matches(size, monkeypatch):
    """round 4: begin_episode(..., teacher_forced=True) -- the steps' LM forward is deferred to finish_episode() and runs as ONE batch
    over all the steps' suffix rows (navillm_amd/episode.py::_forward_lazy).  Against the per-step-forward form of the same mode
    (same kernels on the same rows, only the GEMMs' M differs: last-bit differences) and against the per-step recompute (the
    reference's formulation); also with the batch flushed in segments."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd import config as nvcfg
    from navillm_amd.losses import DeferredLogits
    cfg = _mid_cfg() if size == "mid" else nvcfg.vicuna_7b(image_feat_size=768, num_layers=2, base_vocab_size=2000)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()
    steps = 5
    l_ref, g_ref, _ = _episode(m, cfg, steps, use_prefix=False)
    l_ps, g_ps, _ = _episode(m, cfg, steps, use_prefix=True)
    l_tf, g_tf, st_tf = _episode(m, cfg, steps, use_prefix=True, teacher_forced=True)
    assert m.episode.prefix is None and st_tf["segments_flushed"] == 0
    monkeypatch.setenv("NAVILLM_EPISODE_MAX_ROWS", str(st_tf["prefix_rows"] + 2 * max(st_tf["suffix_rows"])))
    l_sg, g_sg, st_sg = _episode(m, cfg, steps, use_prefix=True, teacher_forced=True)
    monkeypatch.delenv("NAVILLM_EPISODE_MAX_ROWS")
    assert st_sg["segments_flushed"] >= 1
    max_ulps, max_rel = (3.0, 2.5e-2) if size == "mid" else (4.5, 3.3e-2)
    w_ps = w_ref = w_sg = 0.0
    for t in range(steps):
        w_ps = max(w_ps, bf16_ulps_at_scale(l_tf[t], l_ps[t]))
        w_ref = max(w_ref, bf16_ulps_at_scale(l_tf[t], l_ref[t]))
        w_sg = max(w_sg, bf16_ulps_at_scale(l_sg[t], l_tf[t]))
    rel = {g: tuple(round(((a[g] - b[g]).norm() / (b[g].norm() + 1e-20)).item(), 4) for a, b in ((g_tf, g_ps), (g_tf, g_ref), (g_sg, g_tf))) for g in g_ref}
    print(f"[teacher-forced {size}] logits, worst bf16 spacings: batched vs per-step forward {w_ps:.2f}, vs recompute {w_ref:.2f}, segmented vs one batch "
          f"{w_sg:.2f}; gradient rel err (vs per-step forward, vs recompute, segmented vs one batch): {rel}")
    # (round 5: the prefix joins the batch of whichever segment runs first, so the segmented and the one-batch run push the prefix rows
    # through GEMMs of different row counts -- other split-K tails, other last bits: 2.25 spacings at 7B width where round 4, with the
    # prefix computed on its own in both runs, measured <= 2.0)
    assert w_ps <= 2.0 and w_ref <= max_ulps and w_sg <= (2.0 if size == "mid" else 3.0)
    for g, (a, b, c) in rel.items():
        assert a < 1.5e-2 and b < max_rel and c < 1.5e-2, (g, a, b, c)
    # the handle is all a rollout gets before finish_episode(): reading the logits early is an error, not a silent zero
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    from navillm_amd.losses import CrossEntropyLoss
    ep = SyntheticEpisodes(cfg, 3, seed=31, instr_len=180, device=torch.device(DEV))
    m.zero_grad()
    m.begin_episode(ep.prefix_ids(), teacher_forced=True)
    loss, lg = nav_step(m, CrossEntropyLoss(), ep, train=True, last=False)
    assert isinstance(lg, DeferredLogits)
    with pytest.raises(RuntimeError, match="finish_episode"):
        lg.value
    with pytest.raises(RuntimeError, match="finish_episode"):
        float(loss)
    with pytest.raises(AttributeError):
        lg.argmax(1)
    m.finish_episode()
    assert torch.isfinite(lg.value[torch.isfinite(lg.value)]).all() and float(loss) > 0


@pytest.mark.parametrize("size", ["mid", "7b-width"])
def test_episode_forward_attention_one_launch_equals_the_per_step_cache_form(size, monkeypatch):
    """round 5: `nv_attn_fwd_episode_bf16` (the attention of ALL steps of a teacher-forced episode in one launch per layer, reading the
    episode row buffers in place) against round 4's form (scatter each step's q|k|v rows into the K/V-cache layout, one strided forward
    per step, gather the outputs back): every query row sees the same 64-key tiles in the same order, so logits and every gradient
    buffer must be bit-identical -- also when a long episode is flushed in segments."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd import config as nvcfg
    cfg = _mid_cfg() if size == "mid" else nvcfg.vicuna_7b(image_feat_size=768, num_layers=2, base_vocab_size=2000)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()
    steps = 5
    _episode(m, cfg, steps, use_prefix=True, teacher_forced=True)          # (cold start: buffers sized)
    monkeypatch.setenv("NAVILLM_EPISODE_ATTN_FWD", "steps")
    l_old, g_old, st_old = _episode(m, cfg, steps, use_prefix=True, teacher_forced=True)
    monkeypatch.setenv("NAVILLM_EPISODE_ATTN_FWD", "episode")
    l_new, g_new, st_new = _episode(m, cfg, steps, use_prefix=True, teacher_forced=True)
    assert st_new["segments_flushed"] == 0 and st_old["suffix_rows"] == st_new["suffix_rows"]
    for t in range(steps):
        assert torch.equal(l_new[t], l_old[t]), f"step {t}"
    for g in g_old:
        assert torch.equal(g_new[g], g_old[g]), g
    # segmented: the same two forms with the episode buffers capped
    monkeypatch.setenv("NAVILLM_EPISODE_MAX_ROWS", str(st_new["prefix_rows"] + 2 * max(st_new["suffix_rows"])))
    l_seg_new, g_seg_new, st_seg = _episode(m, cfg, steps, use_prefix=True, teacher_forced=True)
    monkeypatch.setenv("NAVILLM_EPISODE_ATTN_FWD", "steps")
    l_seg_old, g_seg_old, _ = _episode(m, cfg, steps, use_prefix=True, teacher_forced=True)
    monkeypatch.delenv("NAVILLM_EPISODE_MAX_ROWS")
    assert st_seg["segments_flushed"] >= 1
    for t in range(steps):
        assert torch.equal(l_seg_new[t], l_seg_old[t]), f"segmented, step {t}"
    for g in g_old:
        assert torch.equal(g_seg_new[g], g_seg_old[g]), ("segmented", g)
    # the per-step-forward form (sampled / argmax rollouts: every step's forward runs at once, its backward is deferred): one table-step
    # of the same kernel per step instead of scatter -> strided forward over the K/V cache -> gather
    monkeypatch.setenv("NAVILLM_EPISODE_ATTN_FWD", "steps")
    l_ps_old, g_ps_old, _ = _episode(m, cfg, steps, use_prefix=True, teacher_forced=False)
    monkeypatch.setenv("NAVILLM_EPISODE_ATTN_FWD", "episode")
    l_ps_new, g_ps_new, _ = _episode(m, cfg, steps, use_prefix=True, teacher_forced=False)
    for t in range(steps):
        assert torch.equal(l_ps_new[t], l_ps_old[t]), f"per-step forward, step {t}"
    for g in g_old:
        assert torch.equal(g_ps_new[g], g_ps_old[g]), ("per-step forward", g)


def test_first_writer_wgrad_store_equals_accumulate_into_zeros(monkeypatch):
    """round 5: when nothing has written a decoder-layer gradient since zero_grad() (`FlatStore.layers_zero`), the batched backward's
    weight-gradient GEMMs STORE instead of read-accumulating 13.5 GB of zeros: bit-identical; and a SECOND episode accumulated on top
    (no zero_grad in between) read-adds as before"""
    from navillm_amd.nav_model import NavModel
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    from navillm_amd.losses import CrossEntropyLoss
    cfg = _mid_cfg()
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()
    res = {}
    for form in ("0", "1"):
        monkeypatch.setenv("NAVILLM_WGRAD_STORE", form)
        _, g1, _ = _episode(m, cfg, 4, use_prefix=True, teacher_forced=True)       # (zero_grad inside: first writer)
        # a second episode on top of the first one's gradients: must ADD
        ep = SyntheticEpisodes(cfg, 3, seed=77, instr_len=120, device=torch.device(DEV))
        m.begin_episode(ep.prefix_ids(), teacher_forced=True)
        for t in range(3):
            torch.manual_seed(900 + t)
            nav_step(m, CrossEntropyLoss(), ep, train=True, last=(t == 2))
        assert not m.store.layers_zero
        m.finish_episode()
        torch.cuda.synchronize()
        res[form] = (g1, {g: t.detach().float().clone() for g, t in m.store.grad.items()})
    # the per-step recompute path (LlamaStack.backward): the first backward after zero_grad() stores, the later ones read-add
    rc = {}
    for form in ("0", "1"):
        monkeypatch.setenv("NAVILLM_WGRAD_STORE", form)
        _, rc[form], _ = _episode(m, cfg, 3, use_prefix=False)
    for g in rc["0"]:
        assert torch.equal(rc["1"][g], rc["0"][g]), ("recompute", g)
    for g in res["0"][0]:
        assert torch.equal(res["1"][0][g], res["0"][0][g]), ("first episode", g)
        assert torch.equal(res["1"][1][g], res["0"][1][g]), ("accumulated second episode", g)
        assert not torch.equal(res["1"][1][g], res["1"][0][g])


def _window_run(m, cfg, plan, accumulate, Bs=2, flush=None):
    """the episodes of `plan` [(seed, instruction length, steps)] one after the other, every step loss scaled 1 / Bs / len(plan) as the
    reference's accumulation does (mp3d_agent.py:750); accumulate > 1: opened as an accumulation window"""
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    from navillm_amd.losses import CrossEntropyLoss
    crit = CrossEntropyLoss()
    m.zero_grad()
    m.store.touched.clear()
    handles = []
    for e, (seed, il, steps) in enumerate(plan):
        ep = SyntheticEpisodes(cfg, Bs, seed=seed, instr_len=il, device=torch.device(DEV))
        ep.instr[1] = ep.instr[1][: il - 17]                       # ragged prefixes inside an episode too
        m.begin_episode(ep.prefix_ids(), teacher_forced=True, accumulate=accumulate)
        for t in range(steps):
            torch.manual_seed(7000 + 31 * e + t)
            loss, lg = nav_step(m, crit, ep, train=True, last=(t == steps - 1), accum=len(plan))
            handles.append((lg, loss))
        m.finish_episode()
    if flush is not None:
        flush()
    torch.cuda.synchronize()
    logits = [lg.value.detach().float().cpu() for lg, _ in handles]
    losses = [float(ls) for _, ls in handles]
    grads = {g: t.detach().float().clone() for g, t in m.store.grad.items()}
    return logits, losses, grads


@pytest.mark.parametrize("size", ["mid", "7b-width"])
def test_accumulation_window_batches_the_episodes_between_two_optimizer_steps(size):
    """round 5 (second half): `begin_episode(..., teacher_forced=True, accumulate=n)` -- the reference launches
    `--batch_size 1 --gradient_accumulation_step 8` (scripts/multi_wo_pretrain.sh:16); the weights are frozen between two optimizer
    steps (train.py:68,86-89), so the n episodes of a window run as ONE batch at the n-th finish_episode() (episode.py::_begin_window).
    Against the same episodes finished one by one (same kernels on the same rows; the GEMMs' M and the weight gradients' summation
    order differ: last bits): logits of every step, loss values, every gradient buffer.  Episodes of different lengths (a shorter
    episode has no rows in the later table-steps), ragged prefixes, and a window that is cut short by the optimizer."""
    from navillm_amd.nav_model import NavModel
    from navillm_amd import config as nvcfg
    from navillm_amd.optim import FlatAdamW
    cfg = _mid_cfg() if size == "mid" else nvcfg.vicuna_7b(image_feat_size=768, num_layers=2, base_vocab_size=2000)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.eval()
    plan = [(41, 150, 3), (42, 96, 2), (43, 171, 4), (44, 120, 3)]
    l_one, s_one, g_one = _window_run(m, cfg, plan, accumulate=1)
    assert m._window is None
    l_win, s_win, g_win = _window_run(m, cfg, plan, accumulate=len(plan))
    assert m._window is not None and m._window.prefix is None, "the window's last finish_episode() runs it"
    st = m._window.stats
    assert st["window_episodes"] == len(plan) and len(st["suffix_rows"]) == sum(p[2] for p in plan)
    worst = max(bf16_ulps_at_scale(a, b) for a, b in zip(l_win, l_one))
    rel = {g: round(((g_win[g] - g_one[g]).norm() / (g_one[g].norm() + 1e-20)).item(), 4) for g in g_one}
    lrel = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(s_win, s_one))
    print(f"[accumulation window {size}] {len(plan)} episodes x 2 prompts, {len(l_one)} steps: logits worst {worst:.2f} bf16 spacings vs one "
          f"finish per episode, loss values rel {lrel:.2e}, gradient buffers rel err {rel}")
    assert worst <= (2.0 if size == "mid" else 3.0) and lrel < 2e-2
    for g, v in rel.items():
        assert v < 1.5e-2, (g, v)
    # a window sized for 8 episodes that the optimizer cuts short after 4: clip_grad_norm_ hands its gradients over first
    opt = FlatAdamW(m, lr=0.0)
    l_cut, _, g_cut = _window_run(m, cfg, plan, accumulate=8, flush=lambda: opt.clip_grad_norm_(40.0))
    assert m._window.prefix is None
    for a, b in zip(l_cut, l_win):
        assert bf16_ulps_at_scale(a, b) <= 2.0
    for g in g_one:
        assert ((g_cut[g] - g_win[g]).norm() / (g_win[g].norm() + 1e-20)).item() < 1.5e-2, g
    # inside an open window nothing exists yet: reading a value is an error, not a silent zero; a plain episode flushes the window first
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    from navillm_amd.losses import CrossEntropyLoss
    m.zero_grad()
    ep = SyntheticEpisodes(cfg, 2, seed=5, instr_len=100, device=torch.device(DEV))
    m.begin_episode(ep.prefix_ids(), teacher_forced=True, accumulate=4)
    loss, lg = nav_step(m, CrossEntropyLoss(), ep, train=True, last=True, accum=4)
    m.finish_episode()
    assert m._window.window_open() and m._window.has_pending_gradients()
    with pytest.raises(RuntimeError, match="finish_episode"):
        lg.value
    assert float(m.store.grad["lm"].float().abs().max()) == 0.0
    ep2 = SyntheticEpisodes(cfg, 2, seed=6, instr_len=100, device=torch.device(DEV))
    m.begin_episode(ep2.prefix_ids(), teacher_forced=False)         # not part of the window: the window runs now
    assert not m._window.window_open() and float(loss) > 0 and float(m.store.grad["lm"].float().abs().max()) > 0.0
    m.episode_abort()
    m.zero_grad()
