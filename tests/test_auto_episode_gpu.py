"""GPU: AUTOMATIC episodes -- the fast prefix-reuse path behind the reference's own call pattern.

`navillm_amd.synthetic.reference_rollout` / `reference_train_steps` restate `MP3DAgent.rollout` (tasks/agents/mp3d_agent.py:660-778)
and `train_one_epoch` (train.py:60-91) call for call and contain NO `begin_episode` / `finish_episode`: the model opens the episode
on the first grad-enabled training-mode navigation call and hands its gradients over when `torch.nn.utils.clip_grad_norm_(
model.parameters(), 40.)`, the optimizer, or the next rollout's first navigation call comes.  Asserted: logits, loss values and every
gradient buffer are BIT-IDENTICAL to the same rollout inside an explicit `begin_episode(..., teacher_forced=False)` /
`finish_episode()` pair on a fresh model; the closing triggers; the loud failures (zero_grad over pending gradients).
(The reference-pinned G12 episode through the automatic path: tests/test_parity_gpu.py::test_g12_...[auto].)"""
import pytest
import torch

from test_round2_gpu import _mid_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(size):
    from navillm_amd import config as nvcfg
    return _mid_cfg() if size == "mid" else nvcfg.vicuna_7b(image_feat_size=768, num_layers=2, base_vocab_size=2000)


def _model(cfg, auto):
    from navillm_amd.nav_model import NavModel
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=21)
    m.train()
    m.auto_episode = auto
    return m


def _rollouts(m, cfg, plan, explicit, accum):
    """the rollouts of `plan` [(B, steps, instruction length, feedback)] one after the other as the training loop runs them; explicit:
    every rollout wrapped in begin_episode / finish_episode.  -> (logits per step, loss per rollout, flat gradient clones)"""
    from navillm_amd.synthetic import SyntheticEpisodes, reference_rollout
    from navillm_amd.losses import CrossEntropyLoss
    import navillm_amd.synthetic as syn
    crit = CrossEntropyLoss()
    m.zero_grad()
    m.store.touched.clear()
    seen, losses = [], []
    orig = m.forward_navigation

    def spy(mode, batch, **kw):
        out = orig(mode, batch, **kw)
        seen.append(out["fuse_logits"].detach().float().cpu())
        return out
    m.forward_navigation = spy
    try:
        for e, (B, steps, il, fb) in enumerate(plan):
            ep = SyntheticEpisodes(cfg, B, seed=300 + e, instr_len=il, device=torch.device(DEV))
            if B > 1:
                ep.instr[1] = ep.instr[1][: il - 13]
            torch.manual_seed(9000 + e)
            if explicit:
                m.begin_episode(ep.prefix_ids(), teacher_forced=False)
            loss = reference_rollout(m, crit, ep, steps, feedback=fb, accum=accum)
            if explicit:
                m.finish_episode()
            losses.append(float(loss))
    finally:
        m.forward_navigation = orig
    return seen, losses


@pytest.mark.parametrize("size", ["mid", "7b-width"])
def test_unmodified_rollout_is_bit_identical_to_the_explicit_episode_form(size):
    cfg = _cfg(size)
    plan = [(2, 3, 140, "teacher"), (2, 2, 111, "sample"), (2, 3, 165, "teacher")]
    # automatic: nothing but the reference's calls; train.py:87's clip (a bound that never clips) hands the last episode over
    a = _model(cfg, auto=True)
    lg_a, ls_a = _rollouts(a, cfg, plan, explicit=False, accum=len(plan))
    assert a._auto_open and a.episode.has_pending_gradients()
    assert a.auto_stats["opened"] == 3 and a.auto_stats["closed_by"] == {"next_episode": 2}
    torch.nn.utils.clip_grad_norm_(a.parameters(), 1e9)
    assert not a._auto_open and a.episode.prefix is None and a.auto_stats["closed_by"] == {"next_episode": 2, "parameters": 1}
    torch.cuda.synchronize()
    g_a = {g: t.detach().clone() for g, t in a.store.grad.items()}
    del a
    b = _model(cfg, auto=False)
    lg_b, ls_b = _rollouts(b, cfg, plan, explicit=True, accum=len(plan))
    torch.cuda.synchronize()
    g_b = {g: t.detach().clone() for g, t in b.store.grad.items()}
    assert len(lg_a) == len(lg_b) == sum(p[1] for p in plan)
    for t, (x, y) in enumerate(zip(lg_a, lg_b)):
        assert torch.equal(x, y), f"{size}: logits of navigation call {t} differ: {(x - y).abs().max().item():.3e}"
    assert ls_a == ls_b
    for g in g_b:
        assert torch.equal(g_a[g], g_b[g]), (size, g, ((g_a[g].float() - g_b[g].float()).norm() / (g_b[g].float().norm() + 1e-30)).item())
    # and it is the prefix-reuse path that ran, not the full-prompt recompute: same episodes with automatic episodes off
    c = _model(cfg, auto=False)
    lg_c, _ = _rollouts(c, cfg, plan[:1], explicit=False, accum=len(plan))
    assert c.episode is None
    assert not all(torch.equal(x, y) for x, y in zip(lg_a, lg_c)), "the recompute path is a different evaluation (RoPE frame, row order)"


def test_reference_training_loop_with_flat_adamw_and_with_the_wrapper():
    """train.py:60-91 verbatim (`reference_train_steps`): torch's clip through `model.parameters()`, FlatAdamW.step / zero_grad, B = 1 x 2
    accumulation; then the same loop with the model inside NavDataParallel (world of one, exchange forced): `wrapped.parameters()`
    must hand the episode over too (nn.Module reads the child's `_parameters` directly).  The two runs end with identical
    parameters."""
    from navillm_amd.synthetic import SyntheticEpisodes, reference_train_steps
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.optim import FlatAdamW
    cfg = _cfg("mid")

    def run(wrap):
        m = _model(cfg, auto=True)
        model = m
        if wrap:
            from navillm_amd.parallel import NavDataParallel
            model = NavDataParallel(m, comm=None)
        opt = FlatAdamW(m, lr=1e-3)
        ep = SyntheticEpisodes(cfg, 1, seed=77, instr_len=90, device=torch.device(DEV))
        torch.manual_seed(5)
        norms = []
        real_clip = torch.nn.utils.clip_grad_norm_

        def clip(params, max_norm):
            n = real_clip(params, max_norm)
            norms.append(float(n))
            return n
        torch.nn.utils.clip_grad_norm_ = clip
        try:
            losses = reference_train_steps(model, opt, CrossEntropyLoss(), ep, meta_steps=4, steps=3, accum=2, stage="multi")
        finally:
            torch.nn.utils.clip_grad_norm_ = real_clip
        torch.cuda.synchronize()
        assert m.auto_stats["opened"] == 4 and m.auto_stats["closed_by"] == {"next_episode": 2, "parameters": 2}, m.auto_stats
        assert not m._auto_open and opt.step_count == 2 and all(n > 0 for n in norms)
        assert float(m.store.grad["lm"].float().abs().max()) == 0.0
        return losses, norms, {g: t.detach().clone() for g, t in m.store.param.items()}
    l0, n0, p0 = run(False)
    l1, n1, p1 = run(True)
    assert l0 == l1 and n0 == n1
    for g in p0:
        assert torch.equal(p0[g], p1[g]), g


def test_automatic_episode_guards_and_opt_out(monkeypatch):
    from navillm_amd.synthetic import SyntheticEpisodes, reference_rollout
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.nav_model import NavModel
    from navillm_amd.optim import FlatAdamW
    cfg = _cfg("mid")
    crit = CrossEntropyLoss()
    m = _model(cfg, auto=True)
    ep = SyntheticEpisodes(cfg, 2, seed=8, instr_len=80, device=torch.device(DEV))
    reference_rollout(m, crit, ep, 2)
    # nothing has read the gradients: zeroing them now would leak the episode into the next optimizer step -> loud
    with pytest.raises(RuntimeError, match="automatic prefix-reuse episode still holds"):
        m.zero_grad()
    opt = FlatAdamW(m, lr=0.0)                 # (its constructor lists the parameters: that alone hands the episode over)
    assert not m._auto_open
    m.zero_grad()
    # FlatAdamW's own clip closes an automatic episode as well
    ep.reset()
    reference_rollout(m, crit, ep, 2)
    assert m._auto_open
    opt.clip_grad_norm_(40.0)
    assert not m._auto_open and m.auto_stats["closed_by"].get("optimizer") == 1
    opt.step(); opt.zero_grad()
    # an explicit begin_episode() takes precedence over an open automatic episode
    ep.reset()
    reference_rollout(m, crit, ep, 1)
    assert m._auto_open
    ep2 = SyntheticEpisodes(cfg, 2, seed=9, instr_len=70, device=torch.device(DEV))
    m.begin_episode(ep2.prefix_ids(), teacher_forced=True)
    assert not m._auto_open and m.auto_stats["closed_by"].get("begin_episode") == 1
    m.episode_abort()
    m.zero_grad()
    # eval mode / no_grad / a batch that does not tell its prefix: the full path, no episode
    m.eval()
    ep.reset()
    reference_rollout(m, crit, ep, 1)
    assert not m._auto_open
    m.train()
    m.zero_grad()
    # the environment knob
    monkeypatch.setenv("NAVILLM_AUTO_EPISODE", "0")
    off = NavModel(nav_config=cfg, device=torch.device(DEV), seed=21)
    assert off.auto_episode is False
    monkeypatch.delenv("NAVILLM_AUTO_EPISODE")
    on = NavModel(nav_config=cfg, device=torch.device(DEV), seed=21)
    assert on.auto_episode is True
