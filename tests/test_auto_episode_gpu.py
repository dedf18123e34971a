"""GPU: AUTOMATIC episodes -- the fast prefix-reuse path behind the reference's own call pattern.

`navillm_amd.synthetic.reference_rollout` / `reference_train_steps` restate `MP3DAgent.rollout` (tasks/agents/mp3d_agent.py:660-778)
and `train_one_epoch` (train.py:60-91) call for call and contain NO `begin_episode` / `finish_episode`: the model opens the episode
on the first grad-enabled training-mode navigation call and hands its gradients over when `torch.nn.utils.clip_grad_norm_(
model.parameters(), 40.)`, the optimizer, or the next rollout's first navigation call comes.  Two forms (NavModel.auto_form):
  "lazy" (default): `fuse_logits` is a `losses.LazyLogits` handle; a teacher-forced rollout never reads it, so the LM forward of all its
          steps runs as one batch -- bit-identical to an explicit `begin_episode(..., teacher_forced=True)` episode; a rollout that
          samples (reads the logits at every step) forces step by step and stays within the per-step-forward form's tolerances;
  "step": every step's forward runs when called -- bit-identical to an explicit `teacher_forced=False` episode.
Also: the closing triggers, the loud failures (zero_grad over pending gradients), the training loop of train.py verbatim with the
model bare and inside NavDataParallel.  (The reference-pinned G12 episode through the automatic path:
tests/test_parity_gpu.py::test_g12_...[auto].)"""
import pytest
import torch

from util import bf16_ulps_at_scale
from test_round2_gpu import _mid_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(size):
    from navillm_amd import config as nvcfg
    return _mid_cfg() if size == "mid" else nvcfg.vicuna_7b(image_feat_size=768, num_layers=2, base_vocab_size=2000)


def _model(cfg, auto, form="lazy"):
    from navillm_amd.nav_model import NavModel
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=21)
    m.train()
    m.auto_episode, m.auto_form = auto, form
    return m


def _rollouts(m, cfg, plan, explicit, accum):
    """the rollouts of `plan` [(B, steps, instruction length, feedback)] one after the other as the training loop runs them; explicit:
    None = nothing but the reference's calls, else every rollout wrapped in begin_episode(teacher_forced=explicit) / finish_episode.
    -> (the fuse_logits every navigation call returned, the loss of every rollout as the loop reads it)"""
    from navillm_amd.synthetic import SyntheticEpisodes, reference_rollout
    from navillm_amd.losses import CrossEntropyLoss
    crit = CrossEntropyLoss()
    m.zero_grad()
    m.store.touched.clear()
    seen, losses = [], []
    orig = m.forward_navigation

    def spy(mode, batch, **kw):
        out = orig(mode, batch, **kw)
        seen.append(out["fuse_logits"])
        return out
    m.forward_navigation = spy
    try:
        for e, (B, steps, il, fb) in enumerate(plan):
            ep = SyntheticEpisodes(cfg, B, seed=300 + e, instr_len=il, device=torch.device(DEV))
            if B > 1:
                ep.instr[1] = ep.instr[1][: il - 13]
            torch.manual_seed(9000 + e)
            if explicit is not None:
                m.begin_episode(ep.prefix_ids(), teacher_forced=explicit)
            loss = reference_rollout(m, crit, ep, steps, feedback=fb.split("+")[0], accum=accum, follow_teacher=fb.endswith("+teacher"))
            if explicit is not None:
                m.finish_episode()
            losses.append(float(loss))                       # train.py:83 `loss.item()`
    finally:
        m.forward_navigation = orig
    return seen, losses


def _values(handles):
    torch.cuda.synchronize()
    return [(h.value if hasattr(h, "value") else h).detach().float().cpu() for h in handles]


def _grads(m):
    torch.cuda.synchronize()
    return {g: t.detach().clone() for g, t in m.store.grad.items()}


@pytest.mark.parametrize("size,form", [("mid", "lazy"), ("7b-width", "lazy"), ("mid", "step"), ("7b-width", "step")])
def test_unmodified_rollout_is_bit_identical_to_the_explicit_episode_form(size, form):
    from navillm_amd.losses import LazyLogits
    cfg = _cfg(size)
    # lazy: imitation learning (all of pre-training, every other meta-step of fine-tuning); step: any feedback
    plan = [(2, 3, 140, "teacher"), (2, 2, 111, "teacher" if form == "lazy" else "sample+teacher"), (2, 3, 165, "teacher")]
    a = _model(cfg, auto=True, form=form)
    h_a, ls_a = _rollouts(a, cfg, plan, explicit=None, accum=len(plan))
    assert a._auto_open and a.episode.has_pending_gradients()
    assert a.auto_stats["opened"] == 3 and a.auto_stats["closed_by"] == {"next_episode": 2}
    if form == "lazy":
        assert all(isinstance(h, LazyLogits) for h in h_a) and a.episode.stats.get("forced_reads", 0) == 0
    else:
        assert all(torch.is_tensor(h) for h in h_a)
    # train.py:87's clip (a bound that never clips) hands the last episode over
    torch.nn.utils.clip_grad_norm_(a.parameters(), 1e9)
    assert not a._auto_open and a.episode.prefix is None and a.auto_stats["closed_by"] == {"next_episode": 2, "parameters": 1}
    lg_a, g_a = _values(h_a), _grads(a)
    del a
    b = _model(cfg, auto=False)
    h_b, ls_b = _rollouts(b, cfg, plan, explicit=(form == "lazy"), accum=len(plan))
    lg_b, g_b = _values(h_b), _grads(b)
    assert len(lg_a) == len(lg_b) == sum(p[1] for p in plan)
    for t, (x, y) in enumerate(zip(lg_a, lg_b)):
        assert torch.equal(x, y), f"{size}/{form}: logits of navigation call {t} differ: {(x - y).abs().max().item():.3e}"
    assert ls_a == ls_b
    for g in g_b:
        assert torch.equal(g_a[g], g_b[g]), (size, form, g, ((g_a[g].float() - g_b[g].float()).norm() / (g_b[g].float().norm() + 1e-30)).item())
    # and it is the prefix-reuse path that ran, not the full-prompt recompute: same episodes with automatic episodes off
    c = _model(cfg, auto=False)
    h_c, _ = _rollouts(c, cfg, plan[:1], explicit=None, accum=len(plan))
    assert c.episode is None
    assert not all(torch.equal(x, y) for x, y in zip(lg_a, _values(h_c))), "the recompute path is a different evaluation (RoPE frame, row order)"


@pytest.mark.parametrize("size", ["mid", "7b-width"])
def test_lazy_logits_read_at_every_step_match_the_per_step_forward_form(size):
    """a DAgger / argmax rollout reads the logits at once (`Categorical(nav_probs.float())`, mp3d_agent.py:762-765): the LazyLogits
    handle then pushes the pending step through the decoder then and there (with the prefix, the first time) and hands back a real
    tensor the rollout's own loss backpropagates through.  Same episode in the explicit per-step-forward form (another forward path:
    the K/V-cache layout): logits within bf16 rounding, gradient buffers within the per-step form's tolerance."""
    cfg = _cfg(size)
    plan = [(2, 3, 150, "sample+teacher"), (2, 2, 120, "sample+teacher")]
    a = _model(cfg, auto=True, form="lazy")
    h_a, ls_a = _rollouts(a, cfg, plan, explicit=None, accum=len(plan))
    assert a.episode.stats.get("forced_reads", 0) == 2 and a.auto_stats["opened"] == 2
    torch.nn.utils.clip_grad_norm_(a.parameters(), 1e9)
    lg_a, g_a = _values(h_a), _grads(a)
    del a
    b = _model(cfg, auto=False)
    h_b, ls_b = _rollouts(b, cfg, plan, explicit=False, accum=len(plan))
    lg_b, g_b = _values(h_b), _grads(b)
    worst = max(bf16_ulps_at_scale(x, y) for x, y in zip(lg_a, lg_b))
    rel = {g: ((g_a[g].float() - g_b[g].float()).norm() / (g_b[g].float().norm() + 1e-30)).item() for g in g_b}
    lrel = max(abs(x - y) / max(abs(y), 1e-6) for x, y in zip(ls_a, ls_b))
    print(f"[lazy logits read per step, {size}] logits worst {worst:.2f} bf16 spacings vs the per-step-forward form, loss values rel {lrel:.2e}, "
          f"gradient buffers rel {rel}")
    assert worst <= (2.0 if size == "mid" else 3.0) and lrel < 2e-2
    for g, v in rel.items():
        assert v < 2.5e-2, (g, v)


def test_reference_training_loop_with_flat_adamw_and_with_the_wrapper():
    """train.py:60-91 verbatim (`reference_train_steps`): torch's clip through `model.parameters()`, FlatAdamW.step / zero_grad, B = 1 x 2
    accumulation, teacher forcing alternating with sampling; then the same loop with the model inside NavDataParallel (world of one):
    `wrapped.parameters()` must hand the episode over too (nn.Module reads the child's `_parameters` directly).  The two
    runs end with identical parameters."""
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
        assert not m._auto_open and opt.step_count == 2 and all(n > 0 for n in norms) and all(l == l and l > 0 for l in losses)
        assert float(m.store.grad["lm"].float().abs().max()) == 0.0
        return losses, norms, {g: t.detach().clone() for g, t in m.store.param.items()}
    l0, n0, p0 = run(False)
    l1, n1, p1 = run(True)
    assert l0 == l1 and n0 == n1
    for g in p0:
        assert torch.equal(p0[g], p1[g]), g


def test_automatic_episode_guards_and_opt_out(monkeypatch):
    from navillm_amd.synthetic import SyntheticEpisodes, reference_rollout
    from navillm_amd.losses import CrossEntropyLoss, LazyLogits
    from navillm_amd.nav_model import NavModel
    from navillm_amd.optim import FlatAdamW
    cfg = _cfg("mid")
    crit = CrossEntropyLoss()
    m = _model(cfg, auto=True)
    ep = SyntheticEpisodes(cfg, 2, seed=8, instr_len=80, device=torch.device(DEV))
    loss = reference_rollout(m, crit, ep, 2)
    # nothing has read the gradients: zeroing them now would leak the episode into the next optimizer step -> loud
    with pytest.raises(RuntimeError, match="automatic prefix-reuse episode still holds"):
        m.zero_grad()
    assert float(loss) > 0 and m._auto_open              # reading the loss runs the batched forward + heads, the episode stays open
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
    # a handle answers shape questions without forcing, a tensor method forces; after the episode is over it still has its value
    ep.reset()
    pin = ep.panorama_inputs()
    pano = m("panorama", pin)
    ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
    nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
    nav["input_ids"], nav["attention_mask"] = ep.tokenise(nav, "<cls_1>")
    lg = m("navigation", nav)["fuse_logits"]
    assert isinstance(lg, LazyLogits) and lg.shape == (2, nav["gmap_masks"].shape[1]) and lg.dim() == 2 and lg.size(0) == 2
    probs = torch.softmax(lg / 0.5, 1)
    assert m.episode.stats.get("forced_reads", 0) == 0
    _, a_t = lg.max(1)                                    # mp3d_agent.py:767 (argmax feedback)
    assert m.episode.stats["forced_reads"] == 1 and a_t.shape == (2,)
    assert torch.allclose(probs.float().sum(1), torch.ones(2, device=DEV), atol=1e-2)
    (crit(lg, ep.teacher_targets(nav, False).to(DEV)) / 2).backward()
    # an explicit begin_episode() takes precedence over an open automatic episode
    assert m._auto_open
    ep2 = SyntheticEpisodes(cfg, 2, seed=9, instr_len=70, device=torch.device(DEV))
    m.begin_episode(ep2.prefix_ids(), teacher_forced=True)
    assert not m._auto_open and m.auto_stats["closed_by"].get("begin_episode") == 1
    assert torch.isfinite(lg.value[torch.isfinite(lg.value)]).all()
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
    monkeypatch.setenv("NAVILLM_AUTO_EPISODE", "step")
    st = NavModel(nav_config=cfg, device=torch.device(DEV), seed=21)
    assert st.auto_episode is True and st.auto_form == "step"
    monkeypatch.delenv("NAVILLM_AUTO_EPISODE")
    on = NavModel(nav_config=cfg, device=torch.device(DEV), seed=21)
    assert on.auto_episode is True and on.auto_form == "lazy"


def test_wrapped_model_clip_hands_over_a_partial_accumulation_window():
    """`torch.nn.utils.clip_grad_norm_(model.parameters(), 40.)` on the WRAPPED model (train.py:87 after
    tools/optims.py:52-54) used to walk the child's `_parameters` without running `NavModel.parameters()`: an accumulation window that
    was still open (fewer episodes than `accumulate`: normal in a multi-task mix) kept its gradients, the clip norm was taken without
    them and `FlatAdamW.step()` added them after the clip.  `NavDataParallel.named_parameters` hands the window over first."""
    from test_episode_gpu import _window_run
    from navillm_amd.parallel import NavDataParallel
    cfg = _cfg("mid")
    plan = [(41, 150, 3), (42, 96, 2)]
    norms = {}

    def run(wrapped_clip):
        m = _model(cfg, auto=False)
        m.eval()
        w = NavDataParallel(m, comm=None)

        def flush():
            assert m._window is not None and m._window.window_open(), "two of four episodes: the window is still open"
            if wrapped_clip:
                norms[wrapped_clip] = float(torch.nn.utils.clip_grad_norm_(w.parameters(), 1e9))
            else:
                m.flush_accumulation_window()
                norms[wrapped_clip] = float(torch.nn.utils.clip_grad_norm_(m.parameters(), 1e9))
            assert not m._window.window_open()
        _, losses, grads = _window_run(m, cfg, plan, accumulate=4, flush=flush)
        return losses, grads
    l1, g1 = run(True)
    l0, g0 = run(False)
    assert l1 == l0 and norms[True] == norms[False] and norms[True] > 0
    for g in g0:
        assert torch.equal(g1[g], g0[g]) and float(g1[g].abs().max()) > 0, g
