"""GPU: episodes of UNLIKE forms back to back on one model leave nothing behind for one another.

Fine-tuning alternates teacher-forced and sampled meta-steps on one model (tasks/agents/mp3d_agent.py:509-525), i.e. the three
episode forms of navillm_amd/episode.py -- per-step forward with the deferred batched backward, teacher-forced with the batched
forward, and the accumulation window -- run one after the other over the SAME persistent row buffers, statistics slabs, K/V slabs and
scratch.  Each episode of that sequence must produce, bit for bit, the logits and gradient buffers the same episode produces on a
FRESH model (fresh process state as far as the episode is concerned: newly allocated buffers, nothing recorded).  Runs in training mode
(in-kernel dropout, re-keyed per step) and also passes with NAVILLM_POISON=1, where everything an episode leaves behind is NaN."""
import os
import subprocess
import sys

import pytest
import torch

from test_round2_gpu import _mid_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(size):
    from navillm_amd import config as nvcfg
    return _mid_cfg() if size == "mid" else nvcfg.vicuna_7b(image_feat_size=768, num_layers=2, base_vocab_size=2000)


def _model(cfg):
    from navillm_amd.nav_model import NavModel
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=12)
    m.train()
    return m


def _run(m, cfg, kind, seed):
    """one episode (or one 2-episode accumulation window) of `kind` -> (per-step logits on the host, flat gradient clones)"""
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    from navillm_amd.losses import CrossEntropyLoss
    crit = CrossEntropyLoss()
    m.zero_grad()
    m.store.touched.clear()
    plan = {"per_step": [(2, 3, 150)], "teacher": [(2, 2, 171)], "window": [(1, 2, 96), (1, 3, 133)]}[kind]
    handles = []
    for e, (B, steps, il) in enumerate(plan):
        ep = SyntheticEpisodes(cfg, B, seed=seed + e, instr_len=il, device=torch.device(DEV))
        if B > 1:
            ep.instr[1] = ep.instr[1][: il - 19]                    # ragged prefixes
        if kind == "window":
            m.begin_episode(ep.prefix_ids(), teacher_forced=True, accumulate=len(plan))
        else:
            m.begin_episode(ep.prefix_ids(), teacher_forced=(kind == "teacher"))
        for t in range(steps):
            torch.manual_seed(4000 + 97 * e + t)                    # candidate permutation + the dropout keys of this step
            _, lg = nav_step(m, crit, ep, train=True, last=(t == steps - 1), accum=len(plan))
            handles.append(lg)
        m.finish_episode()
    torch.cuda.synchronize()
    logits = [(lg.value if hasattr(lg, "value") else lg).detach().float().cpu() for lg in handles]
    grads = {g: t.detach().clone() for g, t in m.store.grad.items()}
    for g, t in grads.items():
        assert bool(torch.isfinite(t.float()).all()), f"{kind}: non-finite values in the {g} gradient buffer"
    for lg in logits:
        assert not bool(torch.isnan(lg).any()), f"{kind}: NaN logits"
    return logits, grads


SEQUENCE = [("per_step", 61), ("teacher", 62), ("window", 63), ("per_step", 64), ("window", 65), ("teacher", 66)]


@pytest.mark.parametrize("size", ["mid", "7b-width"])
def test_unlike_episodes_back_to_back_equal_the_same_episodes_on_fresh_models(size):
    cfg = _cfg(size)
    shared = _model(cfg)
    got = [_run(shared, cfg, kind, seed) for kind, seed in SEQUENCE]
    shared.episode_release()
    del shared
    torch.cuda.empty_cache()
    for (kind, seed), (lg_s, g_s) in zip(SEQUENCE, got):
        fresh = _model(cfg)
        lg_f, g_f = _run(fresh, cfg, kind, seed)
        del fresh
        assert len(lg_f) == len(lg_s)
        for t, (a, b) in enumerate(zip(lg_s, lg_f)):
            assert torch.equal(a, b), f"{size}: {kind} (seed {seed}) step {t}: logits differ from the fresh model's, max |d| {(a - b).abs().max().item():.3e}"
        for g in g_f:
            assert torch.equal(g_s[g], g_f[g]), (f"{size}: {kind} (seed {seed}): gradient buffer {g} differs from the fresh model's, rel "
                                                 f"{((g_s[g].float() - g_f[g].float()).norm() / (g_f[g].float().norm() + 1e-30)).item():.3e}")


def test_isolation_and_episode_parity_under_poisoned_buffers():
    """the same test, the one-launch attention identities and the prefix-vs-recompute parity once more in a child process with
    NAVILLM_POISON=1 (navillm_amd/debug.py): NaN-filled allocations, canaries, everything re-poisoned at the end of every episode"""
    if os.environ.get("NAVILLM_POISON", "0") not in ("0", ""):
        pytest.skip("already running under NAVILLM_POISON=1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NAVILLM_POISON="1")
    sel = ["tests/test_episode_isolation_gpu.py::test_unlike_episodes_back_to_back_equal_the_same_episodes_on_fresh_models",
           "tests/test_episode_gpu.py::test_episode_forward_attention_one_launch_equals_the_per_step_cache_form",
           "tests/test_episode_gpu.py::test_teacher_forced_episode_batches_the_forward_and_matches",
           "tests/test_parity_gpu.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + sel, cwd=root, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1500:]
    assert r.returncode == 0, tail
    assert "navillm_amd poison:" in r.stdout and " 0 guarded allocations" not in r.stdout, tail


@pytest.mark.parametrize("size", ["mid", "7b-width"])
def test_overlapped_optimizer_update_equals_the_serial_update(size):
    """round 6: FlatAdamW.step() runs the LM group's update on a side stream, decoder layer by decoder layer, while the launch stream
    goes on with the next episode's scene-encoder steps; every consumer of an LM parameter / gradient waits for the part it needs
    (FlatStore.wait_params).  The same training loop -- teacher-forced episodes, per-step-forward episodes, a recompute step, an
    immediate no-grad navigation step and a state_dict() read right after step() -- with the overlap on and off: bit-identical
    logits, parameters and optimizer moments."""
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step, prefix_reuse_episode
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.optim import FlatAdamW
    cfg = _cfg(size)
    crit = CrossEntropyLoss()

    def run(overlap):
        m = _model(cfg)
        opt = FlatAdamW(m, lr=3e-3)
        opt.overlap_update = overlap
        ep = SyntheticEpisodes(cfg, 2, seed=91, instr_len=120, device=torch.device(DEV))
        seen = []
        for k, form in enumerate(["teacher", "teacher", "per_step", "recompute", "teacher"]):
            ep.reset()
            torch.manual_seed(700 + k)
            if form == "recompute":
                nav_step(m, crit, ep, train=True, last=True)
            else:
                prefix_reuse_episode(m, crit, ep, 2, teacher_forced=(form == "teacher"))
            opt.clip_grad_norm_(40.0)
            opt.step()
            opt.zero_grad()
            if overlap and k == 0:
                assert m.store._upd is not None, "the LM group's update should still be pending on the side stream's events"
            if k == 1:
                # consumers right behind the step: a no-grad navigation step (full LM path) ...
                m.eval()
                with torch.no_grad():
                    _, lg = nav_step(m, crit, ep, train=False)
                seen.append(lg.detach().float().cpu())
                m.train()
            if k == 2:
                # ... and the checkpoint writer
                sd = m.state_dict()
                seen.append(sd["lang_model.model.layers.1.mlp.down_proj.weight"].detach().float().cpu())
                seen.append(sd["out_head.0.weight"].detach().float().cpu())
        torch.cuda.synchronize()
        return seen, {g: t.detach().clone() for g, t in m.store.param.items()}, {g: t.detach().clone() for g, t in m.store.exp_avg_sq.items()}
    s1, p1, v1 = run(True)
    s0, p0, v0 = run(False)
    for a, b in zip(s1, s0):
        assert torch.equal(a, b)
    for g in p0:
        assert torch.equal(p1[g], p0[g]), ("param", g)
        assert torch.equal(v1[g], v0[g]), ("exp_avg_sq", g)
