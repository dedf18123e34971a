#!/usr/bin/env python3
"""Headline benchmark: navigation-steps/sec (whole job) of the NaviLLM training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one iteration of the rollout loop (tasks/agents/mp3d_agent.py:660) for the rank's batch of
B=8 synthetic R2R-shaped episodes: model('panorama') + model('navigation') + action CE + full
backward through Vicuna-7B; every 6th step ends the episodes and runs clip(40)+AdamW+zero_grad
(train.py:86-89) inside the timed region.  value = B * K * N / max-over-ranks wall time.
Workload = BASELINE.json configs[1]: "Vicuna-7B + 36-view features, R2R synthetic batch=8, 1xMI355X bf16".
Weak scaling: per-GPU work is fixed, gradients are all-reduced (RCCL) on the last step of each episode.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STEPS_PER_EPISODE = 6          # T=6 steps per R2R-shaped episode
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--model", default="vicuna-7b", choices=["vicuna-7b", "vicuna-13b", "tiny"])
    ap.add_argument("--feat", type=int, default=768, help="view feature size (768 per BASELINE.json, 1024 = EVA-CLIP-L)")
    ap.add_argument("--instr-len", type=int, default=512)
    ap.add_argument("--lr", type=float, default=3e-5)
    ap.add_argument("--prewarm", type=int, default=6, help="untimed setup steps before the W warmup steps (one full episode: "
                    "first-use kernel/attribute/allocator/RCCL initialisation)")
    ap.add_argument("--infer-steps", type=int, default=6, help="extra, untimed-for-`value` forward-only steps reported aside")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras-on-tiny", action="store_true", help="run the extras with --model tiny too (rehearsals)")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-3 (mixed tasks) and config-4 (64-step) extra measurements")
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch GEMM event timing")
    ap.add_argument("--no-tf-batch", action="store_true", help="prefix_reuse mode: run every step's LM forward when the step is called (round-3 form) "
                    "instead of batching the teacher-forced episode's forward into finish_episode()")
    ap.add_argument("--no-other-mode", action="store_true", help="do not measure the other training mode (kernel traces of one mode)")
    ap.add_argument("--explicit-episodes", action="store_true", help="prefix_reuse mode: call begin_episode(..., teacher_forced=...) / finish_episode() around "
                    "every episode (rounds 3-5) instead of letting the model open and close its episodes itself (round 6: automatic episodes, "
                    "the default -- the timed loop then contains nothing but the reference's calls)")
    ap.add_argument("--mode", default=os.environ.get("NAVILLM_BENCH_MODE", "prefix_reuse"), choices=["prefix_reuse", "recompute"],
                    help="how the training step treats the prompt's static prefix (instruction + template, ~530 of ~650 tokens): "
                         "prefix_reuse (default) = forward once per episode, K/V reused by every step, one deferred prefix backward "
                         "(navillm_amd/episode.py; gradients pinned to the reference by fixture G12); recompute = the whole prompt "
                         "through the LM at every step like the reference's rollout.  The other mode is measured too and reported "
                         "under `other_mode`.")
    return ap.parse_args()


def make_cfg(a):
    from navillm_amd import config as C
    if a.model == "vicuna-7b":
        return C.vicuna_7b(image_feat_size=a.feat)
    if a.model == "vicuna-13b":
        return C.vicuna_13b(image_feat_size=a.feat)
    return C.tiny(image_feat_size=a.feat)


class GemmTimer:
    """HIP events around the bf16 GEMM launches, on the stream the kernels are launched on.  Only while `active`: bracketing
    all 384 GEMM launches of every step with two events each costs the step 0.9 % (43.30 vs 43.68 nav-steps/s, ABAB on one box), so
    the timed region instruments ONE WHOLE EPISODE (its second one, steps 6..11 of the window; the steps of an episode differ a lot
    in the prefix-reuse mode: the first carries the prefix forward, the last the prefix backward and the per-weight wgrad GEMMs).
    Shorter windows (< 12 steps) instrument every step."""

    @staticmethod
    def sampled(k, steps):
        """is step k (0-based inside the timed window of `steps` steps) instrumented?"""
        return STEPS_PER_EPISODE <= k < 2 * STEPS_PER_EPISODE if steps >= 2 * STEPS_PER_EPISODE else True

    @staticmethod
    def n_sampled(steps):
        return STEPS_PER_EPISODE if steps >= 2 * STEPS_PER_EPISODE else steps

    def __init__(self):
        self.recs = []
        self.active = True

    def install(self, ops):
        self.ops, self.orig = ops, ops.gemm_bf16
        timer = self

        def timed(layout, A, B, out=None, R=None, epilogue=0, tile_cfg=0):
            if not timer.active:
                return timer.orig(layout, A, B, out=out, R=R, epilogue=epilogue, tile_cfg=tile_cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = timer.orig(layout, A, B, out=out, R=R, epilogue=epilogue, tile_cfg=tile_cfg)
            e1.record()
            M, N = r.shape
            K = A.shape[1] if layout != 2 else A.shape[0]
            # algorithmic bytes of the launch: both operands read once, C written once (+ read once by the residual / accumulate epilogues)
            byts = 2.0 * (M * K + N * K + M * N) + (2.0 * M * N if (R is not None or epilogue == ops.EPI_ACCUM) else 0.0)
            timer.recs.append((2.0 * M * N * K, e0, e1, layout, byts))
            return r
        ops.gemm_bf16 = timed
        self.orig_rope = ops.gemm_qkv_rope

        def timed_rope(x, W, cos_t, sin_t, S, rope_cols, out=None, pos_i32=None):     # the q|k|v projection (NT + RoPE epilogue)
            if not timer.active:
                return timer.orig_rope(x, W, cos_t, sin_t, S, rope_cols, out=out, pos_i32=pos_i32)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = timer.orig_rope(x, W, cos_t, sin_t, S, rope_cols, out=out, pos_i32=pos_i32)
            e1.record()
            timer.recs.append((2.0 * r.shape[0] * r.shape[1] * x.shape[1], e0, e1, 0,
                               2.0 * (r.shape[0] * x.shape[1] + r.shape[1] * x.shape[1] + r.shape[0] * r.shape[1])))
            return r
        ops.gemm_qkv_rope = timed_rope

    def uninstall(self):
        self.ops.gemm_bf16 = self.orig
        self.ops.gemm_qkv_rope = self.orig_rope

    def summary(self, layouts=(0,)):
        """Event brackets are kernel durations as long as the launches run alone on their stream: always true for the
        forward (NT) GEMMs, and for dgrad (NN) / wgrad (TN) with the default single-stream backward."""
        recs = [r for r in self.recs if r[3] in layouts]
        if not recs:
            return None
        fl = sum(r[0] for r in recs)
        sec = sum(r[1].elapsed_time(r[2]) for r in recs) * 1e-3
        return {"launches": len(recs), "flops_per_launch": fl / len(recs), "bytes_per_launch": sum(r[4] for r in recs) / len(recs),
                "avg_launch_ms": sec / len(recs) * 1e3, "tflops": fl / sec / 1e12, "gemm_seconds": sec}


def gemm_traffic_from_profile(mode):
    """HBM-side bytes per GEMM launch of training mode `mode` from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 per
    the gfx950 correction + WRITE_SIZE; separate --pmc runs of this same command, tools/gpu_pmc_bench_r4.sh: two whole episodes of the
    mode, so the average is over the mode's own launch mix).  -> (bytes or None, note).  A profile is
    only used when it was taken from THIS kernel source: tools/pmc_traffic.py records the sha256 of gemm_bf16.hip, and a file
    whose hash differs (the kernel changed and the counters were not collected again) is refused -- `traffic` is then null
    rather than stale."""
    import hashlib
    try:
        with open(os.path.join(ROOT, "navillm_amd", "csrc", "gemm_bf16.hip"), "rb") as f:
            cur = hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None, "kernel source not found"
    stale = []
    names = (f"r06_gemm_traffic_{mode}.json", f"r05_gemm_traffic_{mode}.json", f"r04_gemm_traffic_{mode}.json") + (("r03_gemm_traffic.json", "r02_gemm_traffic.json", "r01_gemm_traffic.json") if mode == "recompute" else ())
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
        except Exception:
            continue
        if d.get("gemm_source_sha256") != cur:
            stale.append(name)
            continue
        return d.get("hbm_bytes_per_gemm_launch_all_layouts", d["hbm_bytes_per_forward_gemm_launch"]), f"profiles/{name}"
    return None, ("no PMC profile of the current gemm_bf16.hip under profiles/ (older ones refused: " + ", ".join(stale) + ")") if stale \
        else "no PMC profile under profiles/"


MODE_WHAT = {
    "prefix_reuse": "per 6-step episode the prompt's static prefix (task sentence + instruction + history header, ~530 of ~650 tokens) goes "
                    "through the LM forward ONCE (K/V cached per layer); every nav step pushes only its suffix rows (history, candidates, "
                    "hints, <cls_1>) forward over the cache and its backward() records the gradient of its B output rows; finish_episode() then "
                    "walks the layers ONCE for every token row of the episode (prefix + all steps' rows: dgrad and ONE weight-gradient GEMM "
                    "per weight over ~7 700 rows, the attention backward of all steps in one launch per kernel, each step's visual-token gradient sent "
                    "into that step's scene-encoder graph) (navillm_amd/episode.py).  Same losses and gradients as the per-step recompute up to bf16 rounding order -- the weights "
                    "are frozen inside an episode (train.py:86-89) -- pinned to the reference's own 3-step episode by fixture G12 "
                    "(tests/test_parity_gpu.py::test_g12_episode_accumulated_gradients_vs_reference[prefix_reuse])",
    "recompute": "the whole ~650-token prompt goes through the LM forward and backward at every nav step, as the reference's rollout "
                 "does it (tasks/agents/mp3d_agent.py:726,756)",
}


AUTO_WHAT = ("; round 6: none of this is called from the timed loop -- the model opens the episode itself on the first training-mode "
             "navigation call (NavModel._auto_*), hands out LazyLogits handles, and optimizer.clip_grad_norm_ runs the deferred work "
             "(bit-identical to the explicit begin_episode / finish_episode form, tests/test_auto_episode_gpu.py)")
TF_WHAT = ("; round 4: the rollout is imitation learning (teacher forcing: the next action is the teacher's, the next history token the "
           "fusion output -- neither depends on the LM, tasks/agents/mp3d_agent.py:760-761,774-778), so the steps' LM FORWARD is deferred too: "
           "model('navigation') returns fuse_embeds at once and a deferred-logits handle, criterion(handle, targets) * w / B and .backward() "
           "record the loss, and finish_episode() pushes the suffix rows of ALL six steps through the decoder as ONE batch (~4 200 rows per "
           "GEMM instead of six launches of ~700) before the batched backward (begin_episode(..., teacher_forced=True); sampling / argmax "
           "rollouts keep the per-step forward)")


def roofline_of(timer, dt, steps, mode, model):
    """the `roofline` object for the bf16 GEMM family over the launches the timer bracketed (HIP events on the launch stream)"""
    g = timer.summary()
    if g is None:
        return None
    allg = timer.summary(layouts=(0, 1, 2))
    n_sampled = GemmTimer.n_sampled(steps)           # timed steps that carried the events: one whole episode
    # the dominant kernel is ONE template (gemm_bf16_kernel) in three operand layouts.  With the backward on a single stream (the
    # default) every launch's event bracket is its kernel duration, so the roofline is taken over all of them; with the optional
    # wgrad side stream only the forward launches run alone.
    r = g if model.overlap_wgrad else allg
    per = {n: timer.summary(layouts=(l,)) for n, l in (("forward_NT", 0), ("dgrad_NN", 1), ("wgrad_TN", 2))}
    traffic, traffic_src = gemm_traffic_from_profile(mode)
    return {"bound": "mfma", "achieved": round(r["tflops"], 1), "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(r["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": round(r["bytes_per_launch"]),
            "traffic_over_algorithmic": (round(traffic / r["bytes_per_launch"], 2) if traffic else None),
            "kernel": "gemm_bf16_kernel<256,256,2,4,64,2,*,*,*,PIPE,TME> -- every bf16 GEMM launch of one whole episode of the timed region "
                      "(%d of the %d steps: forward y = x W^T, dgrad, wgrad of qkv / o / gate|up / down in every layer; in prefix_reuse mode the "
                      "suffix steps' few-hundred-row GEMMs run the cut-off 128..224 x 256 tiles; bracketing all steps costs the step 0.9 %%)"
                      % (n_sampled, steps)
                      if not model.overlap_wgrad else
                      "gemm_bf16_kernel<256,256,2,4,64,2,true,true,*,4,*> (forward launches only: with the wgrad side stream the backward brackets overlap)",
            "launches": r["launches"], "avg_launch_ms": round(r["avg_launch_ms"], 4),
            "flops_per_launch": r["flops_per_launch"],
            "by_layout_tflops": {n: (round(v["tflops"], 1) if v else None) for n, v in per.items()},
            "all_gemm_flops_per_step": allg["flops_per_launch"] * allg["launches"] / n_sampled,
            "gemm_share_of_step": round(allg["gemm_seconds"] / n_sampled / (dt / steps), 3),
            "note": "peak = nominal dense bf16 MFMA (2.5 PF).  Sustained ceiling, ONE statement (DESIGN.md §8 item 2): with N(0,1) operands the part "
                    "clocks to ~1.71 GHz inside this kernel's loop (tools/gemm_clock.py), where the dense bf16 MFMA peak is 1.79 PF; register-only "
                    "micro-benchmarks on the same data reach 1.85-2.05 PF (tools/ubench/mfma_rate.hip); the fastest GEMM measured on this box in any "
                    "implementation is a hipBLASLt cell at 1.52 PF (profiles/r05_gemm_vs_blaslt.txt); traffic = 2*FETCH_SIZE+WRITE_SIZE per launch from the "
                    "separate rocprofv3 --pmc passes of THIS training mode committed under profiles/ (`traffic_source`; null when that profile was taken from "
                    "another version of gemm_bf16.hip); it counts the L2s' fabric requests, Infinity-Cache hits included (MI355X_MICROARCH.md "
                    "§HBM), so it is an upper bound of the HBM bytes; algorithmic_bytes_per_launch = operands read once + C written once "
                    "(+ C / R read once by the accumulate / residual epilogues)"}


def inference_extras(a, model, wrapped, crit, ep):
    """untimed-for-`value` extras: the same nav step without loss/backward (validation rollout, mp3d_agent.py:530-590),
    plain and with prompt-prefix K/V reuse"""
    from navillm_amd.synthetic import nav_step
    model.eval()
    ep.reset()
    with torch.no_grad():
        for i in range(STEPS_PER_EPISODE):      # one whole warm episode: every prompt length of the timed one has been seen
            nav_step(wrapped, crit, ep, train=False)
        ep.reset()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(a.infer_steps):
            nav_step(wrapped, crit, ep, train=False)
        torch.cuda.synchronize()
    infer = {"nav_steps_per_s_per_gpu": round(a.batch * a.infer_steps / (time.perf_counter() - t1), 2),
             "steps": a.infer_steps, "what": "panorama + navigation forward only, argmax actions, eval mode"}
    infer["two_batches_in_flight"] = two_batches_in_flight(a, model, wrapped, ep, a.infer_steps, reuse=False)
    model.enable_kv_cache(a.batch, capacity=1024)
    with torch.no_grad():
        for rep in range(2):                    # one warm episode, one timed episode
            ep.reset()
            model.reset_kv_cache()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(STEPS_PER_EPISODE):
                nav_step(wrapped, crit, ep, train=False)
            torch.cuda.synchronize()
    infer_kv = {"nav_steps_per_s_per_gpu": round(a.batch * STEPS_PER_EPISODE / (time.perf_counter() - t1), 2),
                "steps": STEPS_PER_EPISODE, "last_step_new_tokens": model.kv.last_stats["new"],
                "what": "one whole episode (first step = full prefill), K/V of the prompt prefix reused from step to step"}
    infer_kv["two_batches_in_flight"] = two_batches_in_flight(a, model, wrapped, ep, STEPS_PER_EPISODE)
    return infer, infer_kv


def two_batches_in_flight(a, model, wrapped, ep, steps, reuse=True, **ep_kw):
    """the same K/V-reuse rollout with TWO independent batches of `a.batch` episodes software-pipelined (navillm_amd/synthetic.py:
    rollout_interleaved): run alone a batch alternates ~16 ms of host work with ~20 ms of GPU work per step; two batches hide
    each other's host phase.  Same kernels, same batch per forward, own K/V cache per batch."""
    from navillm_amd.synthetic import SyntheticEpisodes, rollout_interleaved
    from navillm_amd.kvcache import KVCacheLM
    eps = [SyntheticEpisodes(model.cfg, a.batch, seed=ep.seed + 101 * k, instr_len=a.instr_len, device=model.device, **ep_kw) for k in range(2)]
    keep = model.kv
    kvs = [KVCacheLM(model, a.batch, capacity=1024) if reuse else False for _ in eps]
    for rep in range(2):
        for e, kv in zip(eps, kvs):
            e.reset()
            if reuse:
                kv.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rollout_interleaved(wrapped, eps, kvs, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    model.kv = keep
    return {"nav_steps_per_s_per_gpu": round(2 * a.batch * steps / dt, 2), "steps": steps, "batches": 2,
            "what": "two batches of %d episodes, the host phase of one overlapped with the GPU phase of the other" % a.batch}


def algorithmic_flops(cfg, log):
    """F_fwd = tokens * 2 * P_blk + L * 2 * sum_b S_b^2 * d (causal attention) per LM call, x3 with its backward;
    lm_head (2 d V per token) only where the LM-loss modes compute it.  `log` = NavModel.flop_log entries."""
    d, ff, L, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_layers, cfg.vocab_size
    p_blk = L * (4 * d * d + 3 * d * ff)
    tot = 0.0
    for kind, n, sq, grad in log:
        f = (2.0 * p_blk * n + 2.0 * L * d * sq) if kind == "lm" else 2.0 * d * V * n
        tot += f * (3.0 if grad else 1.0)
    return tot


def mixed_task_extra(a, cfg, model, wrapped, opt, crit, device, seed):
    """BASELINE config 3 on this rank: one meta-step per task of the multi-task mix (R2R + fine-grained R2R + summarization;
    REVERIE and SOON with the object-grounding sub-task + summarization; CVDN; a ScanQA batch), each a full episode with every
    backward the reference runs, then clip + AdamW.  One untimed warm pass, one timed pass."""
    from navillm_amd.synthetic import SyntheticEpisodes, mixed_task_episode, qa_step
    eps = {t: SyntheticEpisodes(cfg, a.batch, seed=seed + i, instr_len=(a.instr_len if t != "soon" else min(a.instr_len, 256)),
                                device=device, task=t) for i, t in enumerate(("r2r", "reverie", "soon", "cvdn"))}
    res = {}
    # round 3: the same meta-steps with the navigation steps over the episode's cached prompt prefix (the sub-tasks -- fine-grained R2R,
    # object grounding, summarization, ScanQA -- have prompts of their own and run through the whole LM in either mode)
    pre = None
    for rep in range(2):
        nav_steps = 0
        per_task = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t, ep in eps.items():
            ep.reset()
            t1 = time.perf_counter()
            mixed_task_episode(wrapped, crit, ep, STEPS_PER_EPISODE, prefix_reuse=True, teacher_forced=TF_BATCH)
            opt.clip_grad_norm_(40.0); opt.step(); opt.zero_grad()
            nav_steps += STEPS_PER_EPISODE * a.batch
            torch.cuda.synchronize()
            per_task[t] = round((time.perf_counter() - t1) * 1e3, 1)
        t1 = time.perf_counter()
        qa_step(wrapped, eps["r2r"], sync="final")
        opt.clip_grad_norm_(40.0); opt.step(); opt.zero_grad()
        torch.cuda.synchronize()
        per_task["scanqa"] = round((time.perf_counter() - t1) * 1e3, 1)
        dt = time.perf_counter() - t0
        pre = {"seconds": round(dt, 3), "ms_per_meta_step": per_task, "nav_steps_per_s_per_gpu": round(nav_steps / dt, 2),
               "episodes_per_s_per_gpu": round((4 * a.batch + a.batch) / dt, 2)}
    model.episode_abort()
    model.zero_grad()
    # round 6: the SAME meta-steps with nothing but the reference's calls (no begin_episode / finish_episode): automatic episodes -- the
    # navigation steps run over the cached prefix by themselves (teacher-forced: their forward batched), the sub-task calls take the full path
    auto = None
    try:
        model.auto_episode, model.auto_form = True, "lazy"
        for rep in range(2):
            nav_steps = 0
            per_task = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t, ep in eps.items():
                ep.reset()
                t1 = time.perf_counter()
                mixed_task_episode(wrapped, crit, ep, STEPS_PER_EPISODE)
                opt.clip_grad_norm_(40.0); opt.step(); opt.zero_grad()
                nav_steps += STEPS_PER_EPISODE * a.batch
                torch.cuda.synchronize()
                per_task[t] = round((time.perf_counter() - t1) * 1e3, 1)
            t1 = time.perf_counter()
            qa_step(wrapped, eps["r2r"], sync="final")
            opt.clip_grad_norm_(40.0); opt.step(); opt.zero_grad()
            torch.cuda.synchronize()
            per_task["scanqa"] = round((time.perf_counter() - t1) * 1e3, 1)
            dt = time.perf_counter() - t0
            auto = {"seconds": round(dt, 3), "ms_per_meta_step": per_task, "nav_steps_per_s_per_gpu": round(nav_steps / dt, 2),
                    "episodes_per_s_per_gpu": round((4 * a.batch + a.batch) / dt, 2),
                    "what": "no begin_episode / finish_episode in the loop: automatic episodes (NavModel._auto_*)"}
    except Exception as e:
        auto = {"error": f"{type(e).__name__}: {e}"}
        model.episode_abort()
    model.auto_episode = False
    model.episode_abort()
    model.zero_grad()
    for rep in range(2):
        model.flop_log = []
        nav_steps = 0
        per_task = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t, ep in eps.items():
            ep.reset()
            t1 = time.perf_counter()
            mixed_task_episode(wrapped, crit, ep, STEPS_PER_EPISODE)
            opt.clip_grad_norm_(40.0); opt.step(); opt.zero_grad()
            nav_steps += STEPS_PER_EPISODE * a.batch
            torch.cuda.synchronize()
            per_task[t] = round((time.perf_counter() - t1) * 1e3, 1)
        t1 = time.perf_counter()
        qa_step(wrapped, eps["r2r"], sync="final")
        opt.clip_grad_norm_(40.0); opt.step(); opt.zero_grad()
        torch.cuda.synchronize()
        per_task["scanqa"] = round((time.perf_counter() - t1) * 1e3, 1)
        dt = time.perf_counter() - t0
        fl = algorithmic_flops(cfg, model.flop_log)
        res = {"meta_steps": 5, "seconds": round(dt, 3), "ms_per_meta_step": per_task, "nav_steps_per_s_per_gpu": round(nav_steps / dt, 2),
               "episodes_per_s_per_gpu": round((4 * a.batch + a.batch) / dt, 2),
               "algorithmic_tflops": round(fl / dt / 1e12, 1), "frac_of_mfma_peak": round(fl / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
               "what": f"one episode each of r2r (+fgr2r on steps 0,2,4 +summarization), reverie and soon (+object grounding +summarization), "
                       f"cvdn, {STEPS_PER_EPISODE} nav steps each with per-step backward, and one ScanQA batch; B={a.batch}; clip+AdamW after each; "
                       f"algorithmic FLOPs per SURVEY.md §8d incl. lm_head in the LM-loss modes (top level: the whole prompt recomputed at "
                       f"every nav step; `navigation_over_cached_prefix`: the same meta-steps in prefix_reuse mode)"}
        res["navigation_over_cached_prefix"] = pre
        res["unmodified_rollout_automatic_episodes"] = auto
    model.flop_log = None
    return res


def long_horizon_extra(a, cfg, model, wrapped, crit, device, seed, T=64):
    """BASELINE config 4: a 64-step episode.  (i) the validation rollout with history K/V reuse (what "history-KV caching" buys:
    only the new suffix of each prompt is computed), (ii) training steps at the far end of the horizon (history of 58..63
    tokens, map of ~100 slots)."""
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    ep = SyntheticEpisodes(cfg, a.batch, seed=seed, instr_len=a.instr_len, device=device, max_frontier=35)
    model.eval()
    model.enable_kv_cache(a.batch, capacity=1024)
    out = {}
    with torch.no_grad():
        for rep in range(2):
            ep.reset(); model.reset_kv_cache()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(T):
                nav_step(wrapped, crit, ep, train=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    out["inference_kv_reuse"] = {"nav_steps_per_s_per_gpu": round(a.batch * T / dt, 2), "steps": T, "S_last": int(ep.S_hist[-1]),
                                 "new_tokens_last_step": model.kv.last_stats["new"]}
    out["inference_kv_reuse"]["two_batches_in_flight"] = two_batches_in_flight(a, model, wrapped, ep, T, max_frontier=35)
    model.kv = None
    model.train()
    # training at the far end: the episode state above is at t = 64; run 6 more training steps there
    model.flop_log = []
    for i in range(2):
        nav_step(wrapped, crit, ep, train=True, last=False)
    model.flop_log = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 6
    for i in range(n):
        nav_step(wrapped, crit, ep, train=True, last=(i == n - 1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fl = algorithmic_flops(cfg, model.flop_log)
    model.flop_log = None
    model.zero_grad()
    out["training_at_t64"] = {"nav_steps_per_s_per_gpu": round(a.batch * n / dt, 2), "ms_per_step": round(dt / n * 1e3, 1),
                              "S": int(ep.S_hist[-1]), "algorithmic_tflops": round(fl / dt / 1e12, 1),
                              "what": "6 per-step-recompute training steps at the far end of the horizon (the reference's formulation)"}
    # round 4: the WHOLE 64-step episode as one prefix-reuse training episode -- the steps' deferred backward is flushed in segments
    # when their rows no longer fit the episode buffers (navillm_amd/episode.py::flush_segment), prompts that reach the 1024-token
    # truncation limit fall back to the reference's formulation for that step
    try:
        from navillm_amd.synthetic import prefix_reuse_episode
        model.zero_grad()
        for rep in range(3):          # (the episode buffers settle over the first TWO episodes of a new shape: tools/t64_probe.py 94 / 117 / 141)
            ep.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            prefix_reuse_episode(wrapped, crit, ep, T, teacher_forced=TF_BATCH)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            model.zero_grad()
        stt = model.episode.stats
        out["training_episode_T64_prefix_reuse"] = {
            "nav_steps_per_s_per_gpu": round(a.batch * T / dt, 2), "ms_per_step": round(dt / T * 1e3, 1), "steps": T,
            "teacher_forced_forward_batched": bool(TF_BATCH), "segments_flushed": int(stt.get("segments_flushed", 0)), "steps_recomputed_after_left_truncation": int(stt.get("recomputed_steps", 0)),
            "prefix_rows": int(stt["prefix_rows"]), "suffix_rows_first_last": [int(stt["suffix_rows"][0]), int(stt["suffix_rows"][-1])] if stt["suffix_rows"] else None,
            "S_last": int(ep.S_hist[-1])}
    except Exception as e:
        out["training_episode_T64_prefix_reuse"] = {"error": f"{type(e).__name__}: {e}"}
        model.episode_abort()
        model.zero_grad()
    model.episode_release()                      # the long episode's buffers (up to 30 % of the card): the next extra loads a 13B model
    return out


def reference_launch_extra(a, cfg, model, opt, crit, device, seed):
    """The reference's own launch line (scripts/multi_wo_pretrain.sh:16: `--batch_size 1 --gradient_accumulation_step 8`, one rank of
    its 8): B = 1 episode per meta-step, the loss scaled by 1 / B / 8, `clip + AdamW + zero_grad` every 8th episode (train.py:68,86-89).
    Every GEMM of a step then has M ~ 650 (recompute) / ~100 (suffix steps) / ~1 150 (a teacher-forced episode's batched backward)
    rows instead of 8 x that -- the regime the cut-off tiles exist for.  Measured for the four training forms, each over FOUR consecutive whole
    accumulation windows (8 episodes x 6 steps = 48 nav steps + 1 optimizer step each) after a warm window, with the GEMM roofline of
    the timed region.  NOT `value`: BASELINE config 2 is B = 8.  The same 8 episodes as ONE batch (`--batch_size 8 --gradient_accumulation_step
    1`: same loss scale 1/8, same gradient -- tests/test_parity_r5_gpu.py) is the headline line above."""
    from navillm_amd import ops
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    ACC = 8
    ep1 = SyntheticEpisodes(cfg, 1, seed=seed, instr_len=a.instr_len, device=device)
    out = {"batch_per_gpu": 1, "gradient_accumulation_step": ACC, "steps_per_window": ACC * STEPS_PER_EPISODE,
           "windows_timed": 4, "best_of": 2, "what": "scripts/multi_wo_pretrain.sh:16 on one rank: 8 episodes of B = 1, 6 nav steps each, loss / 1 / 8, one clip + AdamW per window"}

    def window(form):
        for e in range(ACC):
            ep1.reset()
            if form != "recompute":
                model.begin_episode(ep1.prefix_ids(), teacher_forced=form.startswith("prefix_reuse_teacher_forced"),
                                    accumulate=ACC if form.endswith("_window") else 1)
            for t in range(STEPS_PER_EPISODE):
                nav_step(model, crit, ep1, train=True, last=(t == STEPS_PER_EPISODE - 1), accum=ACC)
            if form != "recompute":
                model.finish_episode()
        opt.clip_grad_norm_(40.0)
        opt.step()
        opt.zero_grad()

    # `..._window` (round 5): begin_episode(..., accumulate=8) -- the eight teacher-forced episodes of the window run as ONE batch at the
    # eighth finish_episode(): the rows of a B = 8 episode, on the reference's own launch line
    for form in ("recompute", "prefix_reuse", "prefix_reuse_teacher_forced", "prefix_reuse_teacher_forced_window"):
        try:
            model.episode_abort()
            model.zero_grad()
            window(form)                                   # warm: buffers sized, kernels' attributes set
            best = None
            NW = 4                                         # consecutive windows per timed region: the host prepares window k + 1 while
            for rep in range(2):                           # the GPU runs window k, as in a training run (the windowed form does ALL its GPU
                tm = GemmTimer()                           # work at the window's end).  Two regions, the faster one reported (the first one
                tm.install(ops)                            # after a form change now and then still meets the allocator: 59 vs 76 seen)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(NW):
                    window(form)
                torch.cuda.synchronize()
                dt_ = time.perf_counter() - t0
                tm.uninstall()
                if best is None or dt_ < best[0]:
                    best = (dt_, tm)
            dt, tm = best
            g = tm.summary(layouts=(0, 1, 2))
            n = NW * ACC * STEPS_PER_EPISODE
            out[form] = {"nav_steps_per_s_per_gpu": round(n / dt, 2), "ms_per_step": round(dt / n * 1e3, 2),
                         "roofline": None if g is None else {
                             "bound": "mfma", "achieved": round(g["tflops"], 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(g["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4), "launches": g["launches"],
                             "avg_launch_ms": round(g["avg_launch_ms"], 4), "gemm_share_of_window": round(g["gemm_seconds"] / dt, 3)}}
        except Exception as e:
            out[form] = {"error": f"{type(e).__name__}: {e}"}
    model.episode_abort()
    model.zero_grad()
    model.episode_release()
    model.episode = model._window = model._episode_plain = None      # the B = 1 caches: the next begin_episode() builds its own
    return out


def unmodified_rollout_extra(a, cfg, model, opt, crit, device, seed):
    """What an UNMODIFIED tasks/agents/mp3d_agent.py + train.py get with INTEGRATION.md's three edits and nothing else (round 6; VERDICT
    r5 next-2): `navillm_amd.synthetic.reference_train_steps` restates train.py:60-91 around MP3DAgent.rollout (:660-778) call for call
    -- `model('panorama')`, `model('navigation')`, `torch.softmax(nav_logits / T, 1)`, criterion, `backward()` per step, `loss.item()`
    per meta-step, clip + step + zero_grad every `gradient_accumulation_step` meta-steps, teacher forcing alternating with DAgger
    sampling (the multi-task stage, mp3d_agent.py:509-525) -- and never calls begin_episode / finish_episode.  The model opens the
    prefix-reuse episode itself and the optimizer's clip / the next rollout's first navigation call closes it (NavModel._auto_*); `fuse_logits`
    is a LazyLogits handle: a teacher-forced rollout never reads it and its LM forward runs as ONE batch when `loss.item()` asks for the
    value, a sampled rollout reads it at every step and runs step by step (`automatic`; `automatic_step_form`: NAVILLM_AUTO_EPISODE=step,
    plain tensors, every forward at once).  Beside it, same process and loop: the same rollouts inside EXPLICIT begin_episode(teacher_forced=False) /
    finish_episode calls, and with NAVILLM_AUTO_EPISODE=0 (the reference's formulation: the whole prompt at every step)."""
    from navillm_amd import ops
    from navillm_amd.synthetic import SyntheticEpisodes, reference_train_steps
    out = {"what": "train.py:60-91 + mp3d_agent.py:660-778 verbatim over the synthetic driver, no begin_episode / finish_episode in the loop; "
                   "feedback alternates teacher / sample per meta-step; clip = optimizer.clip_grad_norm_(40.) (INTEGRATION.md edit 3)"}
    for tag, B, accum, metas in (("B8", a.batch, 1, 8), ("B1x8", 1, 8, 16)):
        epx = SyntheticEpisodes(cfg, B, seed=seed, instr_len=a.instr_len, device=device)
        res = {"batch_per_gpu": B, "gradient_accumulation_step": accum, "meta_steps_timed": metas, "nav_steps_timed": metas * STEPS_PER_EPISODE}
        # (name, automatic episodes?, their form, training stage: "multi" alternates teacher forcing and sampling per meta-step, "pretrain" is
        # all teacher-forced -- mp3d_agent.py:509-525)
        forms = (("automatic", True, "lazy", "multi"), ("automatic_pretrain_stage", True, "lazy", "pretrain"),
                 ("automatic_step_form", True, "step", "multi"), ("explicit_per_step_forward", False, None, "multi"),
                 ("auto_off_recompute", False, None, "multi"))
        for form, auto, aform, stage in forms:
            try:
                model.episode_abort()
                model.zero_grad()
                model.auto_episode, model.auto_form = auto, (aform or "lazy")
                model.auto_stats = {"opened": 0, "closed_by": {}}
                begin = (lambda step: model.begin_episode(epx.prefix_ids(), teacher_forced=False)) if form.startswith("explicit") else None
                end = (lambda step: model.finish_episode()) if form.startswith("explicit") else None

                def loop(n, stage=stage):
                    return reference_train_steps(model, opt, crit, epx, n, STEPS_PER_EPISODE, accum=accum, stage=stage, fused_clip=True,
                                                 on_rollout=begin, after_rollout=end)
                loop(2 * accum if accum > 1 else 2)            # warm: buffers sized for this shape
                tm = GemmTimer()
                tm.install(ops)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loop(metas)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                tm.uninstall()
                g = tm.summary(layouts=(0, 1, 2))
                n = metas * STEPS_PER_EPISODE * B
                res[form] = {"nav_steps_per_s_per_gpu": round(n / dt, 2), "ms_per_step": round(dt / (metas * STEPS_PER_EPISODE) * 1e3, 2),
                             "gemm_frac_of_mfma_peak": None if g is None else round(g["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4)}
                if auto:
                    res[form]["episodes_opened"] = int(model.auto_stats["opened"])
                    res[form]["closed_by"] = dict(model.auto_stats["closed_by"])
            except Exception as e:
                res[form] = {"error": f"{type(e).__name__}: {e}"}
                model.episode_abort()
        try:
            res["automatic_over_explicit"] = round(res["automatic"]["nav_steps_per_s_per_gpu"] / res["explicit_per_step_forward"]["nav_steps_per_s_per_gpu"], 3)
        except Exception:
            pass
        out[tag] = res
    model.auto_episode = False
    model.episode_abort()
    model.zero_grad()
    model.episode_release()
    model.episode = model._window = model._episode_plain = None
    return out


def fp8_13b_extra(a, device, seed):
    """BASELINE config 5 on this rank: Vicuna-13B + scene encoder, inference.  The same episodes first on the bf16 model, then after
    `to_fp8_weight_only()` (decoder Linear weights as e4m3fn codes + per-output-channel scales, navillm_amd/fp8.py): forward-only
    nav steps, nav steps with prefix-K/V reuse, and greedy decoding (the decode steps stream the fp8 codes)."""
    import types
    from navillm_amd import config as C
    from navillm_amd.nav_model import NavModel
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step, qa_step
    from navillm_amd import ops
    cfg13 = C.vicuna_13b(image_feat_size=a.feat)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated(device)
    m = NavModel(nav_config=cfg13, device=device, seed=0)
    m.eval()
    m.lang_model.tokenizer = types.SimpleNamespace(eos_token_id=2, unk_token_id=0)
    crit = CrossEntropyLoss()
    out = {"model": "vicuna-13b (d=5120, L=40, H=40, ff=13824)"}

    def measure(tag):
        r = {}
        for B in (4, 8):
            ep = SyntheticEpisodes(cfg13, B, seed=seed, instr_len=a.instr_len, device=device)
            with torch.no_grad():
                for rep in range(2):
                    ep.reset()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(STEPS_PER_EPISODE):
                        nav_step(m, crit, ep, train=False)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                r[f"forward_only_B{B}"] = round(B * STEPS_PER_EPISODE / dt, 2)
                m.enable_kv_cache(B, capacity=1024)
                for rep in range(2):
                    ep.reset(); m.reset_kv_cache()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(STEPS_PER_EPISODE):
                        nav_step(m, crit, ep, train=False)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                r[f"kv_reuse_B{B}"] = round(B * STEPS_PER_EPISODE / dt, 2)
                m.kv = None
        ep = SyntheticEpisodes(cfg13, 8, seed=seed, instr_len=32, device=device)
        with torch.no_grad():
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g = qa_step(m, ep, train=False, max_new_tokens=24, do_sample=False)[1]
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
        r["greedy_decode_tokens_per_s_B8"] = round(sum(len(x) for x in g["generated_ids"]) / dt, 1)
        out[tag] = r

    measure("bf16_nav_steps_per_s")
    lm_bf16 = m.store.param["lm"].numel() * 2
    # first with the de-quantised operands kept resident (38 GB of 288: prefill GEMMs skip the pre-pass, decode streams the codes),
    # then the memory-lean form (codes only + one shared bf16 scratch panel)
    f8 = m.to_fp8_weight_only(resident_bf16=True)
    measure("fp8_codes_plus_resident_bf16_nav_steps_per_s")
    m.fp8_release_resident()
    torch.cuda.synchronize()
    out["decoder_weight_bytes"] = {"bf16": int(sum(2 * q.numel() for c in f8.codes for q in c.values())), "fp8_codes_plus_scales": int(f8.bytes)}
    out["resident_bytes_after_quantisation"] = int(torch.cuda.memory_allocated(device) - base)
    out["lm_buffer_bytes_bf16_before"] = int(lm_bf16)
    measure("fp8_weight_only_nav_steps_per_s")
    # (default nv_gemm_fp8w mode 7: operands bf16(s * q), bit-identical to the pre-pass.)  The same lean form with the codes converted
    # unscaled and s[n] applied to the fp32 accumulator (opt-in mode 9: one VALU op per pair of weights; one bf16 rounding per weight off
    # the de-quantised-weights semantics, within one output spacing per GEMM)
    try:
        f8.gemm_mode = 9
        ops._L().nv_gemm_fp8w_default_mode(9)
        measure("fp8_weight_only_accumulator_scale_nav_steps_per_s")
    finally:
        f8.gemm_mode = 7
        ops._L().nv_gemm_fp8w_default_mode(7)
    out["what"] = ("inference nav steps/s per GPU over one 6-step episode (panorama + navigation forward, argmax actions) at B=4 and 8, and "
                   "greedy decoding of 24 tokens at B=8 (prefill included; device-side loop replayed from a hipGraph); decode steps stream "
                   "the fp8 codes (gemv_stream.hip); prefill / K/V-reuse GEMMs read either the de-quantised operands kept resident "
                   "(fp8_codes_plus_resident_bf16: 38 GB of weights) or one shared bf16 scratch panel filled per GEMM (fp8_weight_only: "
                   "12.7 GB; round 4: the few-hundred-row GEMMs of K/V-reuse steps multiply with the codes themselves -- weight tile DMA'd as bytes, "
                   "converted to bf16(s*q) on the MFMA fragment path, bit-identical to the pre-pass (nv_gemm_fp8w mode 7; shapes the kernel declines keep "
                   "the 3 B/weight pre-pass); fp8_weight_only_accumulator_scale: opt-in mode 9, the codes converted unscaled and s[n] applied to "
                   "the fp32 accumulator (to_fp8_weight_only(gemm_mode=9)))")
    del m
    torch.cuda.empty_cache()
    return out


def usable_cpus():
    """CPUs this process may actually use: min(cpu_count, affinity mask, cgroup quota).  The MI355X boxes show 256 hardware
    threads but a cgroup quota of 16 CPUs (cpu.max = 1600000/100000): 256 torch threads then spend their time throttled --
    the same two oracle layers took 36 s with 256 threads and 0.7 s with 32 (tools/cpu_probe.py, profiles/r02_cpu_probe.txt)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(a, cfg, seed):
    """The oracle (CPU restatement, kind='port') timed on this box's host cores on a bounded sample:
    ONE episode, ONE nav step, forward + backward, with 8 of the 32 decoder layers (the LM is 99.9% of
    the FLOPs), scaled linearly in layers to the full model."""
    import importlib.util
    from navillm_amd.config import NavConfig
    from navillm_amd.params import synth_state_dict
    from navillm_amd.synthetic import SyntheticEpisodes
    spec = importlib.util.spec_from_file_location("navillm_oracle", os.path.join(ROOT, "oracle", "navillm_oracle.py"))
    O = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(O)
    Ls = min(16, cfg.num_layers)
    c = NavConfig(**{**cfg.__dict__, "num_layers": Ls})
    cores = usable_cpus()
    torch.set_num_threads(cores)
    P = {k: v.requires_grad_(True) for k, v in synth_state_dict(c, seed).items()}
    ep = SyntheticEpisodes(c, 1, seed=seed, instr_len=a.instr_len, device=torch.device("cpu"), fast_maps=False)   # per-node maps: host tensors
    pin = ep.panorama_inputs()
    t0 = time.time()
    pano = O.scene_encoder(P, c, pin["view_img_fts"], pin["view_lens"], pin["loc_fts"], pin["nav_types"])
    pe, pm = pano["pano_embeds"], pano["pano_masks"]
    # map bookkeeping without the HIP mean-pool (host tensors here)
    avg = (pe.detach() * pm.unsqueeze(2)).sum(1) / pm.sum(1, keepdim=True)
    for b, gmap in enumerate(ep.gmaps):
        gmap.node_step_ids[ep.cur[b]] = 1
        gmap.update_node_embed(ep.cur[b], avg[b], rewrite=True)
        for j, vp in enumerate(pin["cand_vpids"][b]):
            if not gmap.graph.visited(vp):
                gmap.update_node_embed(vp, pe[b, j].detach())
    nav = ep.nav_inputs(pe, pm, pin["cand_vpids"])
    ids, am = ep.tokenise(nav, "<cls_1>")
    out = O.navigation(P, c, nav, ids, am)
    loss = O.action_loss(out["fuse_logits"], ep.teacher_targets(nav, False))
    loss.backward()
    dt = time.time() - t0
    scaled = dt * cfg.num_layers / Ls
    return {"value": 1.0 / scaled, "unit": "nav-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/navillm_oracle.py, 1 episode x 1 nav step, forward+backward, S={ids.shape[1]}, "
                      f"{Ls} of {cfg.num_layers} decoder layers timed ({dt:.1f} s) and scaled linearly in layers; "
                      f"torch CPU bf16, {cores} threads = the CPUs this container may use (cpu_count {os.cpu_count()}, cgroup quota applied)"}


def relaunch_one_rank_per_gpu(a):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one process per GPU, exactly the
    launch line of the module docstring) and hand their output through.  Never falls back to a single GPU."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < a.gpus and not REHEARSAL:
        raise SystemExit(f"bench.py: --gpus {a.gpus} requested but only {have} GPU(s) are visible; refusing to report a "
                         f"{a.gpus}-GPU line from fewer devices")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's peer mappings need it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


_T0 = time.perf_counter()
_JSON_OUT = sys.stdout
# NAVILLM_BENCH_REHEARSAL=1: run the N-rank control flow (spawn, barriers, max-over-ranks clock, the data-parallel wrapper's
# hooks inside the real backward, every extra) on a box with FEWER than N GPUs: all ranks share GPU 0, torch.distributed
# runs on gloo and the gradient exchange is staged through the host.  The line is marked "rehearsal": its numbers mean nothing.
REHEARSAL = os.environ.get("NAVILLM_BENCH_REHEARSAL") == "1"


def phase(msg):
    """progress on stderr (the JSON line on stdout stays alone)"""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


TF_BATCH = True                # prefix_reuse mode: batch the teacher-forced episode's LM forward into finish_episode() (--no-tf-batch)


def main():
    global TF_BATCH
    a = parse()
    TF_BATCH = not a.no_tf_batch
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_one_rank_per_gpu(a))
    from navillm_amd.parallel import init_distributed_device, NavDataParallel
    import torch.distributed as dist
    # stdout carries exactly ONE line, the JSON: libraries that print banners on fd 1 (RCCL: "ROCm version / Hostname / Librccl
    # path" per rank, flushed at exit, i.e. AFTER the line; gloo: "[Gloo] Rank ...") are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    global _JSON_OUT
    _JSON_OUT = os.fdopen(json_fd, "w")
    if REHEARSAL:
        os.environ["NAVILLM_COMM"] = "torch"
    # N > 1: the CONTROL plane (rendezvous, barriers, the max-over-ranks clock, the 128-byte RCCL id) runs over gloo, so that nothing of it
    # depends on RCCL; the gradients travel through the C-ABI communicator (RCCL over xGMI), picked and verified by `dp_preflight` below
    control_note = None
    try:
        device, rank, world = init_distributed_device(backend="gloo", device_index=0 if REHEARSAL else None)
    except Exception as e:
        if REHEARSAL:
            raise
        # (a host on which gloo cannot come up: the control plane falls back to ProcessGroupNCCL, as in rounds 1-5)
        control_note = f"gloo control plane failed ({type(e).__name__}: {e}); using nccl"
        print(f"[bench] {control_note}", file=sys.stderr, flush=True)
        device, rank, world = init_distributed_device(backend=None)
    # Host threads for torch's CPU-side glue ops (masks, index lists): a handful.  With the default (all 256 hardware
    # threads) a barrier of the intra-op pool now and then takes 80-100 ms on a tiny tensor (tools/pack_probe.py), longer
    # than a whole forward; and with one process per GPU the ranks must share the host anyway.  (cpu_baseline sets its own.)
    torch.set_num_threads(max(1, min(16, usable_cpus() // max(world, 1))))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the line would not describe the job that ran")
    from navillm_amd import ops
    from navillm_amd.nav_model import NavModel
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.optim import FlatAdamW
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step

    # ---- N > 1 preflight (round 6): a few seconds of verified collectives BEFORE the 7B model is built.  Picks the transport (the C-ABI
    # communicator over RCCL; ProcessGroupNCCL if that fails on any rank) or ends the run with ONE JSON line that names the failing stage
    # per rank -- RCCL with more than one rank has never run on this build's hardware pool, so the first SCALE run must yield a curve or
    # an actionable error, not a hang
    preflight, pre_comm, pre_group = None, None, None
    if world > 1 and not REHEARSAL and os.environ.get("NAVILLM_DP_PREFLIGHT", "1") != "0":
        from navillm_amd.parallel import dp_preflight

        def error_line(reports, why):
            return {"metric": "nav-steps/sec (whole node), Vicuna-7B + 36-view scene enc", "value": None, "unit": "nav-steps/s", "n_gpus": world,
                    "steps": a.steps, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "bf16", "data": "synthetic", "config": {"workload": f"{a.model}, batch={a.batch}/GPU", "parallelism": f"dp{world}"},
                    "error": why, "dp_preflight": reports}

        def on_hang(rep):
            sys.stderr.write(f"[bench rank {rank}] dp_preflight HUNG in stage {rep.get('hang_in_stage')}: {json.dumps(rep)}\n")
            sys.stderr.flush()
            if rank == 0:
                _JSON_OUT.write(json.dumps(error_line([rep], f"dp_preflight hung in stage {rep.get('hang_in_stage')} on rank 0 (watchdog)")) + "\n")
                _JSON_OUT.flush()
        preflight, pre_comm, pre_group = dp_preflight(device, rank, world, watchdog_s=float(os.environ.get("NAVILLM_DP_PREFLIGHT_TIMEOUT", "240")),
                                                      on_hang=on_hang)
        all_reports = [None] * world
        dist.all_gather_object(all_reports, preflight)
        phase(f"dp preflight: transport = {preflight['transport']}")
        if preflight["transport"] is None:
            if rank == 0:
                _JSON_OUT.write(json.dumps(error_line(all_reports, "no working collective transport: see dp_preflight[rank].stages")) + "\n")
                _JSON_OUT.flush()
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(2)
        if preflight["transport"] == "torch":
            os.environ["NAVILLM_COMM"] = "torch"
        preflight = all_reports
    cfg = make_cfg(a)
    seed = 1234 + rank
    torch.manual_seed(seed)
    model = NavModel(nav_config=cfg, device=device, seed=0)   # same weights on every rank
    model.train()
    # every number of this file names ITS training form (explicit begin_episode / finish_episode, or the reference's full-prompt recompute);
    # the automatic episodes the model would otherwise open for training-mode navigation calls are measured in `unmodified_rollout`
    model.auto_episode = False
    model.reserve_activations(a.batch, a.instr_len + 256)
    opt = FlatAdamW(model, lr=a.lr)
    if world == 1 and os.environ.get("NAVILLM_DP_FORCE"):
        # single-GPU rehearsal of the N > 1 path: a one-rank RCCL group, every gradient slice all-reduced from inside the
        # backward exactly as with 8 ranks (the numbers are then NOT comparable with the plain N = 1 line)
        if os.environ.get("NAVILLM_COMM") != "rccl":
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29617", world_size=1, rank=0, device_id=device)
        wrapped = NavDataParallel(model, force_sync=True)
    else:
        wrapped = NavDataParallel(model, comm=pre_comm, group=pre_group) if world > 1 else model
        if world > 1 and os.environ.get("NAVILLM_DP_ALGO") is None:
            wrapped.calibrate()        # all-reduce vs reduce-scatter + all-gather on one layer slice: keep the faster
    crit = CrossEntropyLoss()
    ep = SyntheticEpisodes(cfg, a.batch, seed=seed, instr_len=a.instr_len, device=device)
    import contextlib

    def make_step(mode):
        prefix = mode == "prefix_reuse"
        explicit = prefix and a.explicit_episodes
        # round 6: in prefix_reuse mode the loop below contains NOTHING but the reference's calls (nav_step = panorama, navigation,
        # criterion, backward; then clip + step + zero_grad): the model opens the prefix-reuse episode on the first training-mode
        # navigation call and `optimizer.clip_grad_norm_` hands its gradients over (NavModel._auto_*; teacher-forced rollout: nothing
        # reads the LazyLogits handles, so the LM forward of all steps runs as one batch there).  --explicit-episodes: the
        # begin_episode / finish_episode calls of rounds 3-5 instead (bit-identical results, tests/test_auto_episode_gpu.py)
        model.auto_episode = prefix and not explicit
        model.auto_form = "lazy" if TF_BATCH else "step"
        dp_ctx = wrapped.final_backward if hasattr(wrapped, "final_backward") else contextlib.nullcontext

        def one_step(i, end=False):
            """iteration i of the rollout loop (tasks/agents/mp3d_agent.py:660) for the rank's B episodes; every 6th one (or `end`)
            ends the episodes: (prefix mode: the episode's deferred backward,) clip(40) + AdamW + zero_grad (train.py:86-89)"""
            pos = i % STEPS_PER_EPISODE
            last = pos == STEPS_PER_EPISODE - 1 or end
            if explicit and pos == 0:
                # static prompt prefix: forward once, K/V cached per layer; the rollout is teacher-forced (imitation learning), so the
                # steps' LM forward is batched into finish_episode() as well (round 4; --no-tf-batch: per-step forward as in round 3)
                model.begin_episode(ep.prefix_ids(), teacher_forced=TF_BATCH)
            loss, logits = nav_step(wrapped, crit, ep, train=True, last=last, final=last and not prefix)
            if last:
                if explicit:
                    with dp_ctx():                            # the LAST backward before the optimizer step: DP exchanges from inside it
                        model.finish_episode()
                    opt.clip_grad_norm_(40.0)
                elif prefix:
                    # the automatic episode's deferred backward runs inside the clip (it needs .grad first); under data parallelism
                    # that is the backward the exchange overlaps with (N = 1: a null context)
                    with dp_ctx():
                        opt.clip_grad_norm_(40.0)
                else:
                    opt.clip_grad_norm_(40.0)
                opt.step()
                opt.zero_grad()
                ep.reset()
            return loss
        return one_step

    def run_mode(mode, steps, warmup, prewarm, sync_ranks):
        """untimed setup steps, then `warmup` untimed steps, then `steps` timed ones bracketed by barrier + synchronize on both sides
        -> (seconds (max over ranks), GemmTimer, last loss).
        Episodes are 6 steps long; they open with per-episode work (prefix mode: the prefix forward) and END with most of it (prefix
        mode: the backward of EVERY token row of the episode, deferred to finish_episode(); both modes: clip + AdamW).  Every timed
        step's work must lie inside the timed region, so (1) the region always STARTS at an episode boundary -- the setup phase runs
        `prewarm` steps plus as many more (< 6) as it takes for setup + warmup to be whole episodes -- and (2) the LAST timed step
        always ENDS its episode: with K = 20 the window holds episodes of 6, 6, 6 and 2 steps, each complete (begin, steps, deferred
        backward, optimizer step).  The short last episode spreads its prefix and its optimizer step over 2 steps instead of 6, so the
        K = 20 figure is BELOW the steady-state rate of 6-step episodes (`whole_episodes` in the JSON line: 18 steps = 3 whole
        episodes, measured right after); a window that merely stopped after 20 steps would leave the backward of the last two steps
        outside the timed region."""
        one_step = make_step(mode)
        model.episode_abort()
        model.zero_grad()
        ep.reset()
        setup = prewarm + (-(prewarm + warmup)) % STEPS_PER_EPISODE
        for i in range(setup):          # setup, not part of the protocol's W/K accounting
            one_step(i)
        for i in range(setup, setup + warmup):
            one_step(i)
        base = setup + warmup           # a multiple of 6: step `base` opens an episode
        if getattr(wrapped, "profile", False):
            wrapped.exchange_stats(reset=True)      # the exchange events of the setup / warmup steps (first one: RCCL channel setup) do not count
        tm = GemmTimer()
        if not a.no_profile:
            tm.install(ops)
        torch.cuda.synchronize()
        if world > 1 and sync_ranks:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = None
        for i in range(base, base + steps):
            tm.active = GemmTimer.sampled(i - base, steps)
            loss = one_step(i, end=(i == base + steps - 1))
        torch.cuda.synchronize()
        if world > 1 and sync_ranks:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if not a.no_profile:
            tm.uninstall()
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if (REHEARSAL or not dist.is_initialized() or dist.get_backend() == "gloo") else device)
        if world > 1 and sync_ranks:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()), tm, loss

    phase("model built")
    _run_mode = run_mode

    def run_mode(*args, **kw):           # (every other number of this file names its own form: automatic episodes off again afterwards)
        try:
            return _run_mode(*args, **kw)
        finally:
            model.grad_handover("bench") if getattr(model, "_auto_open", False) else None
            model.auto_episode = False

    if world > 1 and hasattr(wrapped, "exchange_stats"):
        wrapped.profile = True           # HIP events around the exchange of every optimizer step (side stream + the join): `dp.exchange`
    dt, timer, loss = run_mode(a.mode, a.steps, a.warmup, a.prewarm, True)
    seq_len_main = int(max(ep.S_hist)) if ep.S_hist else None
    main_stats = dict(model.episode.stats) if (a.mode == "prefix_reuse" and model.episode is not None) else None

    # ---- N > 1: what the gradient exchange cost inside the timed region, and which RCCL every rank bound
    dp_exchange, rccl_versions = None, None
    if world > 1:
        try:
            dp_exchange = wrapped.exchange_stats()
            wrapped.profile = False
            from navillm_amd import lib as _nvlib
            mine = int(_nvlib.load().nv_comm_rccl_version()) if wrapped.comm is not None else None
            rccl_versions = [None] * world
            dist.all_gather_object(rccl_versions, mine)
        except Exception as e:
            dp_exchange = {"error": f"{type(e).__name__}: {e}"}
    phase(f"timed region done: {dt:.2f} s")
    # ---- the same mode over WHOLE 6-step episodes (steady state; reported aside, never `value`): the K-step window above ends with a
    # short episode whenever K is not a multiple of 6
    whole = None
    if a.steps % STEPS_PER_EPISODE and (not a.no_extras or a.model != "tiny") and not a.no_other_mode:
        try:
            w_steps = 3 * STEPS_PER_EPISODE
            w_dt, _, _ = run_mode(a.mode, w_steps, 0, 0, world > 1)
            whole = {"steps": w_steps, "episodes": 3, "ms_per_step": round(w_dt / w_steps * 1e3, 2),
                     "nav_steps_per_s": round(a.batch * world * w_steps / w_dt, 2),
                     "what": "the headline mode over 3 whole 6-step episodes, same process and model, right after the timed region"}
            if a.mode == "prefix_reuse" and TF_BATCH:
                # the round-3 form of the same mode (every step's LM forward runs when the step is called), same window
                global_tf = TF_BATCH
                try:
                    globals()["TF_BATCH"] = False
                    # (one untimed episode first: the per-step form's K/V-cache layout is allocated on first use since round 6)
                    p_dt, _, _ = run_mode(a.mode, w_steps, 0, STEPS_PER_EPISODE, world > 1)
                    whole["per_step_forward_nav_steps_per_s"] = round(a.batch * world * w_steps / p_dt, 2)
                    # what a fine-tune sees: the multi-task stage alternates teacher forcing with DAgger sampling meta-step by
                    # meta-step (mp3d_agent.py:509-525: `step % 2 == 0` -> feedback="teacher", else "sample"); a sampled step needs
                    # its logits at once, so only the teacher-forced half may batch its forward -> harmonic mean of the two rates
                    tf_r, ps_r = whole["nav_steps_per_s"], whole["per_step_forward_nav_steps_per_s"]
                    whole["finetune_blend_nav_steps_per_s"] = round(2.0 / (1.0 / tf_r + 1.0 / ps_r), 2)
                    whole["finetune_blend_what"] = ("harmonic mean of the teacher-forced (batched forward) and per-step-forward rates: the "
                                                    "50/50 teacher / sample schedule of fine-tuning, mp3d_agent.py:509-525; pre-training is all teacher-forced")
                finally:
                    globals()["TF_BATCH"] = global_tf
        except Exception as e:
            whole = {"error": f"{type(e).__name__}: {e}"}
    # ---- the OTHER training mode, same process, same model (reported under `other_mode`, never `value`)
    other = None
    if (not a.no_extras or a.model != "tiny") and not a.no_other_mode:
        other_mode = "recompute" if a.mode == "prefix_reuse" else "prefix_reuse"
        try:
            o_steps = 2 * STEPS_PER_EPISODE
            o_dt, o_timer, o_loss = run_mode(other_mode, o_steps, 0, STEPS_PER_EPISODE, False)
            other = {"mode": other_mode, "nav_steps_per_s_per_gpu": round(a.batch * o_steps / o_dt, 2), "steps": o_steps,
                     "ms_per_step": round(o_dt / o_steps * 1e3, 2), "roofline": roofline_of(o_timer, o_dt, o_steps, other_mode, model)}
            if other_mode == "prefix_reuse" and model.episode is not None:
                other["token_rows_last_episode"] = {"prefix_once": int(model.episode.stats["prefix_rows"]),
                                                    "suffix_per_step": [int(x) for x in model.episode.stats["suffix_rows"]]}
        except Exception as e:          # never take the headline line (or a rank) down
            other = {"mode": other_mode, "error": f"{type(e).__name__}: {e}"}
        model.episode_abort()
        model.zero_grad()
        ep.reset()
    # ---- untimed extra: the same nav step without loss/backward (validation rollout, mp3d_agent.py:530-590)
    infer = infer_kv = None
    if a.infer_steps > 0:
        try:        # extras must never take the headline number down with them (nor leave a rank behind at the final barrier)
            infer, infer_kv = inference_extras(a, model, wrapped, crit, ep)
        except Exception as e:
            infer = {"error": f"{type(e).__name__}: {e}"}
            infer_kv = None
        model.kv = None
        model.train()
    cpu_line = None
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        phase("cpu baseline")
        try:
            cpu_line = cpu_baseline(a, cfg, 1234)
        except Exception as e:  # the baseline must never take the GPU number down with it
            cpu_line = {"value": None, "unit": "nav-steps/s", "cores": os.cpu_count(), "kind": "port",
                        "sample": f"failed: {type(e).__name__}: {e}"}
        torch.set_num_threads(max(1, min(16, usable_cpus() // max(world, 1))))
    extras = {}
    if not a.no_extras and (a.model != "tiny" or a.extras_on_tiny):
        for name, fn in (("mixed_task_training_config3", lambda: mixed_task_extra(a, cfg, model, wrapped, opt, crit, device, seed + 100)),
                         ("long_horizon_config4", lambda: long_horizon_extra(a, cfg, model, wrapped, crit, device, seed + 200)),
                         ("reference_launch_line", lambda: reference_launch_extra(a, cfg, model, opt, crit, device, seed + 400) if world == 1 else None),
                         ("unmodified_rollout", lambda: unmodified_rollout_extra(a, cfg, model, opt, crit, device, seed + 500) if world == 1 else None),
                         ("fp8_weight_only_13b_config5", lambda: fp8_13b_extra(a, device, seed + 300) if world == 1 else None)):
            phase(name)
            try:        # never take the headline line (or a rank) down
                r = fn()
                if r is not None:
                    extras[name] = r
            except Exception as e:
                extras[name] = {"error": f"{type(e).__name__}: {e}"}
            model.kv = None
            model.flop_log = None
            model.train()
            model.zero_grad()

    if rank == 0:
        value = a.batch * a.steps * world / dt
        line = {
            "metric": "nav-steps/sec (whole node), Vicuna-7B + 36-view scene enc", "value": round(value, 3),
            "unit": "nav-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{a.model} + 36-view x {a.feat}-d scene encoder, R2R-shaped synthetic episodes, "
                                   f"batch={a.batch}/GPU, {a.instr_len}-token instructions, training step (fwd+bwd, "
                                   f"clip+AdamW every {STEPS_PER_EPISODE} steps)",
                       "global_batch": a.batch * world, "seq_len": seq_len_main,
                       "parallelism": f"dp{world}", "loss": float(loss.detach()) if loss is not None else None,
                       "timed_window": "starts at an episode boundary (setup + warmup = whole 6-step episodes) and the last timed step ends "
                                       "its episode: %d complete episodes (%s steps) inside the %d timed steps, each with its begin, its %s"
                                       "clip + AdamW -- no timed step's work is left outside the window%s" % (
                                           (a.steps + STEPS_PER_EPISODE - 1) // STEPS_PER_EPISODE,
                                           " + ".join(["6"] * (a.steps // STEPS_PER_EPISODE) + ([str(a.steps % STEPS_PER_EPISODE)] if a.steps % STEPS_PER_EPISODE else [])),
                                           a.steps, "deferred backward (finish_episode) and " if a.mode == "prefix_reuse" else "",
                                           "; the short last episode makes this figure conservative, see whole_episodes" if a.steps % STEPS_PER_EPISODE else ""),
                       "training_mode": a.mode,
                       "teacher_forced_forward_batched": bool(TF_BATCH and a.mode == "prefix_reuse"),
                       "episodes": ("explicit begin_episode / finish_episode calls in the loop (--explicit-episodes)" if a.explicit_episodes else
                                    "AUTOMATIC: the timed loop holds only the reference's calls -- model('panorama'), model('navigation'), criterion, "
                                    "backward(), optimizer.clip_grad_norm_ / step / zero_grad; the model opens the prefix-reuse episode itself and the "
                                    "clip hands its gradients over (INTEGRATION.md section 2's three edits, nothing else)") if a.mode == "prefix_reuse" else None,
                       "training_mode_what": MODE_WHAT[a.mode] + (TF_WHAT if (TF_BATCH and a.mode == "prefix_reuse") else "") +
                                             (AUTO_WHAT if (a.mode == "prefix_reuse" and not a.explicit_episodes) else "")},
        }
        if main_stats is not None:
            line["config"]["token_rows_last_episode"] = {"prefix_once": int(main_stats["prefix_rows"]),
                                                         "suffix_per_step": [int(x) for x in main_stats["suffix_rows"]]}
        if whole is not None:
            line["whole_episodes"] = whole
        if other is not None:
            line["other_mode"] = other
        if REHEARSAL:
            line["rehearsal"] = "all ranks shared GPU 0 over gloo: control-flow check only, the numbers are meaningless"
        if whole is not None and "finetune_blend_nav_steps_per_s" in whole:
            line["finetune_blend"] = {"nav_steps_per_s": whole["finetune_blend_nav_steps_per_s"], "teacher_forced": whole["nav_steps_per_s"],
                                      "per_step_forward": whole["per_step_forward_nav_steps_per_s"],
                                      "recompute_reference_formulation": (other or {}).get("nav_steps_per_s_per_gpu") if (other or {}).get("mode") == "recompute" else None,
                                      "what": whole["finetune_blend_what"]}
        if world > 1:
            line["dp"] = {"transport": "nv_comm (RCCL, C ABI)" if wrapped.comm is not None else "torch.distributed",
                          "reduce": wrapped.reduce, "algo": wrapped.algo, "calibration": wrapped.calibration,
                          "rccl_version_per_rank": rccl_versions, "exchange": dp_exchange, "control_plane": dist.get_backend(), "control_plane_note": control_note, "preflight": preflight,
                          "what": "gradients averaged once per optimizer step, per-layer slices exchanged from inside the "
                                  "episode's last backward on a side stream"}
        if infer is not None:
            line["inference_forward_only"] = infer
        if infer_kv is not None:
            line["inference_prefix_kv_reuse"] = infer_kv
        line.update(extras)
        rl = roofline_of(timer, dt, a.steps, a.mode, model)
        if rl is not None:
            line["roofline"] = rl
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        phase("done")
        _JSON_OUT.write(json.dumps(line) + "\n")
        _JSON_OUT.flush()
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
