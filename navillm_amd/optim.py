"""Optimizer step of the measured path: clip_grad_norm_(params, 40.) + AdamW (train.py:86-89,
tools/optims.py:43-45: AdamW over ALL parameters, default betas/eps/weight_decay, constant LR).

One fused HIP kernel per contiguous run of the flat buffers (K13): the bf16 LM buffer carries bf16 moments, exactly as
torch.optim.AdamW does for bf16 parameters in the reference; the clip coefficient is produced on the device (no host sync)
and folded into the same pass.

`FlatAdamW` IS a `torch.optim.Optimizer` (param_groups / defaults / `initial_lr`), so the reference's
`get_constant_schedule_with_warmup(optimizer, ...)` + `lr_scheduler.step()` (tools/optims.py:47, train.py:91) work on it
unchanged; the learning rate is read from `param_groups[0]["lr"]` at every step.

Which parameters are updated: torch's AdamW skips a parameter whose `.grad` is None and starts that parameter's own step
count (bias correction) with its first gradient.  With the reference's pinned torch 1.10 `zero_grad()` zeroes instead of
dropping gradients, so a parameter is skipped until it receives its FIRST gradient and updated at every step after that:
`og_head` never moves (nav_model.py:78-80 vs :445), `lm_head` only once an LM-loss mode has run, `obj_projector` only once a
batch carried objects.  Here gradients live in flat buffers and are never None, so the backward functions report what they
accumulated into (`FlatStore.touch`), and the update runs over the runs of tensors that have been touched so far, each with
its own step count.

Optimizer state lives in this flat layout; `load_state_dict` also accepts the reference checkpoint's `optimizer` entry
(`torch.optim.AdamW.state_dict()` as written by tools/optims.py:65-78 and read back at :26-29) and `reference_state_dict()`
writes one, so a run can be resumed on either side (the conversion is `reference_optimizer_to_flat` /
`flat_to_reference_optimizer` below; pinned by tests/golden/g13_optimizer_bf16.npz)."""
import os

import torch
from . import ops
from .params import param_specs


def reference_param_orders(cfg):
    """Candidate orders of `[n for n, p in model.named_parameters() if p.requires_grad]` (tools/optims.py:43), which is what
    the integer keys of the reference's optimizer state index.  Module registration order (nav_model.py:33-91,
    image_embedding.py:11-49) = params.param_specs; inside the Llama MLP it depends on the transformers release the checkpoint
    was written under: the pinned 4.28 registers gate, down, up (its published modeling_llama.py), current releases gate, up,
    down (the order tests/golden/make_golden.py::gen_optimizer_state records from the installed one).  The shapes tell the two
    apart ([ff, d] vs [d, ff]), which is how `reference_optimizer_to_flat` picks."""
    new = [n for n, _, _ in param_specs(cfg)]
    old = list(new)
    for i, n in enumerate(new):
        if n.endswith("mlp.up_proj.weight") and new[i + 1].endswith("mlp.down_proj.weight"):
            old[i], old[i + 1] = new[i + 1], new[i]
    return [new, old]


def names_from_model_state_dict(keys, cfg):
    """the parameter order of the model a checkpoint was written from, read off its `model_state_dict` keys (state_dict() and
    named_parameters() walk the module tree in the same order; buffers and the DDP `module.` prefix dropped)"""
    known = {n for n, _, _ in param_specs(cfg)}
    out = [k[7:] if k.startswith("module.") else k for k in keys]
    return [k for k in out if k in known]


def _step_number(v):
    return int(v.item()) if torch.is_tensor(v) else int(v)


def reference_optimizer_to_flat(store, sd, names=None):
    """`torch.optim.AdamW.state_dict()` of the reference -> (exp_avg/exp_avg_sq written into store's flat buffers,
    step_count, born, hyper-parameters).  Parameters without an entry in `sd["state"]` never had a gradient
    (optimizer.step skips `p.grad is None`) and stay untouched here."""
    shapes = store.shape_of
    if len(sd["param_groups"]) != 1:
        raise ValueError(f"reference optimizer state with {len(sd['param_groups'])} param groups: the reference builds ONE (tools/optims.py:43)")
    idx = sd["param_groups"][0]["params"]
    cands = [list(names)] if names is not None else reference_param_orders(store.cfg)
    order = None
    for c in cands:
        if len(c) == len(idx) and all(tuple(sd["state"][k]["exp_avg"].shape) == tuple(shapes[c[j]])
                                      for j, k in enumerate(idx) if k in sd["state"]):
            order = c
            break
    if order is None:
        raise ValueError(f"reference optimizer state: {len(idx)} parameters whose shapes fit none of the known parameter orders "
                         f"({[len(c) for c in cands]} names); pass names=names_from_model_state_dict(ckpt['model_state_dict'], cfg)")
    store.init_optimizer_state()
    for buf in (store.exp_avg, store.exp_avg_sq):        # an optimizer that has already stepped: no stale moments for parameters the
        for t in buf.values():                           # checkpoint has no entry for
            t.zero_()
    steps = {}
    for j, k in enumerate(idx):
        st = sd["state"].get(k)
        if st is None:
            continue
        n = order[j]
        steps[n] = _step_number(st["step"])
        for buf, key in ((store.exp_avg, "exp_avg"), (store.exp_avg_sq, "exp_avg_sq")):
            v = store._view(buf, n)
            v.copy_(st[key].to(device=v.device, dtype=v.dtype))
    step_count = max(steps.values(), default=0)
    born = {n: step_count - s for n, s in steps.items()}
    g = sd["param_groups"][0]
    if g.get("amsgrad", False) or g.get("maximize", False):
        raise ValueError("reference optimizer state with amsgrad/maximize: the reference never sets them (tools/optims.py:43)")
    hyper = {k: g[k] for k in ("lr", "betas", "eps", "weight_decay", "initial_lr") if k in g}
    return step_count, born, hyper


def flat_to_reference_optimizer(store, step_count, born, group, names=None):
    """the inverse: a `torch.optim.AdamW.state_dict()` the reference's `optimizer.load_state_dict` (tools/optims.py:29) accepts"""
    order = list(names) if names is not None else reference_param_orders(store.cfg)[0]
    state = {}
    for j, n in enumerate(order):
        if n not in born:
            continue
        state[j] = {"step": torch.tensor(float(step_count - born[n])),
                    "exp_avg": store._view(store.exp_avg, n).detach().clone().cpu(),
                    "exp_avg_sq": store._view(store.exp_avg_sq, n).detach().clone().cpu()}
    g = {k: v for k, v in group.items() if k != "params"}
    g.setdefault("amsgrad", False)
    g["params"] = list(range(len(order)))
    return {"state": state, "param_groups": [g]}


def active_segments(store, born):
    """-> {group: [(start, end, born_step), ...]}: maximal runs of adjacent touched tensors (alignment padding included)
    that started in the same optimizer step.  Pure host bookkeeping (CPU-testable)."""
    out = {}
    for grp, names in store.names.items():
        segs = []
        for n in names:
            if n not in born:
                continue
            s = store.offsets[n]
            e = s + store.alloc_sizes[n]
            if segs and segs[-1][1] == s and segs[-1][2] == born[n]:
                segs[-1] = (segs[-1][0], e, born[n])
            else:
                segs.append((s, e, born[n]))
        out[grp] = segs
    return out


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=40.0):
        self.model = model
        self.store = model.store
        super().__init__([{"params": list(model.parameters())}], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self.born = {}                   # parameter name -> number of optimizer steps taken before its first gradient
        self._segs, self._segs_key = None, None
        self.store.init_optimizer_state()
        self._clip = torch.ones(2, dtype=torch.float32, device=self.store.device)
        self._clip_valid = False
        # round 4: `optimizer.zero_grad()` (train.py:89) folded into the update kernel -- the gradient is zeroed as it is consumed, the
        # zero_grad() that follows only fills the gaps between the updated segments (tensors that never had a gradient: a few % of
        # the buffer).  NAVILLM_ADAMW_ZERO_GRAD=0: separate full fill as before.
        self.fused_zero_grad = os.environ.get("NAVILLM_ADAMW_ZERO_GRAD", "1") != "0"
        self._zeroed_segs = None
        self._zeroed_at = -1            # FlatStore.grad_writes when step() zeroed them: any gradient write since invalidates the fused form
        # round 6 (OPT-IN: NAVILLM_ADAMW_OVERLAP=1 / `optimizer.overlap_update = True`): the LM group's update runs on a SIDE stream,
        # decoder layer by decoder layer, while the launch stream goes on with the next episode's scene-encoder / fusion steps (fp32 group:
        # updated on the launch stream); whoever touches an LM parameter or gradient next waits for the part it needs
        # (FlatStore.wait_params).  Built to take the 18 ms of weight streaming off the critical path; MEASURED (same box, ABAB,
        # gpurun_out/r6_ab_overlap*.json): whole episodes 150.1 -> 150.9 nav-steps/s (+0.5 %), the driver's window 135.2 -> 134.8 (-0.3 %) --
        # the host runs ahead, so the next episode's ~360 encoder launches are already queued and take ~2 ms, and the rest of the update
        # lands on the batched forward's GEMMs, which slow down by what the update gains (their event-timed rate 0.534 -> 0.519 of the
        # MFMA peak).  Not a gain worth a contract on raw tensor handles, hence off by default.  Contract when on: between step() and the
        # next model call read parameters through model.parameters() / state_dict() (they join the update); a tensor handle kept from
        # BEFORE the step is ordered only after torch.cuda.synchronize().
        self.overlap_update = os.environ.get("NAVILLM_ADAMW_OVERLAP", "0") == "1" and self.store.device.type == "cuda"
        self._side = None

    # `lr` as an attribute mirrors the single param group (tests / callers that poke it directly)
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def _no_open_episode(self):
        """prefix-reuse training (navillm_amd/episode.py) hands over its gradients -- in the default form ALL of the LM's -- at
        `finish_episode()`: an update in front of it would silently train on the encoder's gradients alone"""
        if self.store.tainted:
            raise RuntimeError(f"the gradient buffers are inconsistent ({self.store.tainted}): call zero_grad() before the next "
                               "clip_grad_norm_ / step")
        hand = getattr(self.model, "grad_handover", None)
        if hand is not None:
            hand("optimizer")            # an open accumulation window (begin_episode(..., accumulate=n)) and an AUTOMATIC episode
                                         # (NavModel._auto_*: opened by the unmodified rollout's navigation calls) hand their gradients over now
        ep = getattr(self.model, "episode", None)
        if ep is not None and ep.has_pending_gradients():
            raise RuntimeError("optimizer step inside an open prefix-reuse episode: call model.finish_episode() (under "
                               "`with model.final_backward():` when data-parallel) after the episode's last backward() and before "
                               "clip_grad_norm_ / step, or model.episode_abort() to drop the episode")

    def _dp_flush(self):
        self.store.wait_params()         # a previous step's update still on the side stream: the gradients it zeroes / the clip vector it reads
        self._no_open_episode()
        dp = getattr(self.model, "_dp", None)
        if dp is not None:
            dp.flush()                   # reduce="step": the mean over ranks of the accumulated gradient, once

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm=None):
        """Returns the device tensor [total_norm, coef]; the scaling itself is applied inside step()."""
        self._dp_flush()
        mn = self.max_grad_norm if max_norm is None else max_norm
        ops.clip_coef([self.store.grad["lm"], self.store.grad["f32"]], mn, out2=self._clip)
        self._clip_valid = True
        return self._clip

    @torch.no_grad()
    def step(self, closure=None):
        self._dp_flush()
        st = self.store
        dp = getattr(self.model, "_dp", None)
        if dp is not None:
            dp.merge_touched()           # a tensor any rank has a gradient for is updated (and starts its step count) on every rank
        for n in st.touched:
            if n not in self.born:
                self.born[n] = self.step_count
        self.step_count += 1
        key = len(self.born)
        if self._segs_key != key:
            self._segs, self._segs_key = active_segments(st, self.born), key
        g0 = self.param_groups[0]
        lr, (b1, b2), eps, wd = g0["lr"], g0["betas"], g0["eps"], g0["weight_decay"]
        clip = self._clip if self._clip_valid else None

        def upd(grp, s, e, born):
            ops.adamw_(st.param[grp][s:e], st.grad[grp][s:e], st.exp_avg[grp][s:e], st.exp_avg_sq[grp][s:e],
                       self.step_count - born, lr, b1, b2, eps, wd, clip=clip, zero_grad=self.fused_zero_grad)
        if self.overlap_update and st.grad["lm"].is_cuda:
            # fp32 group (scene encoder, fusion: the very next launches read it) on the launch stream; the LM group on the side stream, cut
            # at the decoder layers' boundaries (the kernel is elementwise: the same values whatever the cut) with an event per layer
            for s, e, born in self._segs.get("f32", ()):
                upd("f32", s, e, born)
            if self._side is None:
                self._side = torch.cuda.Stream(device=st.device)
            main = torch.cuda.current_stream(st.device)
            L = st.cfg.num_layers
            bounds = [st.layer_slice(i) for i in range(L)]
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                def run(lo, hi):
                    for s, e, born in self._segs.get("lm", ()):
                        a, b = max(s, lo), min(e, hi)
                        if a < b:
                            upd("lm", a, b, born)
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                    return ev
                head = run(0, bounds[0][0])
                layer = [run(lo, hi) for lo, hi in bounds]
                done = run(bounds[-1][1], st.total["lm"])
            st.begin_async_update(self._side, head, layer, done, main.cuda_stream)
        else:
            for grp, segs in self._segs.items():
                for s, e, born in segs:
                    upd(grp, s, e, born)
        self._clip_valid = False
        self._zeroed_segs = {grp: [(s, e) for s, e, _ in segs] for grp, segs in self._segs.items()} if self.fused_zero_grad else None
        self._zeroed_at = st.grad_writes

    def zero_grad(self, set_to_none=False):
        chk = getattr(self.model, "assert_no_auto_pending", None)
        if chk is not None:
            chk("optimizer.zero_grad()")
        z, self._zeroed_segs = self._zeroed_segs, None
        # the fused form is only valid when nothing wrote a gradient since step() zeroed the updated segments (with
        # step(); backward(); zero_grad() only the gaps were cleared and the new gradients survived into the next step)
        if z is None or self.store.grad is None or self.store.grad_writes != self._zeroed_at:
            self.store.zero_grad()
            return
        self.store.tainted = None
        # step() already zeroed every updated segment: fill only what lies between them
        for grp, g in self.store.grad.items():
            pos = 0
            for s, e in sorted(z.get(grp, ())):
                if s > pos:
                    g[pos:s].zero_()
                pos = max(pos, e)
            if pos < g.numel():
                g[pos:].zero_()
        self.store.layers_zero = True          # every element is zero now: step() zeroed the updated segments, the gaps were filled here

    def state_dict(self):
        """flat-layout state (copies); not a torch.optim.AdamW state dict"""
        self.store.wait_params()
        return {"step": self.step_count, "born": dict(self.born),
                "exp_avg": {g: t.clone() for g, t in self.store.exp_avg.items()},
                "exp_avg_sq": {g: t.clone() for g, t in self.store.exp_avg_sq.items()},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def reference_state_dict(self, names=None):
        """this optimizer's state as the reference's `optimizer.state_dict()` (per-parameter, indexed in `names` order; default
        the installed transformers' `named_parameters()` order)"""
        self.store.wait_params()
        return flat_to_reference_optimizer(self.store, self.step_count, self.born, self.param_groups[0], names)

    def load_state_dict(self, sd, names=None):
        self.store.wait_params()
        if "state" in sd and "param_groups" in sd and "exp_avg" not in sd:
            # a reference checkpoint's `optimizer` entry (tools/optims.py:26-29 calls exactly this method with it)
            self.step_count, self.born, hyper = reference_optimizer_to_flat(self.store, sd, names)
            self.store.touched.clear()                   # (old flags would re-birth stale tensors at the next step)
            self.store.touched.update(self.born)
            self._segs_key = None
            self.param_groups[0].update(hyper)
            return
        if "exp_avg" not in sd or not isinstance(sd["exp_avg"], dict) or "lm" not in sd["exp_avg"]:
            raise ValueError("FlatAdamW.load_state_dict: neither a FlatAdamW state nor a torch.optim.AdamW state_dict")
        self.step_count = sd["step"]
        self.born = dict(sd.get("born", {}))
        self.store.touched.update(self.born)
        self._segs_key = None
        for g in ("lm", "f32"):
            self.store.exp_avg[g].copy_(sd["exp_avg"][g])
            self.store.exp_avg_sq[g].copy_(sd["exp_avg_sq"][g])
        for g, saved in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in saved.items() if k != "params"})


def constant_schedule_with_warmup(optimizer, num_warmup_steps=0):
    """`transformers.get_constant_schedule_with_warmup` (tools/optims.py:47) without the transformers import"""
    def lr_lambda(step):
        return float(step) / float(max(1.0, num_warmup_steps)) if step < num_warmup_steps else 1.0
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)
