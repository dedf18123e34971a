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
its own step count.  Optimizer state is in this flat layout: not interchangeable with a reference checkpoint's
`optimizer` entry (model weights are: `NavModel.load_reference_state_dict`)."""
import torch
from . import ops


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

    # `lr` as an attribute mirrors the single param group (tests / callers that poke it directly)
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def _dp_flush(self):
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
        for grp, segs in self._segs.items():
            for s, e, born in segs:
                ops.adamw_(st.param[grp][s:e], st.grad[grp][s:e], st.exp_avg[grp][s:e], st.exp_avg_sq[grp][s:e],
                           self.step_count - born, lr, b1, b2, eps, wd, clip=clip)
        self._clip_valid = False

    def zero_grad(self, set_to_none=False):
        self.store.zero_grad()

    def state_dict(self):
        """flat-layout state (copies); not a torch.optim.AdamW state dict"""
        return {"step": self.step_count, "born": dict(self.born),
                "exp_avg": {g: t.clone() for g, t in self.store.exp_avg.items()},
                "exp_avg_sq": {g: t.clone() for g, t in self.store.exp_avg_sq.items()},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        if "exp_avg" not in sd or not isinstance(sd["exp_avg"], dict) or "lm" not in sd["exp_avg"]:
            raise ValueError("FlatAdamW.load_state_dict: not a FlatAdamW state (a reference checkpoint's torch.optim.AdamW "
                             "state is per-parameter and is not interchangeable with the flat layout)")
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
