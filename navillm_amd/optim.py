"""Optimizer step of the measured path: clip_grad_norm_(params, 40.) + AdamW (train.py:86-89,
tools/optims.py:43-45: AdamW over ALL parameters, default betas/eps/weight_decay, constant LR).

One fused HIP kernel per flat buffer (K13): the bf16 LM buffer carries bf16 moments, exactly as
torch.optim.AdamW does for bf16 parameters in the reference; the clip coefficient is produced on
the device (no host sync) and folded into the same pass."""
import torch
from . import ops


class FlatAdamW:
    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=40.0):
        self.model = model
        self.store = model.store
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self.store.init_optimizer_state()
        self._clip = torch.ones(2, dtype=torch.float32, device=self.store.device)
        self.param_groups = [{"lr": lr, "params": list(model.parameters())}]

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm=None):
        """Returns the device tensor [total_norm, coef]; the scaling itself is applied inside step()."""
        mn = self.max_grad_norm if max_norm is None else max_norm
        ops.clip_coef([self.store.grad["lm"], self.store.grad["f32"]], mn, out2=self._clip)
        self._clip_valid = True
        return self._clip

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        clip = self._clip if getattr(self, "_clip_valid", False) else None
        for g in ("lm", "f32"):
            ops.adamw_(self.store.param[g], self.store.grad[g], self.store.exp_avg[g], self.store.exp_avg_sq[g],
                       self.step_count, lr, self.betas[0], self.betas[1], self.eps, self.wd, clip=clip)
        self._clip_valid = False

    def zero_grad(self, set_to_none=False):
        self.store.zero_grad()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.store.exp_avg, "exp_avg_sq": self.store.exp_avg_sq, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        for g in ("lm", "f32"):
            self.store.exp_avg[g].copy_(sd["exp_avg"][g])
            self.store.exp_avg_sq[g].copy_(sd["exp_avg_sq"][g])
