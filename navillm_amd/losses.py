"""Drop-in for `criterion = nn.CrossEntropyLoss(ignore_index=-100, reduction='sum')` (train.py:229),
used on the navigation/object logits at tasks/agents/mp3d_agent.py:750,823."""
import torch
from . import functions as Fn


class DeferredLogits:
    """`fuse_logits` of a navigation step inside a TEACHER-FORCED prefix-reuse episode (navillm_amd/episode.py, round 4): under
    imitation learning the rollout never reads the logits to choose an action (mp3d_agent.py:760-761), so the step's LM forward is
    postponed to `finish_episode()`, where ALL the steps of the episode go through the decoder as one batch.  The only thing the
    rollout may do with this handle is hand it to the criterion; anything else (argmax, softmax, .cpu()) is an AttributeError -- a
    sampling / argmax rollout must not open the episode with teacher_forced=True.  After finish_episode(): `.value` = the [B, G]
    logits."""

    def __init__(self, episode, rec):
        self._episode, self._rec = episode, rec

    @property
    def value(self):
        v = self._rec.get("logits")
        if v is None:
            raise RuntimeError("the logits of a deferred (teacher-forced) step exist after model.finish_episode() -- inside an accumulation "
                               "window (begin_episode(..., accumulate=n)) after the window's last finish_episode() or model.flush_accumulation_window()")
        return v


class DeferredLoss:
    """criterion(DeferredLogits, targets): supports exactly what the rollout does with a step loss -- scale it by python numbers
    (`* train_ml / batch_size / accum`, mp3d_agent.py:750), `.backward()` (records targets and scale: the step's gradient is produced
    by finish_episode()), `.detach()`; `float()` / `.item()` / `.value` work after finish_episode()."""

    def __init__(self, logits, targets, scale=1.0):
        self._logits, self._targets, self._scale = logits, targets, float(scale)

    def _scaled(self, f):
        return DeferredLoss(self._logits, self._targets, self._scale * float(f))

    def __mul__(self, f):
        return self._scaled(f)

    __rmul__ = __mul__

    def __truediv__(self, f):
        return self._scaled(1.0 / float(f))

    def backward(self):
        self._logits._episode.register_loss(self._logits._rec, self._targets, self._scale)

    def detach(self):
        return self

    # `cnt_loss = 0.; cnt_loss += criterion(...) * w` and `ml_loss += cnt_loss.detach()` (mp3d_agent.py:750-752) keep working
    def __add__(self, other):
        return DeferredSum([self]) + other

    __radd__ = __add__

    @property
    def value(self):
        v = self._logits._rec.get("loss_sum")
        if v is None:
            raise RuntimeError("the value of a deferred (teacher-forced) step loss exists after model.finish_episode() -- inside an "
                               "accumulation window (begin_episode(..., accumulate=n)) after the window's last finish_episode() or "
                               "model.flush_accumulation_window()")
        return v * self._scale

    def item(self):
        return float(self.value)

    __float__ = item


class DeferredSum:
    """a sum of deferred step losses (+ plain numbers): the running totals a rollout keeps (`ml_loss += cnt_loss.detach()`)"""

    def __init__(self, terms, const=0.0):
        self._terms, self._const = list(terms), float(const)

    def __add__(self, other):
        if isinstance(other, DeferredLoss):
            return DeferredSum(self._terms + [other], self._const)
        if isinstance(other, DeferredSum):
            return DeferredSum(self._terms + other._terms, self._const + other._const)
        return DeferredSum(self._terms, self._const + float(other))

    __radd__ = __add__

    def __mul__(self, f):
        return DeferredSum([t * f for t in self._terms], self._const * float(f))

    __rmul__ = __mul__

    def __truediv__(self, f):
        return self * (1.0 / float(f))

    def backward(self):
        for t in self._terms:
            t.backward()

    def detach(self):
        return self

    @property
    def value(self):
        return self._const + sum(float(t.value) for t in self._terms)

    def item(self):
        return float(self.value)

    __float__ = item


class CrossEntropyLoss(torch.nn.Module):
    def __init__(self, ignore_index=-100, reduction="sum"):
        super().__init__()
        if ignore_index != -100 or reduction != "sum":
            raise NotImplementedError("the navigation path uses ignore_index=-100, reduction='sum'")

    def forward(self, logits, targets):
        if isinstance(logits, DeferredLogits):
            return DeferredLoss(logits, targets)
        return Fn.ActionCE.apply(logits, targets.to(device=logits.device, dtype=torch.int64).contiguous())
