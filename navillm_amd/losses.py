"""Drop-in for `criterion = nn.CrossEntropyLoss(ignore_index=-100, reduction='sum')` (train.py:229),
used on the navigation/object logits at tasks/agents/mp3d_agent.py:750,823."""
import torch
from . import functions as Fn


class DeferredLogits:
    """`fuse_logits` of a navigation step whose LM forward has been DEFERRED (navillm_amd/episode.py, round 4): under imitation learning
    the rollout never reads the logits to choose an action (mp3d_agent.py:760-761), so the step's forward is postponed and ALL the steps of
    the episode go through the decoder as one batch.  What the rollout may do with the handle without it ever holding numbers:
      * `nav_logits / T`, `x * f` and the softmax family -- `nav_probs = torch.softmax(nav_logits / args.temperature, 1)`,
        mp3d_agent.py:732, is computed unconditionally but only `feedback == 'sample'` reads it -- stay LAZY (`LazyExpr`);
      * `criterion(handle, targets)` -> a `DeferredLoss`;
      * `.shape`, `.size()`, `.dim()`, `.device`, `.dtype`.
    Anything else needs the numbers.  In an episode the caller opened with `begin_episode(..., teacher_forced=True)` that is an error
    (`.value` / a forced expression raise until finish_episode(); a tensor method is an AttributeError): the caller promised not to
    read.  In an AUTOMATIC episode the handle is a `LazyLogits`, which runs the pending steps instead.  After finish_episode():
    `.value` = the [B, G] logits."""

    forceable = False

    def __init__(self, episode, rec, shape=None, device=None, dtype=None):
        self._episode, self._rec = episode, rec
        if shape is not None:
            self.shape, self.device, self.dtype = torch.Size(shape), device, dtype

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return len(self.shape)

    @property
    def value(self):
        v = self._rec.get("logits")
        if v is None:
            raise RuntimeError("the logits of a deferred (teacher-forced) step exist after model.finish_episode() -- inside an accumulation "
                               "window (begin_episode(..., accumulate=n)) after the window's last finish_episode() or model.flush_accumulation_window()")
        return v

    def force(self):
        return self.value

    def _lazy(self, fn):
        return LazyExpr(self, fn)

    def __truediv__(self, f):
        return self._lazy(lambda x: x / f)

    def __mul__(self, f):
        return self._lazy(lambda x: x * f)

    __rmul__ = __mul__

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _LAZY_FUNCS and len(args) >= 1 and isinstance(args[0], (DeferredLogits, LazyExpr)):
            rest = args[1:]
            return args[0]._lazy(lambda x: func(x, *rest, **kwargs))
        return func(*_force_all(args), **_force_all(kwargs))


class LazyLogits(DeferredLogits):
    """`fuse_logits` of a navigation step inside an AUTOMATIC episode (round 6; NavModel._auto_*): the unmodified rollout
    (tasks/agents/mp3d_agent.py:660-778) cannot say whether it will read the logits -- under teacher forcing it never does (the action is
    the teacher's, :760-761), under DAgger sampling / argmax it does (:762-769) -- so this handle decides by itself: what
    `DeferredLogits` keeps lazy stays lazy, and anything that needs numbers -- `.float()`, `.max(1)`, `Categorical(nav_probs.float())`,
    indexing, any other torch function -- FORCES the step: every navigation step of the episode that has not gone through the decoder yet
    does so now, as one batch (`PrefixEpisode.force_logits`).  Read BEFORE the step's loss, the handle answers with a real tensor that
    is connected to autograd (the `criterion(...).backward()` that follows reaches the step through it); read AFTER the loss was
    registered (the reference's order in a sampled rollout: loss and backward() at :750-757, the draw at :762-765) with the values.
    `loss.item()` (train.py:83) on deferred losses runs the batched forward of whatever is pending + the heads, not the backward.
    So a teacher-forced rollout batches the LM forward of all its steps (the headline form), a sampled one runs step by step, and the
    rollout is the reference's, line for line."""

    forceable = True

    @property
    def value(self):
        v = self._rec.get("logits")
        if v is None:
            self._episode.force_logits(self._rec, live=False)
            v = self._rec["logits"]
        return v

    def force(self):
        """the step's logits as a real tensor: the steps recorded so far run their LM forward now"""
        return self._episode.force_logits(self._rec, live=True)

    def __getattr__(self, name):
        # (only reached for attributes this class does not define: tensor methods -- they need the numbers)
        if name.startswith("__") or name in ("_episode", "_rec", "shape", "device", "dtype"):
            raise AttributeError(name)
        return getattr(self.force(), name)

    # indexing, iteration and the rest of the arithmetic need the numbers as well
    def __getitem__(self, i):
        return self.force()[i]

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(self.force())

    def __add__(self, o):
        return self.force() + o

    __radd__ = __add__

    def __sub__(self, o):
        return self.force() - o

    def __neg__(self):
        return -self.force()


class LazyExpr:
    """an elementwise expression over a deferred-logits handle that has not been evaluated (`nav_probs = torch.softmax(nav_logits / T, 1)`)"""

    def __init__(self, root, fn):
        self._root, self._fn = root, fn

    def force(self):
        return self._fn(self._root.force())

    def _lazy(self, fn):
        return LazyExpr(self._root, lambda x, f=self._fn: fn(f(x)))

    def __truediv__(self, f):
        return self._lazy(lambda x: x / f)

    def __mul__(self, f):
        return self._lazy(lambda x: x * f)

    __rmul__ = __mul__

    __torch_function__ = DeferredLogits.__torch_function__

    def __getattr__(self, name):
        if name.startswith("__") or name in ("_root", "_fn"):
            raise AttributeError(name)
        return getattr(self.force(), name)

    def __getitem__(self, i):
        return self.force()[i]

    def __iter__(self):
        return iter(self.force())

    def __add__(self, o):
        return self.force() + o

    __radd__ = __add__

    def __sub__(self, o):
        return self.force() - o


_LAZY_FUNCS = {torch.softmax, torch.nn.functional.softmax, torch.log_softmax, torch.nn.functional.log_softmax, torch.div, torch.true_divide,
               torch.mul}


def _force_all(x):
    if isinstance(x, (DeferredLogits, LazyExpr)):
        return x.force()
    if isinstance(x, (list, tuple)):
        return type(x)(_force_all(v) for v in x)
    if isinstance(x, dict):
        return {k: _force_all(v) for k, v in x.items()}
    return x


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
        if v is None and getattr(self._logits, "forceable", False):
            # an automatic episode (LazyLogits): `loss.item()` after the rollout (train.py:83) runs the batched LM forward of the steps that
            # are still pending and the heads + losses -- not the backward, which waits for whoever reads `.grad`
            self._logits._episode.force_values()
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
        if isinstance(logits, LazyLogits) and logits._rec.get("logits_live") is not None:
            logits = logits._rec["logits_live"]        # the step was forced (its logits were read): an ordinary loss on the real tensor
        elif isinstance(logits, LazyExpr):
            logits = logits.force()
        if isinstance(logits, DeferredLogits):
            return DeferredLoss(logits, targets)
        return Fn.ActionCE.apply(logits, targets.to(device=logits.device, dtype=torch.int64).contiguous())
