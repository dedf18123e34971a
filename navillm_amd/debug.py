"""Debug instrumentation for the persistent HBM buffers of the product path (`NAVILLM_POISON=1`; VERDICT r5 "next" #1).

The path keeps its activations in grow-only slabs that are carved once and re-used for the life of the process (the activation arena,
the per-layer episode buffers, the K/V slabs, the kernels' workspaces): an element that is read before this launch sequence wrote it,
or a kernel that writes past the end of its buffer, does not fail -- it computes with whatever the previous user left there, and the
result then depends on what ran earlier in the process.  With `NAVILLM_POISON=1`:

  * every device allocation made from Python through `torch.empty / empty_like` comes back NaN-filled (floating dtypes), so a read of
    never-written memory reaches the logits / gradients as NaN whatever the allocator handed out (a fresh process gets zero pages
    from the driver, a late test gets another test's data: the difference behind an order-dependent failure);
  * every such allocation (`torch.zeros / zeros_like` too) carries a 4 KiB canary on both sides; `check_guards()` -- called by
    `PrefixEpisode.finish()`, `NavModel.episode_release()` and after every GPU test (tests/conftest.py) -- verifies the canaries of
    every live buffer and names the one that was overrun;
  * buffers are re-poisoned when their owner releases them (`PrefixEpisode` at the end of an episode, the activation arena when the
    next forward takes it over, the K/V cache at `reset()`): state that survives from one episode to the next becomes NaN instead of a
    plausible number.

Off by default (the fills cost launches); nothing here is on the measured path.
"""
import os
import weakref

import torch

POISON = os.environ.get("NAVILLM_POISON", "0") not in ("0", "")
GUARD = 4096            # bytes of canary on each side of a guarded allocation
_PAT = 0xA5
_FLOAT = (torch.float32, torch.bfloat16, torch.float16, torch.float64)

_orig = {}
_registry = []          # (weakref to the payload view, payload bytes, tag)
stats = {"guarded": 0, "checks": 0, "poisoned": 0}


SENTINEL = 1.0e4        # the finite poison of buffers that are read UNDER A MASK by design (see poison_)


def poison_(t, masked_reads=False):
    """overwrite a floating-point tensor with NaN (integer tensors are left alone: a poisoned index would fault the GPU).
    masked_reads: the buffer is one the kernels read under a mask BY DESIGN -- the K/V-cache layout: the strided attention kernels run
    every sample over the batch's longest length, and a shorter sample's rows beyond its own length enter as keys that the causal mask
    zeroes (P = 0) and as queries whose outputs nobody gathers (dO = 0).  `0 x NaN` would turn that design into NaN, so these slabs
    get a large FINITE sentinel instead: multiplied by the zeros it is harmless, read for real it throws the logits off by orders of
    magnitude (every parity test fails)."""
    if POISON and t is not None and t.dtype in _FLOAT and t.numel():
        t.fill_(SENTINEL if masked_reads else float("nan"))
        stats["poisoned"] += 1
    return t


def guard_bytes(nbytes):
    """canary size per side: 4 KiB, growing with the buffer up to 1 MiB (a row of the big slabs is 8-64 KiB: an overrun by whole
    rows must still land in the canary)"""
    return max(GUARD, min(1 << 20, (nbytes // 64 + 4095) // 4096 * 4096))


def _guarded(shape, dtype, device, fill, requires_grad=False, tag=""):
    shape = tuple(int(s) for s in shape)
    n = 1
    for s in shape:
        n *= s
    nbytes = n * dtype.itemsize
    pad = (-nbytes) % 512                    # the upper canary starts on a 512-B boundary, as the allocator's blocks do
    G = guard_bytes(nbytes)
    base = _orig["empty"]((G + nbytes + pad + G,), dtype=torch.uint8, device=device)
    base[:G].fill_(_PAT)
    base[G + nbytes:].fill_(_PAT)
    t = base[G:G + nbytes].view(dtype).view(shape)
    if fill == "zero":
        t.zero_()
    elif dtype in _FLOAT and n:
        t.fill_(float("nan"))
    if requires_grad:
        t.requires_grad_(True)
    if len(_registry) > 20000:
        _prune()
    _registry.append((weakref.ref(t), nbytes, tag))
    stats["guarded"] += 1
    return t


def _raw(t):
    """the whole storage behind a guarded payload view, as bytes (canary | payload | pad | canary)"""
    return _orig["empty"]((0,), dtype=torch.uint8, device=t.device).set_(t.untyped_storage())


def _prune():
    _registry[:] = [e for e in _registry if e[0]() is not None]


def _is_cuda(dev):
    try:
        return dev is not None and torch.device(dev).type == "cuda"
    except Exception:
        return False


def _shape_of(size):
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
        return tuple(size[0])
    return tuple(size)


def _wrap_new(name, fill):
    orig = _orig[name]

    def f(*size, **kw):
        if not _is_cuda(kw.get("device")) or kw.get("out") is not None or kw.get("pin_memory") or kw.get("layout") not in (None, torch.strided) \
                or kw.get("memory_format") not in (None, torch.contiguous_format) or kw.get("names") is not None:
            return orig(*size, **kw)
        shape = _shape_of(size if size else (kw.get("size"),))
        if any(not isinstance(s, int) for s in shape):
            shape = tuple(int(s) for s in shape)
        return _guarded(shape, kw.get("dtype") or torch.get_default_dtype(), kw["device"], fill, bool(kw.get("requires_grad", False)))
    f.__name__ = name
    return f


def _wrap_like(name, fill):
    orig = _orig[name]

    def f(x, **kw):
        if not (torch.is_tensor(x) and _is_cuda(x.device)) or not x.is_contiguous() or kw.get("layout") not in (None, torch.strided) or \
                kw.get("memory_format") not in (None, torch.preserve_format, torch.contiguous_format) or \
                (kw.get("device") is not None and not _is_cuda(kw.get("device"))) or kw.get("pin_memory"):
            return orig(x, **kw)
        return _guarded(tuple(x.shape), kw.get("dtype") or x.dtype, kw.get("device") or x.device, fill, bool(kw.get("requires_grad", False)))
    f.__name__ = name
    return f


def install():
    """patch the four Python-level allocation entry points (idempotent).  Only allocations made from Python see this -- exactly the
    ones navillm_amd/ (and the tests) make; torch's own C++ allocations are untouched."""
    if _orig:
        return
    for n in ("empty", "zeros", "empty_like", "zeros_like"):
        _orig[n] = getattr(torch, n)
    torch.empty = _wrap_new("empty", "nan")
    torch.zeros = _wrap_new("zeros", "zero")
    torch.empty_like = _wrap_like("empty_like", "nan")
    torch.zeros_like = _wrap_like("zeros_like", "zero")


def install_if_enabled():
    if POISON:
        install()


def check_guards(where=""):
    """verify the canaries of every live guarded buffer; raises naming the first one that was overrun.  No-op unless NAVILLM_POISON=1."""
    if not POISON or not _orig:
        return 0
    _prune()
    live = []
    bad = None
    for ref, nbytes, tag in _registry:
        t = ref()
        if t is None:
            continue
        base = _raw(t)
        G = guard_bytes(nbytes)
        if base.numel() < 2 * G + nbytes:
            continue
        live.append((t, base, nbytes, tag))
        b = (base[:G] != _PAT).any() | (base[G + nbytes:] != _PAT).any()
        bad = b if bad is None else (bad | b)
    stats["checks"] += 1
    if bad is not None and bool(bad.item()):
        for t, base, nbytes, tag in live:
            G = guard_bytes(nbytes)
            lo = int((base[:G] != _PAT).sum())
            hi = int((base[G + nbytes:] != _PAT).sum())
            if lo or hi:
                raise RuntimeError(f"NAVILLM_POISON: canary overrun {where}: buffer {tuple(t.shape)} {t.dtype} {tag!r} -- {lo} bytes changed below, "
                                   f"{hi} bytes changed above its {nbytes} payload bytes")
    return len(live)
