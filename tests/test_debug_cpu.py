"""CPU: the host logic of the debug instrumentation (navillm_amd/debug.py, NAVILLM_POISON=1) and of the test-order option
(tests/conftest.py --nv-order) -- the canaries, the NaN fill and the seeded shuffle, exercised on host tensors."""
import pytest
import torch

import conftest as C


@pytest.fixture
def dbg(monkeypatch):
    from navillm_amd import debug
    monkeypatch.setattr(debug, "POISON", True)
    monkeypatch.setattr(debug, "_is_cuda", lambda dev: dev is not None)        # treat the CPU as "the device" for this test
    saved = {n: getattr(torch, n) for n in ("empty", "zeros", "empty_like", "zeros_like")}
    debug._orig.clear()
    debug._registry.clear()
    debug.install()
    yield debug
    for n, f in saved.items():
        setattr(torch, n, f)
    debug._orig.clear()
    debug._registry.clear()


def test_poisoned_allocations_are_nan_and_canaried(dbg):
    a = torch.empty((5, 7), dtype=torch.float32, device="cpu")
    assert a.shape == (5, 7) and bool(torch.isnan(a).all()) and a.is_contiguous()
    b = torch.empty(3, 4, dtype=torch.bfloat16, device="cpu")
    assert b.shape == (3, 4) and bool(torch.isnan(b.float()).all())
    z = torch.zeros((9,), dtype=torch.float32, device="cpu", requires_grad=True)
    assert bool((z == 0).all()) and z.requires_grad and z.is_leaf
    i = torch.empty((6,), dtype=torch.int32, device="cpu")                      # integers are guarded but not poisoned
    assert i.shape == (6,)
    e = torch.empty_like(a)
    assert e.shape == a.shape and bool(torch.isnan(e).all())
    zl = torch.zeros_like(b)
    assert float(zl.float().abs().sum()) == 0.0
    n0 = torch.empty((0, 4), dtype=torch.float32, device="cpu")
    assert n0.numel() == 0
    assert dbg.check_guards("t") >= 6
    # allocations without a device argument (host bookkeeping) are untouched
    h = torch.zeros(4)
    assert h.untyped_storage().nbytes() == 16


def test_canary_overrun_is_reported_with_the_buffer(dbg):
    a = torch.empty((16,), dtype=torch.float32, device="cpu")
    dbg.check_guards("clean")
    dbg._raw(a)[dbg.GUARD + 16 * 4 + 3] = 0                                         # one byte past the payload
    with pytest.raises(RuntimeError, match=r"canary overrun here.*\(16,\).*1 bytes changed above"):
        dbg.check_guards("here")
    dbg._raw(a)[dbg.GUARD + 16 * 4 + 3] = 0xA5
    dbg._raw(a)[dbg.GUARD - 1] = 1                                                  # one byte below
    with pytest.raises(RuntimeError, match="1 bytes changed below"):
        dbg.check_guards("there")


def test_dead_buffers_are_pruned(dbg):
    for _ in range(50):
        torch.empty((8,), dtype=torch.float32, device="cpu")
    keep = torch.empty((8,), dtype=torch.float32, device="cpu")
    assert dbg.check_guards("x") == 1 and len(dbg._registry) == 1
    del keep


def test_poison_is_a_noop_when_disabled(monkeypatch):
    from navillm_amd import debug
    monkeypatch.setattr(debug, "POISON", False)
    t = torch.ones(4)
    assert debug.poison_(t) is t and float(t.sum()) == 4.0
    assert debug.check_guards("off") == 0


def test_order_option():
    items = list(range(10))
    assert C.reorder(items, "") == items
    assert C.reorder(items, "reverse") == items[::-1]
    a, b = C.reorder(items, "random:3"), C.reorder(items, "random:3")
    assert a == b and sorted(a) == items and a != items
    assert C.reorder(items, "random:4") != a
    with pytest.raises(ValueError):
        C.reorder(items, "sideways")
