import os
import random
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    # the GPU suite runs in ONE process over persistent HBM slabs, so a defect that leaves state behind shows up
    # only in some orders.  `--nv-order reverse | random:SEED` (or NAVILLM_TEST_ORDER) re-orders the collected tests; with
    # NAVILLM_POISON=1 (navillm_amd/debug.py) every order also runs over NaN-filled, canaried buffers.
    parser.addoption("--nv-order", action="store", default=os.environ.get("NAVILLM_TEST_ORDER", ""),
                     help="order of the collected tests: '' (file order), 'reverse', or 'random:SEED'")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def reorder(items, spec):
    """'' -> unchanged; 'reverse'; 'random:SEED' (seeded shuffle) -- of whole items, so parametrised cases are shuffled too"""
    if not spec:
        return items
    if spec == "reverse":
        return list(reversed(items))
    if spec.startswith("random"):
        seed = int(spec.split(":", 1)[1]) if ":" in spec else 0
        out = list(items)
        random.Random(seed).shuffle(out)
        return out
    raise ValueError(f"--nv-order {spec!r}: expected 'reverse' or 'random:SEED'")


def pytest_collection_modifyitems(config, items):
    import torch
    items[:] = reorder(items, config.getoption("--nv-order"))
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """printed in -q runs too: what the debug instrumentation did (tests/test_episode_isolation_gpu.py reads it from its child process)"""
    try:
        from navillm_amd import debug
    except Exception:
        return
    if debug.POISON:
        terminalreporter.write_line(f"navillm_amd poison: {debug.stats['guarded']} guarded allocations, {debug.stats['poisoned']} re-poisoned "
                                    f"buffers, {debug.stats['checks']} canary checks; order = {config.getoption('--nv-order') or 'file order'}")


def pytest_report_header(config):
    return [f"navillm_amd: test order = {config.getoption('--nv-order') or 'file order'}, NAVILLM_POISON = {os.environ.get('NAVILLM_POISON', '0')}"]


@pytest.fixture(autouse=True)
def _explicit_episode_forms(request, monkeypatch):
    """NavModel opens AUTOMATIC prefix-reuse episodes for grad-enabled training-mode navigation calls (round 6; the product default).
    The parity tests written before it pin ONE named formulation each -- the reference's full-prompt recompute, the explicit episode
    forms -- and read `.grad` straight after backward(); they keep meaning what they say: automatic episodes are switched off for the
    suite (the constructor reads NAVILLM_AUTO_EPISODE) and the tests of the automatic path switch them on per model
    (tests/test_auto_episode_gpu.py, test_parity_gpu.py::test_g12_...[auto], bench.py's `unmodified_rollout`)."""
    if os.environ.get("NAVILLM_AUTO_EPISODE_IN_TESTS") != "1" and "test_automatic_episode_guards_and_opt_out" not in request.node.name:
        monkeypatch.setenv("NAVILLM_AUTO_EPISODE", "0")
    yield


@pytest.fixture(autouse=True)
def _release_gpu_memory_between_tests(request):
    """the GPU suite runs in ONE process and several tests build a full Vicuna-7B (+ 13B) model with its flat stores, activation arena
    and episode buffers; models sit in reference cycles (autograd functions <-> model <-> episode), so without a collection the next
    big test finds 250 GiB still allocated (seen: torch.OutOfMemoryError in test_full_vicuna_7b_training_step_invariants, round 4).
    NAVILLM_POISON=1: the canaries around every buffer that is still alive are verified after each test (navillm_amd/debug.py)."""
    yield
    if "gpu" in request.keywords:
        import gc
        import torch
        if torch.cuda.is_available():
            from navillm_amd import debug
            debug.check_guards(f"after {request.node.nodeid}")
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
