import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_gpu_memory_between_tests(request):
    """the GPU suite runs in ONE process and several tests build a full Vicuna-7B (+ 13B) model with its flat stores, activation arena
    and episode buffers; models sit in reference cycles (autograd functions <-> model <-> episode), so without a collection the next
    big test finds 250 GiB still allocated (seen: torch.OutOfMemoryError in test_full_vicuna_7b_training_step_invariants, round 4)"""
    yield
    if "gpu" in request.keywords:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
