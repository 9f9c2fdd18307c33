import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


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


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _reset_compute_dtype():
    """csl_gan_amd.ops.set_compute_dtype is process-wide state (a Trainer sets it from --compute_dtype): every test starts
    and ends on the default exact-fp32 kernels."""
    yield
    import sys
    ops = sys.modules.get("csl_gan_amd.ops")
    if ops is not None:
        ops.set_compute_dtype("fp32")
        ops.set_storage_dtype("fp32")


@pytest.fixture(autouse=True)
def _collect_garbage_between_tests():
    """A test's Trainer / GraphedDStep (HIP graph, its private memory pool, side streams, double-backward autograd graphs) dies as
    cyclic garbage.  Collect it HERE, on the main thread with the device idle, instead of whenever the next test's allocation count
    trips the collector — which can be inside an autograd worker thread in the middle of a backward."""
    yield
    import gc
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.synchronize()
    gc.collect()
