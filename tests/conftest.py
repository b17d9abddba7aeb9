import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("PN2_TEST_GC_STRESS") == "1":
        # stress: the cyclic collector runs after every few allocations -- a finaliser that must not run inside a graph
        # capture (ops.capture_region) shows at once instead of once in ten runs
        import gc
        gc.set_threshold(20, 1, 1)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def orc():
    from oracle import pn2_oracle
    pn2_oracle.build()
    return pn2_oracle


@pytest.fixture(scope="session")
def synth():
    from khairil_tum_facade_semantic_segmentation_amd import synth as s
    return s


@pytest.fixture(autouse=True)
def _finalize_between_gpu_tests(request):
    """After every GPU test: wait for the device and collect garbage NOW.  Trainers and inference engines own captured
    hipGraphs, events and pool memory; left to the cyclic collector they are finalised at an arbitrary later allocation --
    possibly in the middle of the next test's graph capture, where the HIP calls of those finalisers abort the process."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass
        gc.collect()
