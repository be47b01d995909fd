import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # gpu tests are selected with `-m gpu`; when selected on a box without a GPU they must FAIL loudly,
    # not skip -- so no auto-skip here.
    pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]

    return load


@pytest.fixture(autouse=True)
def _poison_codes_only_carriers(request):
    """GPU tests run with every codes-only carrier tensor NaN-filled (FQSS_DEBUG_CARRIER=0 turns it off): a consumer that reads a
    carrier instead of its codes turns a parity check into a NaN (the bug class of the row linears' weight gradient, round 1)"""
    if request.node.get_closest_marker("gpu") is None or os.environ.get("FQSS_DEBUG_CARRIER", "1") == "0":
        yield
        return
    from fqss_amd import ops
    prev, ops.DEBUG_POISON = ops.DEBUG_POISON, True
    yield
    ops.DEBUG_POISON = prev
