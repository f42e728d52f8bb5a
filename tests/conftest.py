import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests never run by accident on a box without a device."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rnd(seed, *shape, scale=1.0):
    """Same frozen stream tests/golden/make_golden.py draws its inputs from."""
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def sd_from_golden(g, prefix):
    """Rebuild a torch state_dict stored by make_golden.sd_np under ``prefix``."""
    import torch
    sd = {}
    for k in g.files:
        if k.startswith(prefix):
            sd[k[len(prefix):].replace("__", ".")] = torch.from_numpy(np.array(g[k]))
    return sd


@pytest.fixture(scope="session")
def golden():
    return load_golden
