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


def _load(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_small():
    return _load("small_stacked.npz")


@pytest.fixture(scope="session")
def golden_set_a():
    return _load("set_a_h720.npz")


@pytest.fixture(scope="session")
def golden_train():
    return _load("train_small.npz")


@pytest.fixture(scope="session")
def golden_inverse():
    return _load("inverse_small.npz")


@pytest.fixture(scope="session")
def golden_embvar():
    return _load("embedder_variants.npz")


@pytest.fixture(scope="session")
def golden_soma():
    return _load("somatosensory_small.npz")


def state_dict_from(g, prefix):
    import torch
    return {k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix + "/")}
