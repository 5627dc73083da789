import glob
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


def golden(prefix):
    files = sorted(glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")))
    assert files, f"no golden fixtures for {prefix}"
    return files


def load_encode_case(path):
    z = np.load(path)
    n, h, w = z["shape"]
    masks = np.unpackbits(z["masks"], axis=-1)[..., :w].astype(np.uint8) * np.uint8(z["value"][0])
    return masks, z["out"], z["fg"]


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
