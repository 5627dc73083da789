import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def oracle_threads():
    """Host threads for the oracle.  More is not faster: on the GPU box (2 x 64-core EPYC, torch default 128 threads) one
    640x480 frame through the fp32 oracle takes 2.10 s with 128 threads, 0.79 s with 64, 0.24 s with 16, 0.30 s with 8
    (tests/oracle_threads_probe.py, profiles/r03c_oracle_threads.txt)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return int(os.environ.get("QUBER_ORACLE_THREADS", min(16, usable)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(oracle_threads())


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
