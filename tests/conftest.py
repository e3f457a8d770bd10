import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


import numpy as np
import torch


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


def relerr(a, b):
    """max|a-b| / max|b| — the 'within 1e-3 rel fp32' measure of BASELINE.json's north_star."""
    if a is None:
        a = torch.zeros_like(b)
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def relerr_l2(a, b):
    """||a-b||_2 / ||b||_2.  Used next to `relerr` where tensors are large enough for ReLU kinks to matter: two fp32
    implementations round a pre-activation of magnitude ~1e-7 to different signs about twice per million elements
    (measured: 2 of 1 003 520 in each conv output of the mid-size head), which flips that element's mask and moves its
    gradient by the full value - a max-norm outlier that is not an arithmetic error."""
    if a is None:
        a = torch.zeros_like(b)
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).norm().item() / max(b.norm().item(), 1e-300)
