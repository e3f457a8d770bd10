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


# ---- fp16 mode ------------------------------------------------------------------------------------
def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def fp16_fixture():
    z = np.load(os.path.join(GOLDEN, 'fp16_e2e_small.npz'))
    d = {k: torch.from_numpy(z[k]) for k in z.files}
    xs = [d[f'x{i}'].clone().requires_grad_(True) for i in range(4)]
    params = {tag: {k[len(tag):]: v.float().requires_grad_(True) for k, v in d.items() if k.startswith(tag)}
              for tag in ('fpn.', 'neck.', 'head.')}
    return d, xs, params


# How far the reference's own half arithmetic (fixture: reference modules run in half on the CPU) may sit from an fp32
# evaluation of the same half-stored values: forward a few half ulps (2^-11 = 4.9e-4 each), gradients several percent
# (its backward accumulates in half).  Measured on this fixture: forward <= 1.5e-3 max-norm, dx <= 4.4e-2 L2,
# parameter gradients <= 5.2e-2 L2 where the gradient is not tiny.
FP16_FWD_TOL = 3e-3
FP16_GRAD_TOL = 8e-2


def check_fp16_grads(d, xs_grads, param_grads):
    for i, g in enumerate(xs_grads):
        assert rel_l2(g, d[f'dx{i}']) <= FP16_GRAD_TOL, i
    for tag, grads in param_grads.items():
        for k, g in grads.items():
            ref = d['d' + tag + k]
            if g is None:
                g = torch.zeros_like(ref)
            if k == 'refine.phi.conv.bias' or ref.float().abs().max() < 0.25:
                continue        # softmax-invariant bias (pure rounding noise) / gradients at the half noise floor
            assert rel_l2(g, ref) <= FP16_GRAD_TOL, tag + k
