"""Helpers shared by the tests: load tests/golden fixtures as torch tensors."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(d, prefix, to_torch=True, device=None):
    """Entries of ``d`` under ``prefix`` with the prefix stripped."""
    out = {}
    for k, v in d.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(np.array(v)).to(device) if to_torch else v
    return out


def t(a, device=None):
    return torch.from_numpy(np.array(a)).to(device)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    """Relative error in the L2 norm, ||a - b|| / ||b||: unlike the max-norm ``rel_err`` it cannot be satisfied by a tensor
    whose few large entries are right while the bulk of small ones is wrong."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def worst_elementwise(a, b, floor=1e-3):
    """max_i |a_i - b_i| / (|b_i| + floor * rms(b)): an element-wise relative error with a floor tied to the tensor's own RMS,
    so that entries near zero are held to ``floor`` of a typical entry instead of to an impossible relative bound."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    rms = float(b.pow(2).mean().sqrt())
    return float(((a - b).abs() / (b.abs() + floor * rms + 1e-30)).max())
