"""Deterministic tensor / parameter generators shared by the golden-vector
generator (make_golden.py, runs only where /root/reference exists) and by the
tests (run anywhere).  Nothing here touches the reference.

Values come from numpy's legacy ``RandomState`` (MT19937, frozen stream
definition), so the same (shape, seed) gives bit-identical float32 data on any
machine.  Parameters of a module are filled by *name* (crc32 of the
state_dict key), so a module with the reference's state_dict keys gets the
reference's weights without storing them in a fixture.
"""
import zlib

import numpy as np
import torch


def _rs(seed):
    return np.random.RandomState(int(seed) & 0x7FFFFFFF)


def uniform(shape, seed, lo=0.0, hi=1.0):
    """float32 tensor, U[lo,hi)."""
    n = int(np.prod(shape))
    a = _rs(seed).random_sample(n).astype(np.float32) * (hi - lo) + lo
    return torch.from_numpy(a.reshape(shape).astype(np.float32))


def normal(shape, seed, std=1.0, mean=0.0):
    n = int(np.prod(shape))
    a = _rs(seed).standard_normal(n) * std + mean
    return torch.from_numpy(a.reshape(shape).astype(np.float32))


def bits(shape, seed):
    """{0,1} float32 message tensor."""
    n = int(np.prod(shape))
    a = (_rs(seed).random_sample(n) < 0.5).astype(np.float32)
    return torch.from_numpy(a.reshape(shape))


def key_seed(key, salt=0):
    return (zlib.crc32(key.encode()) + 7919 * salt) & 0x7FFFFFFF


@torch.no_grad()
def fill_module(module, salt=0):
    """Fill every parameter / buffer of `module` deterministically by key.

    conv / linear weights ~ N(0, sqrt(2/fan_in)); every bias ~ N(0, 0.1);
    BatchNorm weight ~ 1 + N(0,0.1); running stats are left at
    their defaults (0 / 1) so the first training-mode forward updates them the
    way a fresh reference module would.
    """
    sd = module.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            continue
        s = key_seed(k, salt)
        if v.dim() >= 2:
            fan_in = int(np.prod(v.shape[1:]))
            v.copy_(normal(tuple(v.shape), s, std=float(np.sqrt(2.0 / fan_in))))
        elif k.endswith("weight"):  # 1-D weight == BatchNorm gamma
            v.copy_(normal(tuple(v.shape), s, std=0.1, mean=1.0))
        else:
            v.copy_(normal(tuple(v.shape), s, std=0.1))
    return module


def subsample(t, stride=97):
    """Every `stride`-th element of the flattened tensor (fixture size control)."""
    return t.detach().reshape(-1)[::stride].clone()


@torch.no_grad()
def fill_f1(net):
    """fill_module for the row-f1 networks, then: spectral-norm u / v normalised like torch leaves them; the Bayar filter positive
    (its constraint divides by the filter sum)."""
    fill_module(net)
    for k, v in net.state_dict().items():
        if k.endswith("weight_u") or k.endswith("weight_v"):
            v.copy_(torch.nn.functional.normalize(v, dim=0, eps=1e-12))
        if k == "BayarConv2D.weight":
            v.copy_(uniform(tuple(v.shape), key_seed(k, 7)) + 0.5)
    return net


@torch.no_grad()
def fill_f2(net):
    """fill_module for the invertible embedder: the fixed Haar filters keep their values; each subnet's last conv (zero at the
    reference's initialisation) is scaled down so ten stacked couplings stay well inside fp32 range."""
    keep = {k: v.clone() for k, v in net.state_dict().items() if k.endswith("haar_weights")}
    fill_module(net)
    for k, v in net.state_dict().items():
        if k in keep:
            v.copy_(keep[k])
        elif k.endswith("conv5.weight"):
            v.mul_(0.2)
    return net
