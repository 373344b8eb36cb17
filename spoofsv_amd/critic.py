"""WGAN-GP critics of the reference (``models/discriminator.py:6-80``), SURVEY.md section 8(f) row 1.

The critics need the gradient of a gradient (the penalty of train/adversarial_wasserstein_gp.py:300-308), which the
generator kernels do not.  On a ROCm device their heavy operators run on HIP kernels that are differentiable twice:
  * convolutions -- ``ops.conv1d_dd`` (forward, data gradient and weight gradient are closed under differentiation);
  * LayerNorm over channels and the highway gate -- ``ops.channel_ln_dd`` / ``ops.highway_gate_dd``: the fused forward and
    first-order backward kernels of the generator path plus hand-written second-order kernels (``ssv_channel_ln_bwd2``,
    ``ssv_highway_gate_bwd2``).
Dropout (its own Philox stream), leaky-ReLU and the average pools are HIP kernels too (csrc/critic.hip): each is piecewise
linear, so 'multiply by the saved factor' / 'adjoint pool' makes them differentiable to any order.
There is no CPU branch: a CPU tensor raises (the stock-op restatement used as the parity arm is oracle/critic_oracle.py).
Same sub-module names as the reference, so ``disc_state_dict`` checkpoints interchange.  Dropout (p=0.05) is
active whenever the module is in training mode, as in the reference (which never calls ``disc.eval()``).
"""
import contextlib

import torch
import torch.nn as nn

from . import ops

_P_DROP = 0.05
_MASKS = None          # test hook: a list of pre-scaled keep masks consumed in call order instead of drawing from the RNG


@contextlib.contextmanager
def injected_dropout_masks(masks):
    """Within the block every dropout of the critics multiplies by the next tensor of ``masks`` (already scaled by
    1/(1-p), i.e. exactly what ``nn.Dropout`` multiplies by) instead of drawing one.  The reference never puts its critics
    in eval mode, so parity with dropout ACTIVE needs the same masks on both sides (tests, oracle/critic_oracle.py)."""
    global _MASKS
    prev, _MASKS = _MASKS, list(masks)
    try:
        yield
    finally:
        _MASKS = prev


def _act(x, training, slope=1.0, drop=True):
    """dropout(leaky_relu(x, slope)) as ONE HIP kernel (ops.act_dropout); with injected masks the two factors are applied
    separately (leaky-ReLU kernel, then the given mask)."""
    if _MASKS is not None and drop:
        return ops.mul_const(ops.act_dropout(x, slope, 0.0), _MASKS.pop(0))
    return ops.act_dropout(x, slope, _P_DROP if (drop and training) else 0.0)


def _conv(conv, x):
    """nn.Conv1d forward on the HIP conv kernels through ``ops.conv1d_dd`` (differentiable to any order, as the gradient
    penalty needs; MIOpen's fp32 1-D convolutions were 23 % of a critic iteration)."""
    ops._dev(x, "critic input")
    if conv.kernel_size[0] not in (1, 3) or conv.stride[0] != 1:
        raise RuntimeError("spoofsv_amd.critic: only stride-1 convolutions of kernel size 1 or 3 exist on the HIP path")
    return ops.conv1d_dd(x, conv.weight, conv.bias, conv.kernel_size[0], conv.dilation[0], False)


class _HighwayConvDropout(nn.Module):
    """highwayConv of models/TTSModel_dropout.py:37-84 (the variant the critics import, discriminator.py:4)."""

    def __init__(self, dimension, kernel_size, dilation):
        super().__init__()
        self.dimension = dimension
        pad = dilation * (kernel_size - 1) // 2
        self.conv = nn.Conv1d(dimension, 2 * dimension, kernel_size, padding=pad, dilation=dilation)
        self.ln1 = nn.LayerNorm(dimension)
        self.ln2 = nn.LayerNorm(dimension)
        self.dp = nn.Dropout(p=_P_DROP)

    def forward(self, x):
        h = _conv(self.conv, x)
        return _act(ops.highway_gate_dd(h, x, self.ln1.weight, self.ln1.bias, self.ln2.weight, self.ln2.bias), self.training)


def _ln(x, ln):
    """nn.LayerNorm over the channel axis of a (B, C, T) tensor.  The reference permutes to (B, T, C) and back
    (discriminator.py:24-27); here the tensor stays (B, C, T) and the fused HIP LayerNorm runs on it."""
    if ln.eps != 1e-5:
        raise RuntimeError("spoofsv_amd.critic: the HIP LayerNorm is built for eps = 1e-5")
    return ops.channel_ln_dd(x, ln.weight, ln.bias)


class _Disc(nn.Module):
    def __init__(self, freq_bins, disc_dim, pool1, pool2, last):
        super().__init__()
        self.conv1 = nn.Conv1d(freq_bins, disc_dim, 1)
        self.ln1 = nn.LayerNorm(disc_dim)
        self.dp1 = nn.Dropout(p=_P_DROP)
        self.hc = _HighwayConvDropout(disc_dim, 3, 1)
        self.conv2 = nn.Conv1d(disc_dim, 64, 1)
        self.pl1 = nn.AvgPool1d(kernel_size=pool1)
        self.ln2 = nn.LayerNorm(64)
        self.dp2 = nn.Dropout(p=_P_DROP)
        self.conv3 = nn.Conv1d(64, 16, 1)
        self.pl2 = nn.AvgPool1d(kernel_size=pool2)
        self.ln3 = nn.LayerNorm(16)
        self.conv4 = nn.Conv1d(16, last, 1)
        self.ln4 = nn.LayerNorm(last)
        self.conv5 = nn.Conv1d(last, 1, 1)
        self.pl3 = nn.AdaptiveAvgPool1d(output_size=1)

    def forward(self, inputs):
        tr = self.training
        x = _act(_ln(_conv(self.conv1, inputs), self.ln1), tr)
        x = self.hc(x)
        x = _ln(ops.avg_pool1d(_conv(self.conv2, x), self.pl1.kernel_size[0]), self.ln2)
        x = _act(x, tr, slope=0.05)                                       # dp2(leaky_relu(.)), one kernel
        x = _ln(ops.avg_pool1d(_conv(self.conv3, x), self.pl2.kernel_size[0]), self.ln3)
        x = _ln(_conv(self.conv4, _act(x, tr, slope=0.05, drop=False)), self.ln4)
        x = _conv(self.conv5, _act(x, tr, slope=0.05, drop=False))
        return ops.avg_pool1d(x, x.shape[-1])                              # AdaptiveAvgPool1d(1); no sigmoid: Wasserstein critic


class melDisc(_Disc):
    """models/discriminator.py:6-42."""

    def __init__(self, freq_bins, disc_dim):
        super().__init__(freq_bins, disc_dim, pool1=4, pool2=2, last=4)


class linDisc(_Disc):
    """models/discriminator.py:44-80."""

    def __init__(self, freq_bins, disc_dim):
        super().__init__(freq_bins, disc_dim, pool1=8, pool2=4, last=8)
