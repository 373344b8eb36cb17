"""CPU oracle for the WGAN-GP critics.  TEST INFRASTRUCTURE ONLY.

A functional restatement, on stock torch CPU ops, of ``melDisc`` / ``linDisc`` in the reference's
``models/discriminator.py:6-80`` and of the dropout ``highwayConv`` they import (``models/TTSModel_dropout.py:37-84``),
plus the critic-side expressions of the trainer (``train/adversarial_wasserstein_gp.py:296-321``).  It is the checker for
``spoofsv_amd/critic.py``: only ``tests/`` may import it; the product package never imports anything under ``oracle/``.

Pinning: checked against golden vectors produced by importing the real reference critics in the build container, in
TRAINING mode (dropout active, as the reference always runs them) under ``torch.manual_seed`` -- ``oracle/gen_golden.py``
-> ``tests/golden/critic_dropout.npz``, ``tests/test_oracle_golden.py``.  The restatement issues ``F.dropout`` at the same
three places in the same order, so on the CPU the same seed draws the same masks.

All functions take a flat ``sd`` (the critic's own ``state_dict()`` keys).  Dropout: ``masks=None`` draws like
``nn.Dropout(p=0.05)`` in training mode and appends the (scaled) keep masks to ``drawn`` when given; ``masks=[...]``
multiplies by the given tensors in call order (what ``critic.injected_dropout_masks`` does on the HIP side);
``masks=False`` is eval mode (no dropout).
"""
import torch
import torch.nn.functional as F

P_DROP = 0.05
POOLS = {"mel": (4, 2), "lin": (8, 4)}           # discriminator.py:14,18 / :52,56


def _ln(x, sd, name):
    """discriminator.py:25 etc.: permute -> nn.LayerNorm over channels -> permute back."""
    w = sd[name + ".weight"]
    return F.layer_norm(x.permute(0, 2, 1), (w.shape[0],), w, sd[name + ".bias"], 1e-5).permute(0, 2, 1)


def _conv(x, sd, name, padding=0):
    return F.conv1d(x, sd[name + ".weight"], sd[name + ".bias"], padding=padding)


class _Drop:
    def __init__(self, masks, drawn):
        self.masks = list(masks) if isinstance(masks, (list, tuple)) else masks
        self.drawn = drawn

    def __call__(self, x):
        if self.masks is False:
            return x
        if self.masks is None:
            m = F.dropout(torch.ones_like(x), P_DROP, True)        # the noise tensor nn.Dropout multiplies by (same RNG draw)
            if self.drawn is not None:
                self.drawn.append(m)
            return x * m
        return x * self.masks.pop(0)


def critic(x, sd, kind="mel", masks=None, drawn=None):
    """melDisc.forward (discriminator.py:24-42) / linDisc.forward (:62-80): (B, F, T) -> (B, 1, 1)."""
    drop = _Drop(masks, drawn)
    p1, p2 = POOLS[kind]
    x = drop(_ln(_conv(x, sd, "conv1"), sd, "ln1"))                                  # :24-26
    # highwayConv with dropout, TTSModel_dropout.py:75-84 (k = 3, dilation 1, "same" padding 1)
    h = _conv(x, sd, "hc.conv", padding=1)
    d = h.shape[1] // 2
    h1, h2 = _ln(h[:, :d], sd, "hc.ln1"), _ln(h[:, d:], sd, "hc.ln2")
    g = torch.sigmoid(h1)
    x = drop(g * h2 + (1 - g) * x)
    x = _ln(F.avg_pool1d(_conv(x, sd, "conv2"), p1), sd, "ln2")                      # :28-30
    x = drop(F.leaky_relu(x, 0.05))                                                  # :31
    x = _ln(F.avg_pool1d(_conv(x, sd, "conv3"), p2), sd, "ln3")                      # :32-34
    x = _ln(_conv(F.leaky_relu(x, 0.05), sd, "conv4"), sd, "ln4")                    # :35-36
    x = _conv(F.leaky_relu(x, 0.05), sd, "conv5")                                    # :37
    return F.adaptive_avg_pool1d(x, 1)                                               # :38


def critic_losses(pred, gt, coeff, sd, kind="mel", lam=10.0, masks=None, drawn=None):
    """The critic iteration's two losses, train/adversarial_wasserstein_gp.py:300-313: gradient penalty on the
    interpolate (``coeff``: (B,) mixing coefficients) and the Wasserstein term mean(D(pred) - D(gt)).  Three critic calls
    in the reference's order (interpolate :303, ground truth :313, prediction :314); ``masks``, when a list, holds their
    3 x 3 masks in that order.  Returns (loss_gp, loss_D); both are differentiable w.r.t. ``sd``."""
    take = (lambda: [masks.pop(0) for _ in range(3)]) if isinstance(masks, list) else (lambda: masks)
    masks = list(masks) if isinstance(masks, list) else masks
    c = coeff.view(-1, 1, 1)
    mid = (c * gt.detach() + (1 - c) * pred.detach()).requires_grad_(True)
    out = critic(mid, sd, kind, take(), drawn)
    grads = torch.autograd.grad(outputs=out, inputs=mid, grad_outputs=torch.ones_like(out), retain_graph=True, create_graph=True)[0]
    loss_gp = torch.mean(lam * (torch.norm(grads, p=2, dim=(1, 2)) - 1) ** 2)
    d_gt = critic(gt.detach(), sd, kind, take(), drawn)
    d_pred = critic(pred.detach(), sd, kind, take(), drawn)
    return loss_gp, torch.mean(d_pred - d_gt)
