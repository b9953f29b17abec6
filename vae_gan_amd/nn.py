"""Layer modules with the reference's parameter names/shapes, executed by the HIP kernels.

Every module keeps torch's ``state_dict`` contract (``weight`` OIHW f32, ``bias``, BatchNorm
buffers) so checkpoints interchange with the reference.  ``FusedSequential``
has the same child indices as the reference's ``nn.Sequential`` stacks but runs
conv → (stats in the conv epilogue) → BatchNorm-finalize → normalise+LeakyReLU as fused kernels.
Internal tensors are NHWC (see ``ops``).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import lib as L
from . import ops


class Conv2d(nn.Module):
    """Parameter holder + launcher for nn.Conv2d(in, out, k, stride, padding) (square kernels)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self._packs = ops.PackedWeights()
        self.reset_parameters()

    def reset_parameters(self):
        # same distribution as torch's Conv2d default init
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = self.in_channels * self.kernel_size ** 2
        bound = 1 / math.sqrt(fan_in)
        nn.init.uniform_(self.bias, -bound, bound)

    def run(self, x, act=L.ACT_NONE, want_stats=False, link=None, reslink=None):
        return ops.Conv2dFn.apply(x, self.weight, self.bias, self.stride, self.padding, act, want_stats, self._packs, link,
                                  reslink, ops.claim_bwdstat(x))

    def forward(self, x):
        return self.run(x)[0]


class ConvTranspose2d(nn.Module):
    """nn.ConvTranspose2d(in, out, k, stride, padding); weight is [in, out, k, k] like torch."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self._packs = ops.PackedWeights()
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(out_channels * kernel_size ** 2)   # torch: fan_in computed on dim 1
        nn.init.uniform_(self.bias, -bound, bound)

    def run(self, x, act=L.ACT_NONE, want_stats=False, link=None):
        return ops.ConvTranspose2dFn.apply(x, self.weight, self.bias, self.stride, self.padding, act, want_stats, self._packs,
                                           link, ops.claim_bwdstat(x))

    def forward(self, x):
        return self.run(x)[0]


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1, affine, running stats)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def run(self, x, stats, act=L.ACT_NONE, res=None, link=None, reslink=None):
        """x: raw NHWC conv output; stats: from the conv epilogue (or None → computed here);
        link: ops.BiasLink of the conv that produced x (its bias gradient is taken by this layer's backward)."""
        c = self.num_features
        if not self.training:
            return ops.bn_eval_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, act, c, res)
        if stats is None:
            stats = ops.channel_stats(x.contiguous(), c)
        bsl = ops.BwdStatLink() if (torch.is_grad_enabled() and x.requires_grad) else None
        y = ops.BNActFn.apply(x, stats, self.weight, self.bias, res, self.running_mean, self.running_var,
                              self.num_batches_tracked, self.eps, self.momentum, act, c, link, reslink, bsl)
        if bsl is not None:
            y._vg_bwdstat = bsl         # used only once a container armed it (ops.BwdStatLink)
        return y

    def forward(self, x):
        return self.run(x, None)


class LeakyReLU(nn.Module):
    """Marker for LeakyReLU(0.2): always fused into the preceding conv / BatchNorm kernel."""

    def __init__(self, negative_slope=0.2, inplace=True):
        super().__init__()
        if abs(negative_slope - 0.2) > 1e-12:
            raise ValueError("the HIP kernels implement LeakyReLU(0.2) only")
        self.negative_slope = negative_slope


class Sigmoid(nn.Module):
    """Marker for the output Sigmoid: fused into the last conv's epilogue."""


class ResidualBlock(nn.Module):
    """beta_vae_gans/model.py:11-40 (identical in the other two model.py files).  Every call site of the reference
    uses in_channels == out_channels (Identity skip); with different widths the skip is the reference's 1x1 conv
    (model.py:24-25)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = Conv2d(in_channels, out_channels, 3, padding=1)
        self.conv2 = Conv2d(out_channels, out_channels, 3, padding=1)
        self.bn1 = BatchNorm2d(out_channels)
        self.bn2 = BatchNorm2d(out_channels)
        self.relu = LeakyReLU(0.2)
        self.skip = nn.Identity()
        if in_channels != out_channels:
            self.skip = Conv2d(in_channels, out_channels, 1)

    def forward(self, x):
        train = self.training
        l1 = ops.BiasLink(self.conv1.bias) if train else None
        l2 = ops.BiasLink(self.conv2.bias) if train else None
        if isinstance(self.skip, Conv2d):
            # x fans out to conv1 and the skip conv: their two input gradients are added by the HIP add kernel
            xa, xb = ops.fan_out(x)
            h, st = self.conv1.run(xa, want_stats=train, link=l1)
            h = self.bn1.run(h, st, L.ACT_LRELU, link=l1)
            ops.arm_bwdstat(h)
            h, st = self.conv2.run(h, want_stats=train, link=l2)
            return self.bn2.run(h, st, L.ACT_LRELU, res=self.skip(xb), link=l2)
        rl = ops.ResLink() if (train and x.requires_grad) else None
        h, st = self.conv1.run(x, want_stats=train, link=l1, reslink=rl)
        h = self.bn1.run(h, st, L.ACT_LRELU, link=l1)
        ops.arm_bwdstat(h)              # bn1 feeds conv2 and nothing else
        h, st = self.conv2.run(h, want_stats=train, link=l2)
        return self.bn2.run(h, st, L.ACT_LRELU, res=x, link=l2, reslink=rl)


class FusedSequential(nn.Sequential):
    """nn.Sequential with the reference's child indices; forward fuses
    [conv|convT] → [BatchNorm] → [LeakyReLU|Sigmoid] runs into single kernel chains."""

    def forward(self, x):
        mods = list(self)
        i, n = 0, len(mods)
        while i < n:
            if i > 0:
                ops.arm_bwdstat(x)      # the output of one element feeds the next element and nothing else
            m = mods[i]
            if isinstance(m, (Conv2d, ConvTranspose2d)):
                j = i + 1
                bn = None
                if j < n and isinstance(mods[j], BatchNorm2d):
                    bn = mods[j]
                    j += 1
                act = L.ACT_NONE
                if j < n and isinstance(mods[j], LeakyReLU):
                    act, j = L.ACT_LRELU, j + 1
                elif j < n and isinstance(mods[j], Sigmoid):
                    act, j = L.ACT_SIGMOID, j + 1
                if bn is None:
                    x = m.run(x, act=act)[0]
                else:
                    link = ops.BiasLink(m.bias) if bn.training else None
                    raw, st = m.run(x, want_stats=bn.training, link=link)
                    x = bn.run(raw, st, act, link=link)
                i = j
            elif isinstance(m, BatchNorm2d):
                j = i + 1
                act = L.ACT_NONE
                if j < n and isinstance(mods[j], LeakyReLU):
                    act, j = L.ACT_LRELU, j + 1
                x = m.run(x, None, act)
                i = j
            elif isinstance(m, (LeakyReLU, Sigmoid)):
                raise RuntimeError("stand-alone activation without a producer conv/BN is not supported")
            else:
                x = m(x)
                i += 1
        return x
