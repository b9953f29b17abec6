"""Shared layer-stack builders (child order == the reference's nn.Sequential order, so that
state_dict keys match: SURVEY.md §8(b))."""
from __future__ import annotations

from typing import List

from . import nn as vnn


def down_stack(cin: int, ch: List[int]) -> list:
    """conv3+BN+LReLU, then per level [Res, conv4s2p1, BN, LReLU], then Res
    (beta_vae_gans/model.py:49-64, vq_vae_gan/model.py:143-158)."""
    layers = [vnn.Conv2d(cin, ch[0], 3, padding=1), vnn.BatchNorm2d(ch[0]), vnn.LeakyReLU(0.2)]
    for a, b in zip(ch[:-1], ch[1:]):
        layers += [vnn.ResidualBlock(a, a), vnn.Conv2d(a, b, 4, stride=2, padding=1), vnn.BatchNorm2d(b), vnn.LeakyReLU(0.2)]
    layers.append(vnn.ResidualBlock(ch[-1], ch[-1]))
    return layers


def up_stack(ch: List[int], cout: int) -> list:
    """per level [Res, ConvT4s2p1, BN, LReLU], then Res, conv3, Sigmoid
    (beta_vae_gans/model.py:121-131, vq_vae_gan/model.py:184-194)."""
    layers = []
    for a, b in zip(ch[:-1], ch[1:]):
        layers += [vnn.ResidualBlock(a, a), vnn.ConvTranspose2d(a, b, 4, stride=2, padding=1), vnn.BatchNorm2d(b),
                   vnn.LeakyReLU(0.2)]
    layers += [vnn.ResidualBlock(ch[-1], ch[-1]), vnn.Conv2d(ch[-1], cout, 3, padding=1), vnn.Sigmoid()]
    return layers


def patch_disc_stack(cin: int, ch: List[int]) -> list:
    """conv4s2+LReLU, [conv4s2+BN+LReLU]*, conv4s1p1 -> 1 channel
    (beta_vae_gans/model.py:156-171, vq_vae_gan/model.py:211-221)."""
    layers = [vnn.Conv2d(cin, ch[0], 4, stride=2, padding=1), vnn.LeakyReLU(0.2)]
    for a, b in zip(ch[:-1], ch[1:]):
        layers += [vnn.Conv2d(a, b, 4, stride=2, padding=1), vnn.BatchNorm2d(b), vnn.LeakyReLU(0.2)]
    layers.append(vnn.Conv2d(ch[-1], 1, 4, stride=1, padding=1))
    return layers
