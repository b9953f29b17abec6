"""Deterministic, construction-order-independent state_dict fill.

Parity runs need *identical weights* on three sides: the reference modules (only in the
build container), the CPU oracle (``oracle/``) and the HIP engine.  Instead of storing
up to 688 MB of weights as fixtures, every side regenerates them from the state_dict
*key names* alone: ``fill_state_dict`` seeds one generator per key from crc32(key), so
the result does not depend on module construction order or on torch's default init.

Scales are chosen so that activations stay O(1) through ~30 layers (fan-in scaled
weights, BN affine near identity, positive running_var) and so that every learnable
tensor receives a non-trivial gradient (e.g. SelfAttention.gamma = 0.7, not 0 —
SURVEY.md §3.3).
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping

import torch


def _gen(key: str, salt: int = 0) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode("utf-8")) + 7919 * salt) & 0x7FFFFFFF)
    return g


def fill_tensor(key: str, shape, dtype=torch.float32, salt: int = 0) -> torch.Tensor:
    """Value for one state_dict entry, decided from its key suffix and shape."""
    shape = tuple(shape)
    g = _gen(key, salt)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == "running_var":
        return 0.75 + 0.5 * torch.rand(shape, generator=g)
    if leaf == "gamma":  # SelfAttention residual gate (reference inits to 0 → zero q/k/v grads)
        return torch.full(shape, 0.7)
    if leaf == "ema_cluster_size":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "ema_w" or key.endswith("embeddings.weight"):
        # codebook ~ N(0, 0.5); ema_w shares the codebook key so that ema_w == weight at t=0
        g = _gen(key.replace("ema_w", "embeddings.weight"), salt)
        return 0.5 * torch.randn(shape, generator=g)
    if leaf == "weight" and len(shape) == 1:  # BatchNorm affine scale
        return 0.75 + 0.5 * torch.rand(shape, generator=g)
    if leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == "weight" and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if ".upsample." in key or "ConvTranspose" in key:
            fan_in = shape[0] * shape[2] * shape[3] // 4 if len(shape) == 4 else fan_in
        std = (1.6 / max(fan_in, 1)) ** 0.5
        return std * torch.randn(shape, generator=g)
    return 0.1 * torch.randn(shape, generator=g)


def fill_state_dict(template: Mapping[str, torch.Tensor], salt: int = 0) -> Dict[str, torch.Tensor]:
    """New dict with the same keys/shapes/dtypes as ``template`` and deterministic values."""
    out = {}
    for k, v in template.items():
        t = fill_tensor(k, v.shape, salt=salt)
        out[k] = t.to(v.dtype) if v.dtype != torch.int64 else t
    return out


def synthetic_images(batch: int, size: int, seed: int = 42) -> torch.Tensor:
    """Uniform [0,1) RGB batch, matching ToTensor's range."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.rand(batch, 3, size, size, generator=g)


def synthetic_eps(shape, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn(*shape, generator=g)


def hier_step_eps(batch: int, size: int, iters: int = 2):
    """Injected N(0,1) draws of a hierarchical step: per iteration two forwards x three levels
    (level order 0,1,2, hierarchical_vae_gan/model.py:368)."""
    shapes = [(batch, 32, size // 2, size // 2), (batch, 64, size // 4, size // 4), (batch, 128, size // 8, size // 8)]
    return [[[synthetic_eps(s, 700 + 100 * it + 10 * f + lv) for lv, s in enumerate(shapes)] for f in range(2)]
            for it in range(iters)]


def cotangent(name: str, shape) -> torch.Tensor:
    """Deterministic N(0,1) tensor keyed by a name (projection vector for gradient checks)."""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return torch.randn(*shape, generator=g)


def digest(key: str, t: torch.Tensor):
    """[l2, sum, dot(random probe), first 8 values] of a tensor, as float64 numpy."""
    t = t.detach().double().flatten().cpu()
    probe = cotangent("probe:" + key, (t.numel(),)).double()
    head = torch.zeros(8, dtype=torch.float64)
    head[: min(8, t.numel())] = t[:8]
    return torch.cat([torch.stack([t.norm(), t.sum(), (t * probe).sum()]), head]).numpy()
