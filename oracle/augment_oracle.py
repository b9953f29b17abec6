"""CPU oracle of the training input transform.  TEST INFRASTRUCTURE ONLY.

The reference's train transform (beta_vae_gans/data.py:106-112) is
    RandomHorizontalFlip -> RandomVerticalFlip -> RandomResizedCrop(image_size, scale=(0.8, 1.0)) -> ToTensor
and its eval transform (:116-119) Resize((S, S)) -> ToTensor, all on PIL images.  The arithmetic lives in two
third-party libraries that are not part of /root/reference:
  * torchvision.transforms (requirements.txt: `torchvision>=0.10.0`, NOT installed here): RandomResizedCrop.get_params,
    the flips and ToTensor are restated from its published algorithm (transforms.py / functional.py);
    parameter-draw parity with torchvision's RNG consumption is therefore "parity unpinned";
  * Pillow (requirements.txt: `pillow>=8.0.0`; 12.2.0 is installed in the build container): Image.resize(...,
    BILINEAR) = ImagingResample (src/libImaging/Resample.c), restated below: double-precision triangle-filter
    coefficients with antialiasing support, normalised, rounded to 22-bit fixed point; horizontal pass, rounding to
    uint8, then vertical pass.  This part IS pinned: tests/golden/augment.npz holds outputs of Pillow itself
    (make_golden.py:g_augment) and tests/test_augment_oracle.py also compares with the installed Pillow when present.
"""
from __future__ import annotations

import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


def _coeffs(in_size: int, in0: float, in1: float, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter."""
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(xmax)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Image.resize((out_w, out_h), BILINEAR) of an HxWxC uint8 image (ImagingResample, 8 bits per channel)."""
    h, w, c = img.shape
    bh, kh = _coeffs(w, 0.0, float(w), out_w)
    bv, kv = _coeffs(h, 0.0, float(h), out_h)
    src = img.astype(np.int64)
    if out_w != w:
        first, last = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
        tmp = np.zeros((last - first, out_w, c), dtype=np.uint8)
        for xx in range(out_w):
            x0, n = bh[xx]
            acc = (src[first:last, x0:x0 + n, :] * kh[xx, :n, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        bv = bv.copy()
        bv[:, 0] -= first
        src = tmp.astype(np.int64)
    if out_h != h:
        out = np.zeros((out_h, src.shape[1], c), dtype=np.uint8)
        for yy in range(out_h):
            y0, n = bv[yy]
            acc = (src[y0:y0 + n] * kv[yy, :n, None, None]).sum(axis=0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        return out
    return src.astype(np.uint8)


def to_tensor(img: np.ndarray) -> torch.Tensor:
    """torchvision ToTensor on a uint8 HWC image: CHW float32 = uint8 / 255 (float32 division)."""
    return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).to(torch.float32).div(255)


def random_resized_crop_params(height: int, width: int, scale=(0.8, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """torchvision RandomResizedCrop.get_params: up to 10 draws of (area, log-uniform aspect), then the central
    fallback.  Returns (i, j, h, w)."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1], generator=generator).item()
        aspect = math.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=generator).item())
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,), generator=generator).item()
            j = torch.randint(0, width - w + 1, size=(1,), generator=generator).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def draw_train_params(height: int, width: int, generator=None):
    """The three random transforms of data.py:108-110 in Compose order: (flip_h, flip_v, i, j, h, w)."""
    flip_h = bool(torch.rand(1, generator=generator) < 0.5)
    flip_v = bool(torch.rand(1, generator=generator) < 0.5)
    return (flip_h, flip_v) + tuple(random_resized_crop_params(height, width, generator=generator))


def train_transform(img: np.ndarray, size: int, params) -> torch.Tensor:
    """data.py:107-112 with the random draws given: flips, crop, Pillow-bilinear resize to size x size, ToTensor."""
    flip_h, flip_v, i, j, h, w = params
    if flip_h:
        img = img[:, ::-1]
    if flip_v:
        img = img[::-1]
    return to_tensor(resize_bilinear_u8(np.ascontiguousarray(img[i:i + h, j:j + w]), size, size))


def eval_transform(img: np.ndarray, size: int) -> torch.Tensor:
    """data.py:116-119: Resize((S, S)) + ToTensor."""
    return to_tensor(resize_bilinear_u8(img, size, size))
