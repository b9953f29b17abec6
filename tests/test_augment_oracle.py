"""CPU tier: the input-transform oracle (oracle/augment_oracle.py) against vectors made by Pillow itself
(tests/golden/augment.npz, make_golden.py:g_augment), and against the installed Pillow where there is one."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import augment_oracle as A


def cases():
    g = load_golden("augment")
    for n in range(int(g["n_cases"])):
        S, fh, fv, i, j, h, w = (int(v) for v in g[f"c{n}:params"])
        yield n, g[f"c{n}:img"], S, (bool(fh), bool(fv), i, j, h, w), g[f"c{n}:out"]


def test_oracle_reproduces_pillow_vectors():
    for n, img, S, params, want in cases():
        got = A.train_transform(img, S, params)
        assert torch.equal(got, A.to_tensor(want)), n


def test_oracle_against_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for t in range(25):
        h, w = int(rng.integers(8, 300)), int(rng.integers(8, 300))
        oh, ow = int(rng.choice([16, 64, 96, h])), int(rng.choice([16, 64, 96, w]))
        a = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        want = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(A.resize_bilinear_u8(a, oh, ow), want), (h, w, oh, ow)


def test_crop_params_are_valid_boxes_with_the_reference_scale():
    g = torch.Generator().manual_seed(3)
    areas = []
    for _ in range(300):
        fh, fv, i, j, h, w = A.draw_train_params(512, 768, generator=g)
        assert 0 <= i and i + h <= 512 and 0 <= j and j + w <= 768 and h > 0 and w > 0
        areas.append(h * w / (512 * 768))
    assert 0.79 <= min(areas) and max(areas) <= 1.0 + 1e-9          # scale=(0.8, 1.0), data.py:110
    # a very elongated image exhausts the 10 draws and takes the central fallback with the clamped ratio
    i, j, h, w = A.random_resized_crop_params(100, 1000, generator=g)
    assert (h, w) == (100, 133) and j == (1000 - 133) // 2 and i == 0


def test_identity_box_is_to_tensor():
    a = np.random.default_rng(1).integers(0, 256, (40, 56, 3)).astype(np.uint8)
    assert np.array_equal(A.resize_bilinear_u8(a, 40, 56), a)
