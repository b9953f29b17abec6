import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def as_t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ---- shared comparison helpers ---------------------------------------------------------------
from vae_gan_amd.testing import digest  # noqa: E402


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64).flatten()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64).flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def close_l2(mine, truth, noise=0.0, rtol=1e-4, what="", k=4.0):
    """rel-L2(mine, truth) <= max(rtol, k*noise): 'truth' is the reference run in float64,
    'noise' the reference's own fp32 distance from it (LeakyReLU sign flips under rounding
    make some deep-net gradients irreproducible below that floor, see make_golden.py)."""
    if torch.is_tensor(mine):
        mine = mine.detach().cpu()
    e = rel_l2(mine, truth)
    tol = max(rtol, k * float(noise))
    assert e <= tol, f"{what}: rel-L2 {e:.3e} > tol {tol:.3e} (ref fp32 noise {float(noise):.1e})"
    return e


def check_digests(g, grads, rtol=2e-4, k=4.0, name_map=lambda n: n):
    """Compare gradient digests for every 'gd:' entry of golden dict ``g`` against ``grads``
    (name -> tensor).  l2 norms within tol; the dot with a N(0,1) probe differs by a draw of
    std ||delta||, so |ddot| <= 4*tol*l2.  Conv biases feeding a BatchNorm have a
    mathematically-zero gradient (fp noise on every implementation): for those
    only require 'tiny relative to the sibling weight gradient'."""
    n = 0
    # Module-level reproducibility floor.  A LeakyReLU whose pre-activation is within rounding of 0 flips sign with
    # the summation order; one flip moves every gradient upstream of it by 1e-3..1e-2 (rel-L2), on the reference
    # itself (noise:dx, noise:gd:* = its own fp32 run vs float64) and between two of our own kernel generations
    # (tests/gpu_hier_diff.py).  One parameter's recorded noise is a single sample of that event, so the floor of a
    # module is the largest noise the reference showed anywhere in it (flip-free modules keep rtol).
    ns = [float(v) for kk, v in g.items() if kk.startswith(("noise:gd:", "noise:dx")) and float(v) < 0.1]
    floor = max(ns) if ns else 0.0
    for key, ref in g.items():
        if not key.startswith("gd:"):
            continue
        name = key[3:]
        t = grads[name_map(name)]
        assert t is not None, name
        mine = digest(name, t)
        l2 = max(ref[0], 1e-30)
        sib = g.get("gd:" + name[:-len("bias")] + "weight") if name.endswith(".bias") else None
        if sib is not None and ref[0] < 1e-4 * sib[0]:
            assert mine[0] < 1e-3 * sib[0], f"{name}: expected ~0 grad, l2 {mine[0]:.3e} (weight l2 {sib[0]:.3e})"
            continue
        tol = max(rtol, k * max(float(g.get("noise:" + key, 0.0)), floor))
        assert abs(mine[0] - ref[0]) <= tol * l2, f"{name}: l2 {mine[0]:.6e} vs {ref[0]:.6e} (tol {tol:.1e})"
        assert abs(mine[2] - ref[2]) <= 4 * tol * l2, f"{name}: probe {mine[2]:.6e} vs {ref[2]:.6e} (tol {tol:.1e})"
        n += 1
    assert n > 0
    return n
