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


# ---- whole-step goldens (vq_step_64 / beta_step_256 / hier_step_128) --------------------------
def check_step(g, it, r, keys):
    """Scalars + reconstruction of iteration ``it`` of a whole-step golden (r: name -> float / tensor)."""
    # reference's own fp32 noise floor; from the second iteration on every scalar also inherits the
    # perturbation of the whole reconstruction by Adam's lr*sign(g) on rounding-noise gradients
    # (noise:it1_recon_img), of which one fp32-vs-fp64 pair per scalar is only a point sample
    floor = 4 * float(g[f"noise:it{it}_recon_img"]) if it > 0 else 0.0
    for k in keys:
        ref = float(g[f"it{it}_{k}"])
        tol = max(1e-4, 4 * float(g[f"noise:it{it}_{k}"]), floor)
        # adv = -mean(D(fake)) is a mean of signed O(1) logits (cancellation): its scale is that of d_loss
        scale = max(abs(ref), 1e-3, abs(float(g[f"it{it}_d_loss"])) if k == "adv" else 0.0)
        if k == "adv" and it > 0:
            tol = max(1e-4, floor)
        assert abs(r[k] - ref) <= tol * scale, (it, k, r[k], ref, tol)
    close_l2(r["recon_img"], g[f"it{it}_recon_img"], g[f"noise:it{it}_recon_img"], what="recon")


def check_state(g, sd, last_noise):
    """BatchNorm counters exactly, every other state_dict entry by its digest: the l2 norm AND the dot with a random
    probe (the norm alone cannot see a per-element error of the Adam update).  Two Adam steps move a parameter by
    <= 2*lr per element whatever the gradient, so parameters are compared at tol * ||p||; entries that start at a
    constant (fresh running statistics) share the scale."""
    tol = max(1e-4, 4 * last_noise)
    for key, ref in g.items():
        if key.startswith("sd:") and np.asarray(ref).dtype.kind in "iu":
            assert int(sd[key[3:]]) == int(ref), key
        elif key.startswith("sd:"):       # buffers stored whole (the quantiser's EMA state and codebook)
            close_l2(sd[key[3:]], ref, rtol=tol, what=key[3:])
        elif key.startswith("sdd:"):
            k = key[4:]
            d = digest(k, sd[k])
            assert abs(d[0] - ref[0]) <= tol * max(ref[0], 1e-6), k
            assert abs(d[2] - ref[2]) <= 4 * tol * max(ref[0], 1e-6), (k, "probe", d[2], ref[2])
