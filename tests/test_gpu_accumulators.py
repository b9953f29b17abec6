"""GPU tier: the per-channel statistics accumulators (csrc/common.h) in both modes - f64 atomics and the reproducible
fixed-point bins: a non-finite input must show up as non-finite statistics (the fixed-point path used
to add a NaN as 0), the collapse / load pair of the data-parallel statistics exchange is the identity on what the
finalize kernels read, and a launch whose block count could overflow a fixed-point slot is refused."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def vg():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import lib as L, ops
    was = ops.deterministic()
    yield L, ops
    ops.set_deterministic(was)


def _stats(L, ops, x, c):
    cs = x.shape[-1]
    rows = x.numel() // cs
    npad = ops._npad(c)
    stats = torch.zeros(L.STATS_REPLICAS * 2 * npad, dtype=torch.float64, device="cuda")
    L.check(L.load().vg_channel_stats(x.data_ptr(), stats.data_ptr(), rows, cs, c, npad, L.vg_dtype(x.dtype), L.stream()), "channel_stats")
    return stats, npad


def _read(L, stats, nslots):
    vec = torch.empty(nslots, dtype=torch.float64, device="cuda")
    L.check(L.load().vg_stats_collapse(stats.data_ptr(), nslots, vec.data_ptr(), L.stream()), "stats_collapse")
    torch.cuda.synchronize()
    return vec


@pytest.mark.parametrize("det", [False, True])
def test_nan_and_inf_inputs_give_non_finite_statistics(vg, det):
    L, ops = vg
    ops.set_deterministic(det)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(4096, 64, device="cuda", generator=g).to(torch.bfloat16)
    clean, npad = _stats(L, ops, x, 64)
    v = _read(L, clean, 2 * npad)
    assert torch.isfinite(v).all()
    assert torch.allclose(v[:64], x.double().sum(0), rtol=1e-6, atol=1e-6)
    assert torch.allclose(v[npad:npad + 64], (x.double() ** 2).sum(0), rtol=1e-6)
    x[17, 5] = float("nan")
    x[99, 9] = float("inf")
    bad, _ = _stats(L, ops, x, 64)
    v = _read(L, bad, 2 * npad)
    assert not torch.isfinite(v[5]) and not torch.isfinite(v[npad + 5]), "NaN input, finite statistics"
    assert not torch.isfinite(v[9]) and not torch.isfinite(v[npad + 9]), "Inf input, finite statistics"
    ok = torch.ones(2 * npad, dtype=torch.bool, device="cuda")
    ok[[5, 9, npad + 5, npad + 9]] = False
    assert torch.isfinite(v[ok]).all()                       # the other channels are untouched


@pytest.mark.parametrize("det", [False, True])
def test_collapse_then_load_keeps_what_the_finalize_kernels_read(vg, det):
    L, ops = vg
    ops.set_deterministic(det)
    g = torch.Generator(device="cuda").manual_seed(2)
    x = (torch.randn(20000, 128, device="cuda", generator=g) * 3 + 0.5).to(torch.bfloat16)
    stats, npad = _stats(L, ops, x, 128)
    before = _read(L, stats, 2 * npad)
    doubled = before * 2                                      # what a two-rank all-reduce of identical halves would give
    L.check(L.load().vg_stats_load(doubled.data_ptr(), 2 * npad, stats.data_ptr(), L.stream()), "stats_load")
    after = _read(L, stats, 2 * npad)
    assert torch.equal(after, doubled)                        # exact: a double is at most three f32 pieces of the bins
    # and the finalize kernel sees the same buffer: mean of the doubled sums over twice the rows = the original mean
    vec = torch.empty(4, 128, dtype=torch.float32, device="cuda")
    ones, zeros = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    L.check(L.load().vg_bn_finalize(stats.data_ptr(), 128, npad, 2 * 20000, ones.data_ptr(), zeros.data_ptr(), 1e-5, 0.1, None, None,
                                    None, 1, vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), L.stream()),
            "bn_finalize")
    torch.cuda.synchronize()
    assert torch.allclose(vec[0].double(), x.double().mean(0), rtol=1e-5, atol=1e-6)


def test_deterministic_mode_refuses_more_blocks_than_a_slot_can_hold(vg):
    """47-bit addends leave a fixed-point slot room for 2^16 adds (2^17 blocks with the two sub-replicas): a conv launch
    with fused statistics over more tiles than that is refused with a message instead of wrapping silently."""
    L, ops = vg
    lib = L.load()
    ops.set_deterministic(True)
    n, h, w = 2048, 128, 128                 # 2^25 output pixels: more than 2^17 tiles of 128 pixels
    d = L.ConvDesc(n=n, h=h, w=w, oh=h, ow=w, ci=32, co=32, xcs=32, ycs=32, r=3, s=3, stride=1, pad=1, dtype=L.VG_BF16)
    one = torch.zeros(64, device="cuda")     # never touched: the launch is refused before anything is enqueued
    rc = lib.vg_conv_fwd(C.byref(d), one.data_ptr(), one.data_ptr(), None, one.data_ptr(), one.data_ptr(), 0, 0, L.stream())
    assert rc != 0 and "fixed-point slots" in L.last_error(), L.last_error()
    ops.set_deterministic(False)
    d = L.ConvDesc(n=1, h=32, w=32, oh=32, ow=32, ci=32, co=32, xcs=32, ycs=32, r=3, s=3, stride=1, pad=1, dtype=L.VG_BF16)
    x = torch.zeros(1, 32, 32, 32, device="cuda", dtype=torch.bfloat16)
    wt = torch.zeros(lib.vg_conv_packed_bytes(C.byref(d), 0) // 2, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    stats = torch.zeros(L.STATS_REPLICAS * 2 * 32, dtype=torch.float64, device="cuda")
    L.check(lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), stats.data_ptr(), 0, 0, L.stream()), "conv_fwd")
    torch.cuda.synchronize()
