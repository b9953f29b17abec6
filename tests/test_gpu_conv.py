"""GPU tier: the conv primitives through the raw C ABI against torch CPU float64 convolutions (forward, input
gradient, weight gradient, fused BN statistics, zero padding lanes), and the three kernel generations
(im2col-row, spatial-tile, ping-pong) against each other on the same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import gpu_ops_check as G  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-3)], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", G.CASES, ids=lambda c: "n%d_%dx%d_ci%d_co%d_k%ds%dp%d" % c)
def test_conv_primitives(gpu, case, dtype, tol):
    r = G.check_conv(*case, dtype, verbose=False)
    for k in ("fwd", "dgrad", "stat_sum", "stat_sq"):
        assert r[k] <= tol, (k, r[k])
    # f32 atomics over pixel splits; bf16 inputs are rounded before both sides, so the bound is the same order
    assert r["wgrad"] <= (tol if dtype == torch.float32 else 5e-3), r["wgrad"]
    assert r["pad0"] == 0.0


def test_kernel_generations_agree(gpu):
    import gpu_conv_xcheck as X
    assert X.main(exit=False) == 0


@pytest.mark.parametrize("case", [(8, 64, 64, 64, 64, 3, 1, 1), (4, 64, 64, 128, 256, 4, 2, 1), (2, 40, 36, 64, 64, 3, 1, 1)],
                         ids=lambda c: "n%d_%dx%d_ci%d_co%d_k%ds%dp%d" % c)
def test_spatial_tile_weight_gradient_is_bit_reproducible(gpu, case):
    """The spatial-tile weight-gradient kernels write one partial image per tile group and sum them in a fixed order
    (no float atomics): two launches on the same inputs give the same bits, and accumulate=1 adds exactly once."""
    import ctypes as C
    from vae_gan_amd import lib as L
    n, h, w, ci, co, k, stride, pad = case
    lib = L.load()
    d = G.make_desc(n, h, w, ci, co, k, stride, pad, torch.bfloat16)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(n, h, w, d.xcs, device="cuda", generator=g).bfloat16()
    gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda", generator=g).bfloat16()
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(2):
        dw = torch.full((co, ci, k, k), float("nan"), dtype=torch.float32, device="cuda")
        ws.fill_(float("nan"))          # nothing may depend on what the workspace held
        L.check(lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream()), "wgrad")
        outs.append(dw)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    acc = outs[0].clone()
    L.check(lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), acc.data_ptr(), ws.data_ptr(), 1, L.stream()), "wgrad")
    torch.cuda.synchronize()
    assert torch.equal(acc, outs[0] * 2)


def _bn_vec(c, cpad_unused, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    mean = 0.3 * torch.randn(c, device="cuda", generator=g)
    invstd = 0.5 + torch.rand(c, device="cuda", generator=g)
    scale = invstd * (0.75 + 0.5 * torch.rand(c, device="cuda", generator=g))
    shift = 0.2 * torch.randn(c, device="cuda", generator=g) - mean * scale
    return mean, invstd, scale, shift


# (n, h, w, ci, co, k, stride, pad, accumulate, mask from y): ci = channels of the BatchNorm whose gradient dx is
@pytest.mark.parametrize("case", [(4, 64, 64, 64, 64, 3, 1, 1, 0, False), (4, 64, 64, 64, 64, 3, 1, 1, 1, True),
                                  (2, 32, 64, 128, 128, 3, 1, 1, 1, False), (2, 32, 32, 256, 512, 3, 1, 1, 0, True),
                                  (2, 64, 64, 64, 128, 4, 2, 1, 0, True), (2, 40, 36, 64, 64, 3, 1, 1, 0, False),
                                  (2, 64, 64, 64, 3, 3, 1, 1, 0, False)],
                         ids=lambda c: "n%d_%dx%d_ci%d_co%d_k%ds%dp%d_acc%d_y%d" % c)
def test_dgrad_bnstats_matches_separate_reduce(gpu, case):
    """vg_conv_dgrad_bnstats: the input-gradient kernel takes the BatchNorm backward sums of the tile it writes.  Same
    dx bits as vg_conv_dgrad, and the sums equal vg_bn_bwd_reduce run on that dx (full and ragged tiles, accumulate
    onto a parked skip gradient, stride-2 phases, mask recomputed from x or read from y, thin 3-channel consumer)."""
    import ctypes as C
    from vae_gan_amd import lib as L
    n, h, w, ci, co, k, stride, pad, acc, from_y = case
    lib = L.load()
    dt = torch.bfloat16
    d = G.make_desc(n, h, w, ci, co, k, stride, pad, dt)
    g = torch.Generator(device="cuda").manual_seed(11)
    wt = (torch.randn(co, ci, k, k, device="cuda", generator=g) / (ci * k * k) ** 0.5).contiguous()
    wpt = G.pack(lib, d, wt, 1, dt)
    gy = torch.zeros(n, d.oh, d.ow, d.ycs, device="cuda", dtype=dt)
    gy[..., :co] = torch.randn(n, d.oh, d.ow, co, device="cuda", generator=g).to(dt)
    xraw = torch.zeros(n, h, w, d.xcs, device="cuda", dtype=dt)             # the BatchNorm's raw input
    xraw[..., :ci] = torch.randn(n, h, w, ci, device="cuda", generator=g).to(dt)
    ybn = torch.zeros_like(xraw)                                            # its output (sign = activation mask)
    ybn[..., :ci] = torch.randn(n, h, w, ci, device="cuda", generator=g).to(dt)
    parked = torch.zeros_like(xraw)
    parked[..., :ci] = torch.randn(n, h, w, ci, device="cuda", generator=g).to(dt)
    mean, invstd, scale, shift = _bn_vec(ci, 0, 5)
    cpad = 32 if ci <= 32 else (64 if ci <= 64 else (ci + 127) // 128 * 128)
    # reference: plain dgrad, then the separate reduction pass over (dx, x)
    dx0 = parked.clone()
    L.check(lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx0.data_ptr(), None, L.ACT_NONE, acc, L.stream()), "dgrad")
    want = torch.zeros(L.STATS_REPLICAS * 2 * cpad, dtype=torch.float64, device="cuda")
    L.check(lib.vg_bn_bwd_reduce(dx0.data_ptr(), ybn.data_ptr() if from_y else None, xraw.data_ptr(), mean.data_ptr(),
                                 invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), want.data_ptr(), n * h * w, d.xcs, ci,
                                 cpad, L.ACT_LRELU, L.VG_BF16, L.stream()), "bn_bwd_reduce")
    # fused
    dx1 = parked.clone()
    got = torch.zeros_like(want)
    bs = L.BnBwdStats(x=xraw.data_ptr(), y=ybn.data_ptr() if from_y else None, mean=mean.data_ptr(), invstd=invstd.data_ptr(),
                      scale=scale.data_ptr(), shift=shift.data_ptr(), sums=got.data_ptr(), cpad=cpad, act=L.ACT_LRELU)
    fused = C.c_int(-1)
    L.check(lib.vg_conv_dgrad_bnstats(C.byref(d), gy.data_ptr(), wpt.data_ptr(), dx1.data_ptr(), acc, C.byref(bs),
                                      C.byref(fused), L.stream()), "dgrad_bnstats")
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1)
    if fused.value == 0:      # the weights-stationary kernel (64 -> 64, whole tiles, no accumulate) has no fused sums
        assert lib.vg_last_conv_kernel().startswith(b"igemm4") and float(got.abs().max()) == 0.0, lib.vg_last_conv_kernel()
        return
    assert fused.value == 1, lib.vg_last_conv_kernel()
    a = want.view(L.STATS_REPLICAS, 2, cpad).sum(0)
    b = got.view(L.STATS_REPLICAS, 2, cpad).sum(0)
    scale_ = a.abs().max(dim=1, keepdim=True).values
    assert float(((a - b).abs() / scale_).max()) <= 2e-6, float(((a - b).abs() / scale_).max())
    assert float(b[:, ci:].abs().max()) == 0.0 if cpad > ci else True


@pytest.mark.parametrize("case", [(4, 64, 64, 128, 128, 3, 1, 1, 0), (2, 64, 64, 256, 256, 3, 1, 1, 0), (2, 32, 32, 256, 512, 3, 1, 1, 1), (2, 64, 64, 64, 128, 4, 2, 1, 0),
                                  (2, 40, 36, 128, 64, 3, 1, 1, 1)],
                         ids=lambda c: "n%d_%dx%d_ci%d_co%d_k%ds%dp%d_acc%d" % c)
def test_dgrad_bnstats_mask_out_stores_masked_gradient(gpu, case):
    """vg_bn_bwd_stats.mask_out: the fused launch stores dz = dx * LeakyReLU'(y) instead of dx.  Bits equal to
    vg_bn_bwd_reduce_dz run on the plain vg_conv_dgrad output, and so are the sums (those of the stored values); the
    BatchNorm backward that follows (vg_bn_bwd_apply on dz, act NONE, no y) matches the three-tensor apply pass."""
    import ctypes as C
    from vae_gan_amd import lib as L
    n, h, w, ci, co, k, stride, pad, acc = case
    lib = L.load()
    dt = torch.bfloat16
    d = G.make_desc(n, h, w, ci, co, k, stride, pad, dt)
    g = torch.Generator(device="cuda").manual_seed(12)
    wt = (torch.randn(co, ci, k, k, device="cuda", generator=g) / (ci * k * k) ** 0.5).contiguous()
    wpt = G.pack(lib, d, wt, 1, dt)
    gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda", generator=g).to(dt)
    xraw = torch.randn(n, h, w, d.xcs, device="cuda", generator=g).to(dt)
    ybn = torch.randn(n, h, w, d.xcs, device="cuda", generator=g).to(dt)
    parked = torch.randn(n, h, w, d.xcs, device="cuda", generator=g).to(dt)
    mean, invstd, scale, shift = _bn_vec(ci, 0, 5)
    gamma = torch.rand(ci, device="cuda", generator=g) + 0.5
    cpad, rows = ci, n * h * w
    nsum = L.STATS_REPLICAS * 2 * cpad
    dx0 = parked.clone()
    L.check(lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx0.data_ptr(), None, L.ACT_NONE, acc, L.stream()), "dgrad")
    want = torch.zeros(nsum + 2 * cpad, dtype=torch.float64, device="cuda")
    dz0 = torch.empty_like(dx0)
    L.check(lib.vg_bn_bwd_reduce_dz(dx0.data_ptr(), ybn.data_ptr(), xraw.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                    scale.data_ptr(), shift.data_ptr(), want.data_ptr(), dz0.data_ptr(), rows, d.xcs, ci, cpad,
                                    L.ACT_LRELU, L.VG_BF16, L.stream()), "bn_bwd_reduce_dz")
    ref = torch.where(ybn.float() > 0, dx0.float(), dx0.float() * 0.2).to(dt)
    assert torch.equal(dz0, ref)
    dx1 = parked.clone()
    got = torch.zeros_like(want)
    bs = L.BnBwdStats(x=xraw.data_ptr(), y=ybn.data_ptr(), mean=mean.data_ptr(), invstd=invstd.data_ptr(), scale=scale.data_ptr(),
                      shift=shift.data_ptr(), sums=got.data_ptr(), cpad=cpad, act=L.ACT_LRELU, mask_out=1)
    fused = C.c_int(-1)
    L.check(lib.vg_conv_dgrad_bnstats(C.byref(d), gy.data_ptr(), wpt.data_ptr(), dx1.data_ptr(), acc, C.byref(bs),
                                      C.byref(fused), L.stream()), "dgrad_bnstats")
    torch.cuda.synchronize()
    if fused.value == 0:      # weights-stationary kernels: no fused sums, and the gradient is stored unmasked
        assert lib.vg_last_conv_kernel().startswith(b"igemm4") and torch.equal(dx1, dx0), lib.vg_last_conv_kernel()
        return
    assert fused.value == 1, lib.vg_last_conv_kernel()
    assert torch.equal(dx1, dz0)
    a = want[:nsum].view(L.STATS_REPLICAS, 2, cpad).sum(0)
    b = got[:nsum].view(L.STATS_REPLICAS, 2, cpad).sum(0)
    scale_ = a.abs().max(dim=1, keepdim=True).values
    assert float(((a - b).abs() / scale_).max()) <= 2e-6
    # the apply pass on the masked gradient against the three-tensor pass on the raw one (same sums)
    dxa, dxb, dres = torch.empty_like(dx0), torch.empty_like(dx0), torch.empty_like(dx0)
    dg = torch.zeros(4, ci, device="cuda")
    L.check(lib.vg_bn_bwd_apply(dx0.data_ptr(), ybn.data_ptr(), xraw.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                shift.data_ptr(), gamma.data_ptr(), want.data_ptr(), dxa.data_ptr(), dres.data_ptr(), dg[0].data_ptr(),
                                dg[1].data_ptr(), 0, None, rows, rows, 1.0, d.xcs, ci, cpad, L.ACT_LRELU, L.VG_BF16, L.stream()), "apply")
    L.check(lib.vg_bn_bwd_apply(dx1.data_ptr(), None, xraw.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                shift.data_ptr(), gamma.data_ptr(), got.data_ptr(), dxb.data_ptr(), None, dg[2].data_ptr(),
                                dg[3].data_ptr(), 0, None, rows, rows, 1.0, d.xcs, ci, cpad, L.ACT_NONE, L.VG_BF16, L.stream()), "apply")
    torch.cuda.synchronize()
    assert torch.equal(dres, dz0)                    # the skip gradient of the old pass is the masked tensor
    rel = float((dxa.float() - dxb.float()).norm() / dxa.float().norm())
    assert rel <= 4e-3, rel                          # one more bf16 rounding of 0.2 * dout on the masked elements
    assert float((dg[0] - dg[2]).abs().max() / dg[0].abs().max()) <= 1e-5 and float((dg[1] - dg[3]).abs().max() / dg[1].abs().max()) <= 1e-5


# (n, h, w, ci, co, k, stride, pad, mask_out): the launch is a FORWARD gather (the input gradient of a ConvTranspose2d
# with the desc's ci = the transposed conv's output channels); co = channels of the BatchNorm whose gradient it writes
@pytest.mark.parametrize("case", [(2, 64, 64, 64, 128, 4, 2, 1, 1), (2, 64, 64, 128, 256, 4, 2, 1, 1), (2, 64, 64, 128, 256, 4, 2, 1, 0),
                                  (2, 32, 32, 256, 512, 4, 2, 1, 1), (3, 50, 54, 64, 128, 4, 2, 1, 0)],
                         ids=lambda c: "n%d_%dx%d_ci%d_co%d_k%ds%dp%d_mask%d" % c)
def test_fwd_bnstats_transposed_conv_input_gradient(gpu, case):
    """vg_conv_fwd_bnstats: the input gradient of a stride-2 ConvTranspose2d is a 4x4 stride-2 forward gather; when it feeds
    a (residual) BatchNorm's backward the launch takes that BatchNorm's sums and, with mask_out, stores the masked
    gradient (igemm3s2 with the fused-sums epilogue).  Bits equal to vg_conv_fwd followed by vg_bn_bwd_reduce(_dz)."""
    import ctypes as C
    from vae_gan_amd import lib as L
    n, h, w, ci, co, k, stride, pad, mask_out = case
    lib = L.load()
    dt = torch.bfloat16
    d = G.make_desc(n, h, w, ci, co, k, stride, pad, dt)
    g = torch.Generator(device="cuda").manual_seed(13)
    wt = (torch.randn(co, ci, k, k, device="cuda", generator=g) / (ci * k * k) ** 0.5).contiguous()
    wp = G.pack(lib, d, wt, 0, dt)
    gy = torch.zeros(n, h, w, d.xcs, device="cuda", dtype=dt)                 # upstream gradient (the gather's input)
    gy[..., :ci] = torch.randn(n, h, w, ci, device="cuda", generator=g).to(dt)
    shape = (n, d.oh, d.ow, d.ycs)
    xraw = torch.zeros(shape, device="cuda", dtype=dt)
    xraw[..., :co] = torch.randn(n, d.oh, d.ow, co, device="cuda", generator=g).to(dt)
    ybn = torch.zeros(shape, device="cuda", dtype=dt)
    ybn[..., :co] = torch.randn(n, d.oh, d.ow, co, device="cuda", generator=g).to(dt)
    mean, invstd, scale, shift = _bn_vec(co, 0, 6)
    cpad = 32 if co <= 32 else (64 if co <= 64 else (co + 127) // 128 * 128)
    rows = n * d.oh * d.ow
    dx0 = torch.zeros(shape, device="cuda", dtype=dt)
    L.check(lib.vg_conv_fwd(C.byref(d), gy.data_ptr(), wp.data_ptr(), None, dx0.data_ptr(), None, L.ACT_NONE, 0, L.stream()), "fwd")
    plain_kernel = lib.vg_last_conv_kernel()
    want = torch.zeros(L.STATS_REPLICAS * 2 * cpad, dtype=torch.float64, device="cuda")
    ref = dx0
    if mask_out:
        ref = torch.empty_like(dx0)
        L.check(lib.vg_bn_bwd_reduce_dz(dx0.data_ptr(), ybn.data_ptr(), xraw.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                        scale.data_ptr(), shift.data_ptr(), want.data_ptr(), ref.data_ptr(), rows, d.ycs, co, cpad,
                                        L.ACT_LRELU, L.VG_BF16, L.stream()), "bn_bwd_reduce_dz")
    else:
        L.check(lib.vg_bn_bwd_reduce(dx0.data_ptr(), ybn.data_ptr(), xraw.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                     scale.data_ptr(), shift.data_ptr(), want.data_ptr(), rows, d.ycs, co, cpad, L.ACT_LRELU,
                                     L.VG_BF16, L.stream()), "bn_bwd_reduce")
    dx1 = torch.zeros(shape, device="cuda", dtype=dt)
    got = torch.zeros_like(want)
    bs = L.BnBwdStats(x=xraw.data_ptr(), y=ybn.data_ptr(), mean=mean.data_ptr(), invstd=invstd.data_ptr(), scale=scale.data_ptr(),
                      shift=shift.data_ptr(), sums=got.data_ptr(), cpad=cpad, act=L.ACT_LRELU, mask_out=mask_out)
    fused = C.c_int(-1)
    L.check(lib.vg_conv_fwd_bnstats(C.byref(d), gy.data_ptr(), wp.data_ptr(), dx1.data_ptr(), 0, C.byref(bs), C.byref(fused),
                                    L.stream()), "fwd_bnstats")
    torch.cuda.synchronize()
    assert lib.vg_last_conv_kernel() == plain_kernel
    if d.ow >= 24:
        assert plain_kernel.startswith(b"igemm3s2") and fused.value == 1, (plain_kernel, fused.value)
    if fused.value == 0:
        assert torch.equal(dx1, dx0) and float(got.abs().max()) == 0.0
        return
    assert torch.equal(dx1, ref)
    a = want.view(L.STATS_REPLICAS, 2, cpad).sum(0)
    b = got.view(L.STATS_REPLICAS, 2, cpad).sum(0)
    scale_ = a.abs().max(dim=1, keepdim=True).values
    assert float(((a - b).abs() / scale_).max()) <= 2e-6, float(((a - b).abs() / scale_).max())


def test_dgrad_bnstats_reports_unfused_kernels(gpu):
    """Kernel variants without the fused sums (f32 parity mode, narrow images) leave the sums alone and say so."""
    import ctypes as C
    from vae_gan_amd import lib as L
    lib = L.load()
    d = G.make_desc(2, 16, 16, 64, 64, 3, 1, 1, torch.bfloat16)          # 16 pixels wide: not a ping-pong tile shape
    g = torch.Generator(device="cuda").manual_seed(1)
    wt = torch.randn(64, 64, 3, 3, device="cuda", generator=g)
    wpt = G.pack(lib, d, wt, 1, torch.bfloat16)
    gy = torch.randn(2, 16, 16, 64, device="cuda", generator=g).bfloat16()
    xraw = torch.randn(2, 16, 16, 64, device="cuda", generator=g).bfloat16()
    dx = torch.empty_like(xraw)
    mean, invstd, scale, shift = _bn_vec(64, 0, 2)
    sums = torch.zeros(L.STATS_REPLICAS * 2 * 64, dtype=torch.float64, device="cuda")
    bs = L.BnBwdStats(x=xraw.data_ptr(), y=None, mean=mean.data_ptr(), invstd=invstd.data_ptr(), scale=scale.data_ptr(),
                      shift=shift.data_ptr(), sums=sums.data_ptr(), cpad=64, act=L.ACT_LRELU)
    fused = C.c_int(-1)
    L.check(lib.vg_conv_dgrad_bnstats(C.byref(d), gy.data_ptr(), wpt.data_ptr(), dx.data_ptr(), 0, C.byref(bs), C.byref(fused),
                                      L.stream()), "dgrad_bnstats")
    torch.cuda.synchronize()
    assert fused.value == 0 and float(sums.abs().max()) == 0.0 and torch.isfinite(dx.float()).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_batched_repack_matches_single_pack(gpu, dtype):
    """vg_conv_pack_batch (one launch for every packed image of an optimizer) writes the same bits as vg_conv_pack_weight,
    padding included: plain and transposed images, 3x3 / 4x4 stride 2 (four phases of 2x2 taps) / 1x1 / 4x4 stride 1,
    thin and ragged channel counts."""
    import ctypes as C
    import numpy as np
    from vae_gan_amd import lib as L
    lib = L.load()
    geos = [(64, 64, 3, 1, 1), (128, 64, 4, 2, 1), (3, 64, 3, 1, 1), (64, 3, 3, 1, 1), (512, 64, 1, 1, 0), (512, 1, 4, 1, 1),
            (320, 128, 3, 1, 1), (96, 40, 4, 2, 1), (512, 512, 3, 1, 1)]
    isz = lib.vg_conv_pack_item_bytes()
    table = np.zeros(len(geos) * 2 * isz, dtype=np.uint8)
    keep, want, got = [], [], []
    gen = torch.Generator(device="cuda").manual_seed(2)
    for k_, (ci, co, k, stride, pad) in enumerate(geos):
        d = G.make_desc(1, 16, 16, ci, co, k, stride, pad, dtype)
        w = torch.randn(co, ci, k, k, device="cuda", generator=gen)
        keep.append((d, w))
        for tr in (0, 1):
            want.append(G.pack(lib, d, w, tr, dtype))
            buf = torch.full_like(want[-1], float("nan"))
            got.append(buf)
            item = (C.c_uint8 * isz)()
            L.check(lib.vg_conv_pack_item(C.byref(d), tr, w.data_ptr(), buf.data_ptr(), item), "pack_item")
            table[(2 * k_ + tr) * isz:(2 * k_ + tr + 1) * isz] = np.frombuffer(item, dtype=np.uint8)
    dev = torch.from_numpy(table).cuda()
    L.check(lib.vg_conv_pack_batch(dev.data_ptr(), len(got), L.VG_BF16 if dtype == torch.bfloat16 else L.VG_F32, L.stream()), "pack_batch")
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(want, got)):
        assert torch.equal(a.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                           b.view(torch.int16 if dtype == torch.bfloat16 else torch.int32)), (geos[i // 2], i % 2)


@pytest.mark.parametrize("case", [(64, 16, 16, 512, 0), (2, 8, 8, 512, 0), (3, 9, 7, 64, 1), (2, 5, 33, 24, 0), (2, 6, 20, 1024, 2)],
                         ids=lambda c: "n%d_%dx%d_ci%d_act%d" % c)
def test_patchgan_tail_kernels_match_float64_and_gather_gemm(gpu, case):
    """csrc/tail.hip (vg_conv1_fwd / vg_conv1_wgrad): the one-output-channel 4x4 stride-1 conv on the vector ALUs,
    against torch float64 on the bf16-rounded operands and against vg_conv_fwd / vg_conv_wgrad (same bound as
    test_conv_primitives); padding channels zero; the weight gradient bit-reproducible, workspace-independent, and
    accumulate=1 adds exactly once.  Cases: the step's shape, ragged rows, the widest row (32 outputs), two channel
    rounds (1024), fused LeakyReLU / sigmoid."""
    import ctypes as C
    import torch.nn.functional as F
    from vae_gan_amd import lib as L
    n, h, w, ci, act = case
    dt = torch.bfloat16
    lib = L.load()
    d = G.make_desc(n, h, w, ci, 1, 4, 1, 1, dt)
    assert lib.vg_conv1_supported(C.byref(d)) == 1
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(1, ci, 4, 4, generator=g) / (ci * 16) ** 0.5
    bias = torch.randn(1, generator=g)
    xr, wr = x.to(dt).double(), wt.to(dt).double()
    y_ref = F.conv2d(xr, wr, bias.double(), stride=1, padding=1)
    if act == L.ACT_LRELU:
        y_ref = F.leaky_relu(y_ref, 0.2)
    elif act == L.ACT_SIGMOID:
        y_ref = torch.sigmoid(y_ref)
    gy = torch.randn(y_ref.shape, generator=g)
    dw_ref = torch.nn.grad.conv2d_weight(xr, wr.shape, gy.to(dt).double(), stride=1, padding=1)
    xd, wd, bd, gyd = G.nhwc(x, d.xcs, dt), wt.cuda(), bias.cuda(), G.nhwc(gy, d.ycs, dt)
    y = torch.full((n, d.oh, d.ow, d.ycs), float("nan"), dtype=dt, device="cuda")
    L.check(lib.vg_conv1_fwd(C.byref(d), xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), act, L.stream()), "conv1_fwd")
    y2 = torch.empty_like(y)
    wp = G.pack(lib, d, wd, 0, dt)
    L.check(lib.vg_conv_fwd(C.byref(d), xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), y2.data_ptr(), None, act, 0, L.stream()), "fwd")
    torch.cuda.synchronize()
    assert G.rel(G.from_nhwc(y, 1), y_ref) <= 6e-3
    assert float(y[..., 1:].float().abs().max()) == 0.0
    assert G.rel(y[..., 0].float(), y2[..., 0].float()) <= 6e-3
    ws = torch.empty(lib.vg_conv1_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(2):
        dw = torch.full((1, ci, 4, 4), float("nan"), dtype=torch.float32, device="cuda")
        ws.fill_(float("nan"))
        L.check(lib.vg_conv1_wgrad(C.byref(d), xd.data_ptr(), gyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream()), "conv1_wgrad")
        outs.append(dw)
    torch.cuda.synchronize()
    assert G.rel(outs[0].cpu(), dw_ref) <= 5e-3
    assert torch.equal(outs[0], outs[1])
    acc = outs[0].clone()
    L.check(lib.vg_conv1_wgrad(C.byref(d), xd.data_ptr(), gyd.data_ptr(), acc.data_ptr(), ws.data_ptr(), 1, L.stream()), "conv1_wgrad")
    torch.cuda.synchronize()
    assert torch.equal(acc, outs[0] * 2)


def test_patchgan_tail_rejects_other_shapes(gpu):
    import ctypes as C
    from vae_gan_amd import lib as L
    lib = L.load()
    for n, h, w, ci, co, k, s, p, dt in [(2, 8, 8, 512, 2, 4, 1, 1, torch.bfloat16), (2, 8, 8, 512, 1, 3, 1, 1, torch.bfloat16),
                                         (2, 8, 8, 512, 1, 4, 2, 1, torch.bfloat16), (2, 8, 40, 512, 1, 4, 1, 1, torch.bfloat16),
                                         (2, 8, 8, 512, 1, 4, 1, 1, torch.float32)]:
        d = G.make_desc(n, h, w, ci, co, k, s, p, dt)
        assert lib.vg_conv1_supported(C.byref(d)) == 0
    d = G.make_desc(2, 8, 8, 512, 2, 4, 1, 1, torch.bfloat16)
    t = torch.zeros(16, device="cuda")
    assert lib.vg_conv1_fwd(C.byref(d), t.data_ptr(), t.data_ptr(), None, t.data_ptr(), 0, L.stream()) != 0
