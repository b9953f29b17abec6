"""GPU tier: the thin-layer kernels (csrc/thin.hip: convolutions with a 3-channel side on 4-channel NHWC tensors)
through the raw C ABI against torch CPU float64 convolutions on the same bf16-rounded inputs: forward (+ bias, fused
BatchNorm statistics, LeakyReLU / Sigmoid), input gradient, weight gradient; even and ragged tile shapes."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _nhwc(t, cs):
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cs, dtype=BF)
    out[..., :c] = t.permute(0, 2, 3, 1).to(BF)
    return out.cuda()


def _nchw(t, c):
    return t[..., :c].permute(0, 3, 1, 2).double().cpu()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


# (n, h, w, ci, co, k, stride)
CASES = [(2, 64, 64, 3, 64, 3, 1), (2, 40, 36, 3, 64, 3, 1), (1, 9, 70, 3, 64, 3, 1),
         (2, 64, 64, 3, 64, 4, 2), (2, 40, 72, 3, 64, 4, 2), (1, 18, 130, 3, 64, 4, 2),
         (2, 64, 64, 64, 3, 3, 1), (2, 40, 36, 64, 3, 3, 1), (1, 5, 33, 64, 3, 3, 1), (2, 16, 32, 3, 48, 3, 1)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dx%d_%d_to_%d_k%ds%d" % c)
def test_thin_conv_against_float64(case):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import lib as L
    lib = L.load()
    n, h, w, ci, co, k, stride = case
    pad = 1
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    xcs, ycs = (4 if ci <= 3 else 64), (4 if co <= 3 else 64)
    d = L.ConvDesc(n=n, h=h, w=w, oh=oh, ow=ow, ci=ci, co=co, xcs=xcs, ycs=ycs, r=k, s=k, stride=stride, pad=pad, dtype=L.VG_BF16)
    kind = lib.vg_thin_conv_supported(C.byref(d))
    assert kind == (1 if ci <= 3 else 2)
    g = torch.Generator().manual_seed(n * 1000 + h + w + k)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    bias = torch.randn(co, generator=g)
    act = L.ACT_SIGMOID if kind == 2 else (L.ACT_LRELU if stride == 2 else L.ACT_NONE)
    xr, wr = x.to(BF).double(), wt.to(BF).double()
    pre = F.conv2d(xr, wr, bias.double(), stride=stride, padding=pad)
    pre_r = pre.to(BF).double()                     # the kernels round the pre-activation to bf16 first
    y_ref = torch.sigmoid(pre_r) if act == L.ACT_SIGMOID else (F.leaky_relu(pre_r, 0.2) if act == L.ACT_LRELU else pre_r)
    gy = torch.randn(pre.shape, generator=g)
    gyr = gy.to(BF).double()
    dx_ref = torch.nn.grad.conv2d_input(xr.shape, wr, gyr, stride=stride, padding=pad)
    dw_ref = torch.nn.grad.conv2d_weight(xr, wr.shape, gyr, stride=stride, padding=pad)

    xd, gyd = _nhwc(x, xcs), _nhwc(gy, ycs)
    wd, bd = wt.cuda().contiguous(), bias.cuda()        # master f32 weights: the kernels round them to bf16 themselves
    y = torch.full((n, oh, ow, ycs), float("nan"), dtype=BF, device="cuda")
    cpad = 32 if co <= 32 else 64
    stats = torch.zeros(L.STATS_REPLICAS * 2 * cpad, dtype=torch.float64, device="cuda") if kind == 1 else None
    tws = torch.empty(lib.vg_thin_conv_ws_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.vg_thin_conv_fwd(C.byref(d), xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), L.ptr(stats), act,
                                 tws.data_ptr(), L.stream()), "thin_fwd")
    dx = torch.full((n, h, w, xcs), float("nan"), dtype=BF, device="cuda")
    L.check(lib.vg_thin_conv_dgrad(C.byref(d), gyd.data_ptr(), wd.data_ptr(), dx.data_ptr(), tws.data_ptr(), L.stream()), "thin_dgrad")
    ws = torch.full((lib.vg_thin_conv_wgrad_ws_bytes(C.byref(d)) // 4,), float("nan"), dtype=torch.float32, device="cuda")
    dw = torch.full((co, ci, k, k), float("nan"), dtype=torch.float32, device="cuda")
    L.check(lib.vg_thin_conv_wgrad(C.byref(d), xd.data_ptr(), gyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream()), "thin_wgrad")
    dw2 = dw.clone()
    L.check(lib.vg_thin_conv_wgrad(C.byref(d), xd.data_ptr(), gyd.data_ptr(), dw2.data_ptr(), ws.data_ptr(), 1, L.stream()), "thin_wgrad")
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all() and torch.isfinite(dx.float()).all() and torch.isfinite(dw).all()
    assert _rel(_nchw(y, co), y_ref) <= 6e-3, ("fwd", _rel(_nchw(y, co), y_ref))
    assert _rel(_nchw(dx, ci), dx_ref) <= 6e-3, ("dgrad", _rel(_nchw(dx, ci), dx_ref))
    assert _rel(dw.cpu(), dw_ref) <= 2e-5, ("wgrad", _rel(dw.cpu(), dw_ref))
    assert torch.equal(dw2, dw * 2)                                    # accumulate adds exactly once; fixed-order sums
    # padding lanes are exactly zero (they feed 16-byte-wide consumers)
    assert float(y[..., co:].float().abs().max()) == 0.0 if ycs > co else True
    assert float(dx[..., ci:].float().abs().max()) == 0.0 if xcs > ci else True
    if stats is not None:
        s = stats.view(L.STATS_REPLICAS, 2, cpad).sum(0).cpu()
        assert _rel(s[0, :co], pre_r.sum((0, 2, 3))) <= 1e-5 and _rel(s[1, :co], (pre_r ** 2).sum((0, 2, 3))) <= 1e-5


def test_thin_conv_supported_geometry():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import lib as L
    lib = L.load()

    def desc(ci, co, k, stride, xcs, ycs, dtype=L.VG_BF16, h=32, w=32, pad=1):
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        return L.ConvDesc(n=1, h=h, w=w, oh=oh, ow=ow, ci=ci, co=co, xcs=xcs, ycs=ycs, r=k, s=k, stride=stride, pad=pad, dtype=dtype)
    assert lib.vg_thin_conv_supported(C.byref(desc(3, 64, 3, 1, 4, 64))) == 1
    assert lib.vg_thin_conv_supported(C.byref(desc(3, 64, 4, 2, 4, 64))) == 1
    assert lib.vg_thin_conv_supported(C.byref(desc(64, 3, 3, 1, 64, 4))) == 2
    assert lib.vg_thin_conv_supported(C.byref(desc(3, 64, 3, 1, 32, 64))) == 0        # 32-padded tensor: the ordinary kernels
    assert lib.vg_thin_conv_supported(C.byref(desc(3, 128, 3, 1, 4, 128))) == 0       # more than 64 wide channels
    assert lib.vg_thin_conv_supported(C.byref(desc(64, 3, 4, 2, 64, 4))) == 0         # thin output only for 3x3 stride 1
    assert lib.vg_thin_conv_supported(C.byref(desc(3, 64, 3, 1, 4, 64, dtype=L.VG_F32))) == 0
    rc = lib.vg_thin_conv_fwd(C.byref(desc(3, 128, 3, 1, 4, 128)), 1, 1, None, 1, None, 0, 1, L.stream())
    assert rc != 0 and "thin" in L.last_error()
