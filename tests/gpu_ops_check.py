"""Stand-alone GPU check of the conv primitives through the raw C ABI (also used by
tests/test_gpu_conv.py).  Compares vg_conv_fwd / vg_conv_dgrad / vg_conv_wgrad against
torch CPU float64 convolutions on the same (dtype-rounded) inputs."""
import ctypes as C
import sys

import torch
import torch.nn.functional as F

from vae_gan_amd import lib as L


def nhwc(t, cs, dtype):
    """NCHW float tensor -> NHWC padded tensor of dtype on cuda."""
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cs, dtype=dtype)
    out[..., :c] = t.permute(0, 2, 3, 1).to(dtype)
    return out.cuda()


def from_nhwc(t, c):
    return t[..., :c].permute(0, 3, 1, 2).double().cpu()


def make_desc(n, h, w, ci, co, k, stride, pad, dtype):
    oh = (h + 2 * pad - k) // stride + 1
    ow = (w + 2 * pad - k) // stride + 1
    d = L.ConvDesc(n=n, h=h, w=w, oh=oh, ow=ow, ci=ci, co=co, xcs=L.pad_channels(ci, dtype),
                   ycs=L.pad_channels(co, dtype), r=k, s=k, stride=stride, pad=pad, dtype=L.vg_dtype(dtype))
    return d


def pack(lib, d, w, transposed, dtype):
    nbytes = lib.vg_conv_packed_bytes(C.byref(d), transposed)
    assert nbytes > 0
    buf = torch.empty(nbytes // (2 if dtype == torch.bfloat16 else 4), dtype=dtype, device="cuda")
    L.check(lib.vg_conv_pack_weight(C.byref(d), transposed, w.data_ptr(), buf.data_ptr(), L.stream()), "pack")
    return buf


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def check_conv(n, h, w, ci, co, k, stride, pad, dtype, seed=0, verbose=True):
    lib = L.load()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    bias = torch.randn(co, generator=g)
    d = make_desc(n, h, w, ci, co, k, stride, pad, dtype)
    # inputs rounded to the storage dtype so the comparison isolates the kernel arithmetic
    xr = x.to(dtype).double()
    wr = wt.to(dtype).double()
    y_ref = F.conv2d(xr, wr, bias.double(), stride=stride, padding=pad)
    gy = torch.randn(y_ref.shape, generator=g)
    gyr = gy.to(dtype).double()
    dx_ref = torch.nn.grad.conv2d_input(xr.shape, wr, gyr, stride=stride, padding=pad)
    dw_ref = torch.nn.grad.conv2d_weight(xr, wr.shape, gyr, stride=stride, padding=pad)

    xd = nhwc(x, d.xcs, dtype)
    wd = wt.cuda()
    bd = bias.cuda()
    wp = pack(lib, d, wd, 0, dtype)
    wpt = pack(lib, d, wd, 1, dtype)
    npad = 32 if co <= 32 else (64 if co <= 64 else (co + 127) // 128 * 128)
    stats = torch.zeros(L.STATS_REPLICAS, 2, npad, dtype=torch.float64, device="cuda")
    y = torch.full((n, d.oh, d.ow, d.ycs), float("nan"), dtype=dtype, device="cuda")
    L.check(lib.vg_conv_fwd(C.byref(d), xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), y.data_ptr(), stats.data_ptr(),
                            L.ACT_NONE, 0, L.stream()), "fwd")
    gyd = nhwc(gy, d.ycs, dtype)
    dx = torch.full((n, h, w, d.xcs), float("nan"), dtype=dtype, device="cuda")
    L.check(lib.vg_conv_dgrad(C.byref(d), gyd.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), None, L.ACT_NONE, 0,
                              L.stream()), "dgrad")
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    dw = torch.full((co, ci, k, k), float("nan"), dtype=torch.float32, device="cuda")
    L.check(lib.vg_conv_wgrad(C.byref(d), xd.data_ptr(), gyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream()), "wgrad")
    torch.cuda.synchronize()
    res = {}
    yc = from_nhwc(y, co)
    res["fwd"] = rel(yc, y_ref)
    # padded output channels must be exactly zero
    res["pad0"] = float(y[..., co:].float().abs().max()) if d.ycs > co else 0.0
    res["dgrad"] = rel(from_nhwc(dx, ci), dx_ref)
    res["wgrad"] = rel(dw.cpu(), dw_ref)
    s = stats.sum(0).cpu()
    yq = yc.to(dtype).double()
    res["stat_sum"] = rel(s[0, :co], yq.sum((0, 2, 3)))
    res["stat_sq"] = rel(s[1, :co], (yq ** 2).sum((0, 2, 3)))
    if verbose:
        print(f"conv n{n} {h}x{w} ci{ci} co{co} k{k} s{stride} p{pad} {str(dtype)[6:]}: " +
              " ".join(f"{k_}={v:.2e}" for k_, v in res.items()), flush=True)
    return res


CASES = [
    # n, h, w, ci, co, k, stride, pad
    (2, 16, 16, 64, 64, 3, 1, 1),
    (2, 16, 16, 64, 128, 4, 2, 1),
    (1, 9, 7, 32, 64, 3, 1, 1),       # ragged spatial size, partial tiles
    (2, 16, 16, 3, 64, 3, 1, 1),      # thin input (image)
    (2, 16, 16, 64, 3, 3, 1, 1),      # thin output (decoder tail)
    (2, 16, 16, 3, 64, 4, 2, 1),      # discriminator head
    (2, 8, 8, 512, 1, 4, 1, 1),       # PatchGAN tail
    (2, 8, 8, 512, 64, 1, 1, 0),      # 1x1 to embedding dim
    (1, 16, 16, 128, 256, 4, 2, 1),
    (1, 8, 8, 320, 128, 3, 1, 1),     # hierarchical decoder concat width
    (1, 8, 8, 256, 512, 3, 1, 1),
    # wide rows: the ping-pong spatial-tile kernel (>= 24 px wide, >= 4 taps, stride-1 gather)
    (2, 40, 36, 64, 64, 3, 1, 1),     # 64-channel config, ragged tiles
    (1, 33, 65, 96, 128, 3, 1, 1),    # 128-channel config, 3 chunks, 3 tile columns
    (1, 20, 32, 128, 256, 3, 1, 1),   # 256-channel config
    (1, 32, 32, 64, 256, 4, 1, 1),    # 4x4 stride-1 (discriminator tail), 31x31 output
    (1, 64, 64, 64, 128, 4, 2, 1),    # dgrad = 4 phases of 2x2 taps, 64-channel config with 3 halo buffers
    (1, 64, 64, 128, 64, 4, 2, 1),    # dgrad with 128 output channels
    # forward of 4x4 stride-2 convs on the parity planes (igemm3s2: >= 24 output px wide)
    (2, 48, 80, 64, 128, 4, 2, 1),    # ragged tiles in both directions (24 x 40 outputs)
    (1, 64, 64, 3, 64, 4, 2, 1),      # thin input (discriminator head): one 64-byte chunk per plane
    (1, 64, 128, 256, 512, 4, 2, 1),  # 256-channel config, 8 chunks per plane
    (1, 52, 60, 32, 64, 4, 2, 1),     # 64-channel config, 30-px rows
    # weights-stationary persistent kernel (igemm4: 64 -> 64 3x3, whole 8 x 32 tiles)
    (2, 32, 64, 64, 64, 3, 1, 1),     # fewer tiles than blocks
    (6, 96, 128, 64, 64, 3, 1, 1),    # 288 tiles on 256 blocks: a second tile for some, XCD-ordered hand-out
    # its variants: output channels split over the waves (igemm4c: 128 -> 128 3x3, 4 x 32 tiles), a phase per wave
    # (igemm4p: the input gradient of a 64 -> 128 4x4 stride-2 conv = 128 -> 64 transposed gather, 2 x 32 tiles)
    (3, 36, 96, 128, 128, 3, 1, 1),   # 81 tiles
    (5, 72, 128, 128, 128, 3, 1, 1),  # 360 tiles on 256 blocks
    (5, 72, 128, 64, 128, 4, 2, 1),   # dgrad: 5 x 18 x 2 = 180 tiles of 2 x 32 input pixels
]

if __name__ == "__main__":
    torch.manual_seed(0)
    bad = 0
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 6e-3)):
        for c in CASES:
            r = check_conv(*c, dtype)
            for k_ in ("fwd", "dgrad", "wgrad", "stat_sum", "stat_sq"):
                lim = tol if k_ != "wgrad" or dtype == torch.float32 else 1e-4 * 50
                if not (r[k_] <= lim):
                    print("   ^^^ FAIL", k_, r[k_], "tol", lim)
                    bad += 1
            if r["pad0"] != 0.0:
                print("   ^^^ FAIL padded channels not zero")
                bad += 1
    print("failures:", bad)
    sys.exit(1 if bad else 0)
