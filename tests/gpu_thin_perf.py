"""Timing of the thin-layer kernels at the headline shapes (B=64, 256x256): forward, input gradient, weight gradient,
with the algorithmic HBM bytes beside them."""
import ctypes as C
import torch
from vae_gan_amd import lib as L

BF = torch.bfloat16


def bench(n, h, w, ci, co, k, stride, act, iters=10):
    lib = L.load()
    pad = 1
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    xcs, ycs = (4 if ci <= 3 else 64), (4 if co <= 3 else 64)
    d = L.ConvDesc(n=n, h=h, w=w, oh=oh, ow=ow, ci=ci, co=co, xcs=xcs, ycs=ycs, r=k, s=k, stride=stride, pad=pad, dtype=L.VG_BF16)
    kind = lib.vg_thin_conv_supported(C.byref(d))
    x = torch.randn(n, h, w, xcs, device="cuda").to(BF)
    gy = torch.randn(n, oh, ow, ycs, device="cuda").to(BF)
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.1
    bias = torch.randn(co, device="cuda")
    y = torch.empty(n, oh, ow, ycs, dtype=BF, device="cuda")
    dx = torch.empty(n, h, w, xcs, dtype=BF, device="cuda")
    cpad = 32 if co <= 32 else 64
    stats = torch.zeros(L.STATS_REPLICAS * 2 * cpad, dtype=torch.float64, device="cuda") if kind == 1 else None
    tws = torch.empty(lib.vg_thin_conv_ws_bytes(), dtype=torch.uint8, device="cuda")
    ws = torch.empty(lib.vg_thin_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    dw = torch.empty(co, ci, k, k, dtype=torch.float32, device="cuda")
    mb = (x.numel() + y.numel()) * 2 / 1e6
    out = []
    for name, fn in (("fwd", lambda: lib.vg_thin_conv_fwd(C.byref(d), x.data_ptr(), wt.data_ptr(), bias.data_ptr(), y.data_ptr(), L.ptr(stats), act, tws.data_ptr(), L.stream())),
                     ("dgrad", lambda: lib.vg_thin_conv_dgrad(C.byref(d), gy.data_ptr(), wt.data_ptr(), dx.data_ptr(), tws.data_ptr(), L.stream())),
                     ("wgrad", lambda: lib.vg_thin_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream()))):
        L.check(fn(), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        out.append(f"{name} {us:6.1f} us {mb / us:5.2f} TB/s")
    print(f"{ci:2d}->{co:2d} k{k}s{stride} @{h}x{w} ({mb:.0f} MB): " + " | ".join(out), flush=True)


if __name__ == "__main__":
    bench(64, 256, 256, 3, 64, 3, 1, L.ACT_NONE)       # encoder conv0 (+ BN statistics)
    bench(64, 256, 256, 3, 64, 4, 2, L.ACT_LRELU)      # discriminator conv0
    bench(64, 256, 256, 64, 3, 3, 1, L.ACT_SIGMOID)    # decoder output conv
