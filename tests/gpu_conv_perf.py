"""Micro-benchmark of the conv primitives at the VQ-VAE-GAN layer shapes (B=64, 256x256)."""
import ctypes as C
import sys
import torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc, pack

def bench(n, h, w, ci, co, k, stride, pad, dtype, iters=5):
    lib = L.load()
    d = make_desc(n, h, w, ci, co, k, stride, pad, dtype)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dtype)
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda").to(dtype)
    wp, wpt = pack(lib, d, wt, 0, dtype), pack(lib, d, wt, 1, dtype)
    y = torch.empty(n, d.oh, d.ow, d.ycs, dtype=dtype, device="cuda")
    dx = torch.empty(n, h, w, d.xcs, dtype=dtype, device="cuda")
    npad = 32 if co <= 32 else (64 if co <= 64 else (co + 127) // 128 * 128)
    stats = torch.zeros(L.STATS_REPLICAS, 2, npad, dtype=torch.float64, device="cuda")
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    dw = torch.empty(co, ci, k, k, device="cuda")
    flops = 2.0 * n * d.oh * d.ow * co * ci * k * k
    out = []
    for name, fn in (("fwd", lambda: lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), 0, 0, L.stream())),
                     ("dgrad", lambda: lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), None, 0, 0, L.stream())),
                     ("wgrad", lambda: lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream()))):
        L.check(fn(), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out.append(f"{name} {ms:7.3f} ms {flops/ms/1e9:7.1f} TF/s")
    print(f"n{n} {h}x{w} ci{ci:3d} co{co:3d} k{k} s{stride} {str(dtype)[6:]:8s} {flops/1e9:7.1f} GF | " + " | ".join(out), flush=True)

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dt = torch.bfloat16
    for c in [(B, 256, 256, 3, 64, 3, 1, 1), (B, 256, 256, 64, 64, 3, 1, 1), (B, 256, 256, 64, 128, 4, 2, 1),
              (B, 128, 128, 128, 128, 3, 1, 1), (B, 128, 128, 128, 256, 4, 2, 1), (B, 64, 64, 256, 256, 3, 1, 1),
              (B, 64, 64, 256, 512, 4, 2, 1), (B, 32, 32, 512, 512, 3, 1, 1), (B, 32, 32, 512, 64, 1, 1, 0),
              (B, 256, 256, 64, 3, 3, 1, 1), (B, 256, 256, 3, 64, 4, 2, 1), (B, 16, 16, 512, 1, 4, 1, 1)]:
        bench(*c, dt)
    bench(8, 128, 128, 128, 128, 3, 1, 1, torch.float32)
    bench(8, 256, 256, 64, 64, 3, 1, 1, torch.float32)
