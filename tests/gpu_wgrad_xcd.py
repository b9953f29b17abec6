"""Spatial-tile weight gradient: plain dealing against XCD dealing of the tile groups (VG_WG3_XCD=0/1, read once per
process: run this file once per setting).  Prints time, TFLOP/s and the largest difference to the per-tap kernel."""
import ctypes as C, os, torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc

def run(n, h, w, ci, co, k, stride, pad, iters=10):
    lib = L.load(); dtype = torch.bfloat16
    torch.manual_seed(0)
    d = make_desc(n, h, w, ci, co, k, stride, pad, dtype)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dtype)
    gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda").to(dtype)
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    flops = 2.0 * n * d.oh * d.ow * co * ci * k * k
    out = {}
    for mode in (1, 4):
        dw = torch.empty(co, ci, k, k, dtype=torch.float32, device="cuda")
        lib.vg_set_wgrad_kernel(mode)
        fn = lambda: lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream())
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters if mode == 4 else 1): fn()
        e1.record(); torch.cuda.synchronize()
        out[mode] = (e0.elapsed_time(e1) / (iters if mode == 4 else 1), dw)
    lib.vg_set_wgrad_kernel(4)
    ms = out[4][0]
    err = ((out[4][1] - out[1][1]).abs().max() / out[1][1].abs().max()).item()
    print(f"xcd={os.environ.get('VG_WG3_XCD', '1')} {h}x{w} ci{ci} co{co} k{k}s{stride}: {ms:.3f} ms {flops/ms/1e9:.0f} TF  {lib.vg_last_wgrad_kernel().decode()}  rel diff to per-tap {err:.2e}", flush=True)

for c in [(64, 256, 256, 64, 64, 3, 1, 1), (64, 128, 128, 128, 128, 3, 1, 1), (64, 64, 64, 256, 256, 3, 1, 1),
          (64, 32, 32, 512, 512, 3, 1, 1), (64, 256, 256, 64, 128, 4, 2, 1), (64, 128, 128, 128, 256, 4, 2, 1),
          (64, 64, 64, 256, 512, 4, 2, 1), (64, 32, 32, 256, 512, 4, 2, 1), (16, 64, 64, 320, 128, 3, 1, 1), (2, 64, 64, 256, 256, 3, 1, 1)]:
    run(*c)
