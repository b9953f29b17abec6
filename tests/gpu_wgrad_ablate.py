"""Timing ablations of the 3x3 weight-gradient kernel (ablation build; results are wrong by construction)."""
import ctypes as C, os, torch
from vae_gan_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libvg_hip_ablate.so")
from gpu_ops_check import make_desc

def run(n, h, w, ci, co, k, stride, pad, iters=5):
    lib = L.load(); dt = torch.bfloat16
    d = make_desc(n, h, w, ci, co, k, stride, pad, dt)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda").to(dt)
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    dw = torch.empty(co, ci, k, k, device="cuda")
    flops = 2.0 * n * d.oh * d.ow * co * ci * k * k
    res = []
    for bits, name in ((0, "full"), (16, "oneB"), (32, "noDMA"), (64, "nobarrier"), (16 | 32, "oneB+noDMA"), (16 | 32 | 64, "mfma+A only"), (128, "no stores"), (16 | 32 | 64 | 128, "mfma+A, no stores")):
        lib.vg_set_conv_kernel(3 | (bits << 8))
        fn = lambda: lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream())
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res.append(f"{name} {ms:.3f}ms {flops/ms/1e9:.0f}TF")
    lib.vg_set_conv_kernel(3)
    print(f"{h}x{w} ci{ci} co{co}: " + " | ".join(res), flush=True)

for c in [(64, 256, 256, 64, 64, 3, 1, 1), (64, 128, 128, 128, 128, 3, 1, 1), (64, 64, 64, 256, 256, 3, 1, 1), (64, 32, 32, 512, 512, 3, 1, 1)]:
    run(*c)
