"""Timing ablations of the spatial-tile conv kernel (results are wrong by construction; timing only)."""
import ctypes as C, sys, torch
import os
from vae_gan_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libvg_hip_ablate.so")   # make -C vae_gan_amd/csrc ablate
from gpu_ops_check import make_desc, pack

def run(n, h, w, ci, co, k, stride, pad, iters=5):
    lib = L.load(); dtype = torch.bfloat16
    d = make_desc(n, h, w, ci, co, k, stride, pad, dtype)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dtype)
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    wp = pack(lib, d, wt, 0, dtype)
    y = torch.empty(n, d.oh, d.ow, d.ycs, dtype=dtype, device="cuda")
    flops = 2.0 * n * d.oh * d.ow * co * ci * k * k
    res = []
    for mode, name in ((1, "v1"), (2, "v2"), (3, "v3"), (3 | (1 << 8), "v3-noDMA"), (3 | (2 << 8), "v3-nobarrier"),
                       (3 | (4 << 8), "v3-noLDSaddr"), (3 | (16 << 8), "v3-hotW"), (3 | (5 << 8), "v3-noDMA-noLDSaddr"),
                       (3 | (3 << 8), "v3-noDMA-nobarrier"), (3 | (7 << 8), "v3-mfma-only")):
        lib.vg_set_conv_kernel(mode)
        fn = lambda: lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, 0, 0, L.stream())
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res.append(f"{name} {ms:.3f}ms {flops/ms/1e9:.0f}TF")
    lib.vg_set_conv_kernel(3)
    print(f"{h}x{w} ci{ci} co{co} k{k}s{stride}: " + " | ".join(res), flush=True)

for c in [(64, 32, 32, 512, 512, 3, 1, 1), (64, 64, 64, 256, 256, 3, 1, 1)]:
    run(*c, iters=20)
