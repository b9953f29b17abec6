"""Manual perf script: forward of the 4x4 stride-2 convs at batch 64, im2col-row kernel (vg_set_conv_kernel(3|16))
against the parity-plane spatial-tile kernel (igemm3s2, default), and the two outputs against each other."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc, pack

lib = L.load()
dt = torch.bfloat16
for (n, h, w, ci, co) in [(64, 256, 256, 64, 128), (64, 128, 128, 128, 256), (64, 64, 64, 256, 512), (64, 256, 256, 3, 64),
                          (64, 128, 128, 64, 128), (64, 64, 64, 128, 256), (64, 32, 32, 256, 512)]:
    d = make_desc(n, h, w, ci, co, 4, 2, 1, dt)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    if d.xcs > ci:
        x[..., ci:] = 0
    wt = torch.randn(co, ci, 4, 4, device="cuda") * 0.05
    wp = pack(lib, d, wt, 0, dt)
    npad = 32 if co <= 32 else (64 if co <= 64 else (co + 127) // 128 * 128)
    stats = torch.zeros(L.STATS_REPLICAS, 2, npad, dtype=torch.float64, device="cuda")
    flops = 2.0 * n * d.oh * d.ow * co * ci * 16
    outs, res = [], []
    for mode in (3 | 16, 3):
        lib.vg_set_conv_kernel(mode)
        y = torch.empty(n, d.oh, d.ow, d.ycs, dtype=dt, device="cuda")
        fn = lambda: lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), 1, 0, L.stream())
        L.check(fn()); torch.cuda.synchronize()
        name = lib.vg_last_conv_kernel().decode()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res.append(f"{name} {ms:.3f} ms {flops/ms/1e9:.0f} TF/s")
        outs.append(y.float())
    lib.vg_set_conv_kernel(3)
    err = float((outs[0] - outs[1]).norm() / outs[0].norm())
    print(f"{h}x{w} ci{ci} co{co}: " + " | ".join(res) + f" | rel diff {err:.2e}", flush=True)
