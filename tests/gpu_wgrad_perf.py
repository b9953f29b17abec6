"""Weight-gradient microbench at the VQ-VAE-GAN layer shapes (batch 64, bf16): per-tap kernel vs spatial-tile kernel."""
import ctypes as C, os, torch
from vae_gan_amd import lib as L
if os.environ.get("VG_LIB_EXP"):      # A/B against another build: libvg_hip_<tag>.so
    _v = os.environ["VG_LIB_EXP"]
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libvg_hip_{'exp' if _v == '1' else _v}.so")
from gpu_ops_check import make_desc

def run(n, h, w, ci, co, k, stride, pad, iters=20):
    lib = L.load(); dtype = torch.bfloat16
    d = make_desc(n, h, w, ci, co, k, stride, pad, dtype)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dtype)
    gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda").to(dtype)
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    dw = torch.empty(co, ci, k, k, dtype=torch.float32, device="cuda")
    flops = 2.0 * n * d.oh * d.ow * co * ci * k * k
    gb = (x.numel() + gy.numel()) * 2 / 1e9
    res = []
    for mode in ((4,) if os.environ.get('VG_WGP_FAST') else (1, 3, 4)):
        lib.vg_set_wgrad_kernel(mode)
        fn = lambda: lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream())
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res.append(f"v{mode} {ms:.3f}ms {flops/ms/1e9:.0f}TF {gb/ms*1e3:.0f}GB/s")
    lib.vg_set_wgrad_kernel(4)
    print(f"{h}x{w} ci{ci} co{co} k{k}s{stride}: " + " | ".join(res), flush=True)

for c in [(64, 256, 256, 64, 64, 3, 1, 1), (64, 128, 128, 128, 128, 3, 1, 1), (64, 64, 64, 256, 256, 3, 1, 1),
          (64, 32, 32, 512, 512, 3, 1, 1), (64, 256, 256, 3, 64, 3, 1, 1), (64, 256, 256, 64, 3, 3, 1, 1),
          (64, 256, 256, 64, 128, 4, 2, 1), (64, 128, 128, 128, 256, 4, 2, 1), (64, 64, 64, 256, 512, 4, 2, 1),
          (64, 256, 256, 3, 64, 4, 2, 1)]:
    run(*c)
