"""Manual timing of the split-K forward on the PatchGAN tail (512 -> 1, 4x4 stride 1 on 16x16, batch 64) over the K split:
PYTHONPATH=.:tests python tests/gpu_splitk_perf.py"""
import ctypes as C
import torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc, pack

lib = L.load()
dt = torch.bfloat16
d = make_desc(64, 16, 16, 512, 1, 4, 1, 1, dt)
x = torch.randn(64, 16, 16, d.xcs, device="cuda").to(dt)
wt = torch.randn(1, 512, 4, 4, device="cuda") * 0.02
wp = pack(lib, d, wt, 0, dt)
y = torch.empty(64, d.oh, d.ow, d.ycs, dtype=dt, device="cuda")
ws = torch.empty(lib.vg_conv_splitk_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
for ks in (0, 1, 4, 9, 16, 24, 32):
    fn = lambda: L.check(lib.vg_conv_fwd_splitk(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), 0, ks, ws.data_ptr(), L.stream()))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"ksplit {ks:3d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us  ({lib.vg_last_conv_kernel().decode()})", flush=True)


def timeit(fn, tag):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)


# the vector-ALU kernels of csrc/tail.hip against the gather-GEMM paths they replace
gy = torch.randn(64, d.oh, d.ow, d.ycs, device="cuda").to(dt)
dw = torch.empty(1, 512, 4, 4, device="cuda")
ws1 = torch.empty(lib.vg_conv1_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
wsg = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
timeit(lambda: L.check(lib.vg_conv1_fwd(C.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, L.stream())), "conv1 fwd      ")
timeit(lambda: L.check(lib.vg_conv1_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws1.data_ptr(), 0, L.stream())), "conv1 wgrad    ")
timeit(lambda: L.check(lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), wsg.data_ptr(), 0, L.stream())), "gather wgrad   ")
