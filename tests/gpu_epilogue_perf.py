"""Manual timing of the igemm3 input-gradient launch with each epilogue variant (plain, += onto a parked skip gradient,
BatchNorm-backward sums, sums + masked store, both) at the wide ResidualBlock shapes:
PYTHONPATH=.:tests python tests/gpu_epilogue_perf.py   (VG_LIB_EXP=1: ../libvg_hip_exp.so)"""
import ctypes as C
import os
import torch
from vae_gan_amd import lib as L
if os.environ.get("VG_LIB_EXP"):
    _v = os.environ["VG_LIB_EXP"]
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libvg_hip_{'exp' if _v == '1' else _v}.so")
from gpu_ops_check import make_desc, pack

lib = L.load()
dt = torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        L.check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (n, h, w, ci, co, k, s, p) in [(64, 64, 64, 256, 256, 3, 1, 1), (64, 32, 32, 512, 512, 3, 1, 1), (64, 128, 128, 128, 256, 4, 2, 1)]:
    d = make_desc(n, h, w, ci, co, k, s, p, dt)
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda").to(dt)
    wpt = pack(lib, d, wt, 1, dt)
    dx = torch.zeros(n, h, w, d.xcs, dtype=dt, device="cuda")
    bx = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    by = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    cpad = (ci + 127) // 128 * 128
    vec = torch.rand(4, ci, device="cuda") + 0.5
    sums = torch.zeros(L.STATS_REPLICAS * 2 * cpad + 2 * cpad, dtype=torch.float64, device="cuda")
    fused = C.c_int(0)

    def bs(y, mask_out):
        return L.BnBwdStats(x=bx.data_ptr(), y=(by.data_ptr() if y else None), mean=vec[0].data_ptr(), invstd=vec[1].data_ptr(),
                            scale=vec[2].data_ptr(), shift=vec[3].data_ptr(), sums=sums.data_ptr(), cpad=cpad, act=L.ACT_LRELU,
                            mask_out=mask_out)
    out = []
    out.append(("plain", timeit(lambda: lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), None, 0, 0, L.stream()))))
    out.append(("accum", timeit(lambda: lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), None, 0, 1, L.stream()))))
    for tag, y, m, acc in (("sums", 0, 0, 0), ("sums+y+mask", 1, 1, 0), ("sums+accum", 0, 0, 1), ("sums+y+mask+accum", 1, 1, 1)):
        b = bs(y, m)
        out.append((tag, timeit(lambda: lib.vg_conv_dgrad_bnstats(C.byref(d), gy.data_ptr(), wpt.data_ptr(), dx.data_ptr(), acc, C.byref(b),
                                                                    C.byref(fused), L.stream()))))
    print(f"n{n} {h}x{w} {ci}->{co} k{k}s{s} [{lib.vg_last_conv_kernel().decode()} fused={fused.value}]: " +
          "  ".join(f"{t} {v:.0f}us" for t, v in out), flush=True)
