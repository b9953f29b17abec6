"""A/B of the phase-per-wave weights-stationary kernel (igemm4p) against igemm3 on the 128 -> 64 stride-2 transposed
gathers (ConvTranspose2d(128, 64, 4, 2, 1) forward at 128x128 -> 256x256, and the input gradient of Conv2d(64, 128, 4, 2, 1))."""
import ctypes as C
import sys
import torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc, pack


def main(n=64, h=256, w=256, iters=10):
    lib = L.load()
    dt = torch.bfloat16
    d = make_desc(n, h, w, 64, 128, 4, 2, 1, dt)          # conv 64 -> 128, stride 2: its dgrad is the transposed gather
    g = torch.Generator(device="cuda").manual_seed(5)
    wt = torch.randn(128, 64, 4, 4, device="cuda", generator=g) * 0.05
    gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda", generator=g).to(dt)
    wpt = pack(lib, d, wt, 1, dt)
    res = {}
    for mode, name in ((3 | 64, "igemm3"), (3, "igemm4p")):
        lib.vg_set_conv_kernel(mode)
        dx = torch.full((n, h, w, d.xcs), float("nan"), dtype=dt, device="cuda")
        stats = torch.zeros(L.STATS_REPLICAS, 2, 64, dtype=torch.float64, device="cuda")
        line = []
        for cname, fn in (("dgrad", lambda: lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), None, 0, 0, L.stream())),
                          ("dgrad+stats", lambda: lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), stats.data_ptr(), 0, 0, L.stream()))):
            stats.zero_()
            L.check(fn(), cname)
            torch.cuda.synchronize()
            kern = lib.vg_last_conv_kernel().decode()
            res[(name, cname)] = (dx.clone(), stats.sum(0).clone())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            line.append(f"{cname} {e0.elapsed_time(e1) / iters * 1e3:7.1f} us [{kern}]")
        print(f"{name}: " + " | ".join(line), flush=True)
    lib.vg_set_conv_kernel(3)
    for cname in ("dgrad", "dgrad+stats"):
        a, b = res[("igemm3", cname)], res[("igemm4p", cname)]
        print(cname, "equal bits:", torch.equal(a[0], b[0]), "max|d|", (a[0].float() - b[0].float()).abs().max().item(),
              "stats rel", ((a[1] - b[1]).abs().max() / a[1].abs().max().clamp_min(1e-30)).item())


if __name__ == "__main__":
    main(*(int(v) for v in sys.argv[1:]))
