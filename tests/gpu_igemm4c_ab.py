"""A/B of the channel-split weights-stationary kernel (igemm4c) against igemm3 on the 128 -> 128 3x3 layer (B=64, 128x128)."""
import ctypes as C
import sys
import torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc, pack


def main(n=64, h=128, w=128, iters=10):
    lib = L.load()
    dt = torch.bfloat16
    d = make_desc(n, h, w, 128, 128, 3, 1, 1, dt)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(n, h, w, d.xcs, device="cuda", generator=g).to(dt)
    wt = torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03
    gy = torch.randn(n, h, w, d.ycs, device="cuda", generator=g).to(dt)
    bias = torch.linspace(-1, 1, 128, device="cuda")
    wp, wpt = pack(lib, d, wt, 0, dt), pack(lib, d, wt, 1, dt)
    res = {}
    for mode, name in ((3 | 128, "igemm3"), (3, "igemm4c")):
        lib.vg_set_conv_kernel(mode)
        y = torch.full((n, h, w, d.ycs), float("nan"), dtype=dt, device="cuda")
        dx = torch.full((n, h, w, d.xcs), float("nan"), dtype=dt, device="cuda")
        dxa = torch.empty(n, h, w, d.xcs, dtype=dt, device="cuda")
        stats = torch.zeros(L.STATS_REPLICAS, 2, 128, dtype=torch.float64, device="cuda")
        calls = (("fwd+bias+stats+lrelu", lambda: lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                                  stats.data_ptr(), 1, 0, L.stream())),
                 ("dgrad", lambda: lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), None, 0, 0, L.stream())),
                 ("dgrad+acc", lambda: lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dxa.data_ptr(), None, 0, 1, L.stream())))
        line = []
        for cname, fn in calls:
            stats.zero_()
            dxa.copy_(x)
            L.check(fn(), cname)
            torch.cuda.synchronize()
            kern = lib.vg_last_conv_kernel().decode()
            res[(name, cname)] = (y.clone(), dx.clone(), stats.sum(0).clone(), dxa.clone())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            line.append(f"{cname} {e0.elapsed_time(e1) / iters * 1e3:7.1f} us [{kern}]")
        print(f"{name}: " + " | ".join(line), flush=True)
    lib.vg_set_conv_kernel(3)
    for cname, idx in (("fwd+bias+stats+lrelu", 0), ("dgrad", 1), ("dgrad+acc", 3)):
        a, b = res[("igemm3", cname)], res[("igemm4c", cname)]
        print(cname, "equal bits:", torch.equal(a[idx], b[idx]), "max|d|", (a[idx].float() - b[idx].float()).abs().max().item(),
              "stats rel", ((a[2] - b[2]).abs().max() / a[2].abs().max().clamp_min(1e-30)).item())


if __name__ == "__main__":
    main(*(int(v) for v in sys.argv[1:]))
