"""Timing ablations of the weights-stationary conv kernel (ablation build; results are wrong by construction)."""
import ctypes as C, os, torch
from vae_gan_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libvg_hip_ablate.so")   # make -C vae_gan_amd/csrc ablate
from gpu_ops_check import make_desc, pack

def run(n=64, h=256, w=256, iters=10):
    lib = L.load(); dt = torch.bfloat16
    d = make_desc(n, h, w, 64, 64, 3, 1, 1, dt)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    wp = pack(lib, d, wt, 0, dt)
    y = torch.empty(n, h, w, d.ycs, dtype=dt, device="cuda")
    stats = torch.zeros(L.STATS_REPLICAS, 2, 64, dtype=torch.float64, device="cuda")
    for act, st, tag in ((0, None, "plain"), (1, stats, "stats+lrelu")):
        res = []
        for dbg, name in ((0, "full"), (1, "noDMA"), (16, "noMFMA"), (32, "noStore"), (64, "noEpi"), (1 | 64, "noDMA+noEpi"),
                          (16 | 64, "DMAonly"), (1 | 16, "epiOnly"), (1 | 16 | 64, "empty")):
            lib.vg_set_conv_kernel(3 | (dbg << 8))
            fn = lambda: lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None if st is None else st.data_ptr(), act, 0, L.stream())
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(f"{name} {e0.elapsed_time(e1) / iters * 1e3:.0f}")
        print(f"{tag} [{lib.vg_last_conv_kernel().decode()}] us: " + " | ".join(res), flush=True)
    lib.vg_set_conv_kernel(3)

run()
