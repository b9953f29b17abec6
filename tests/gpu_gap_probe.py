"""What a tiny dependent kernel really costs between two big ones (not collected by pytest).
Chains conv(+statistics) -> bn_finalize -> bn_act_fwd as the step runs them, un-profiled, and times the chain with and
without its middle launch: the difference is the finalize kernel's marginal cost including any dispatch gap around it
(rocprofv3's kernel trace shows ~5 us gaps there, which may be the profiler's).   PYTHONPATH=.:tests python tests/gpu_gap_probe.py"""
import ctypes as C, torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc, pack

lib = L.load(); dt = torch.bfloat16


def chain(n, h, w, ci, co, iters=60):
    d = make_desc(n, h, w, ci, co, 3, 1, 1, dt)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    wp = pack(lib, d, torch.randn(co, ci, 3, 3, device="cuda") * 0.05, 0, dt)
    y = torch.empty(n, h, w, d.ycs, dtype=dt, device="cuda")
    a = torch.empty_like(y)
    cpad = max(64, (co + 127) // 128 * 128) if co > 64 else 64
    stats = torch.zeros(L.STATS_REPLICAS, 2, cpad, dtype=torch.float64, device="cuda")
    gamma, beta = torch.ones(co, device="cuda"), torch.zeros(co, device="cuda")
    rm, rv = torch.zeros(co, device="cuda"), torch.ones(co, device="cuda")
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    vec = torch.zeros(4, co, device="cuda"); vec[2] = 1.0
    rows = n * h * w
    st = L.stream()
    conv = lambda: L.check(lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), 0, 0, st), "conv")
    fin = lambda: L.check(lib.vg_bn_finalize(stats.data_ptr(), co, cpad, rows, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(),
                                             rv.data_ptr(), nbt.data_ptr(), 1, vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                             vec[3].data_ptr(), st), "fin")
    act = lambda: L.check(lib.vg_bn_act_fwd(y.data_ptr(), None, a.data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), rows, d.ycs, co, 1,
                                            L.vg_dtype(dt), st), "act")

    def timed(fns):
        for f in fns: f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            for f in fns: f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    res = {}
    for rnd in range(2):
        for name, fns in (("conv,fin,act", (conv, fin, act)), ("conv,act", (conv, act)), ("conv", (conv,)), ("act", (act,)),
                          ("fin", (fin,)), ("conv,fin", (conv, fin))):
            res[name] = timed(fns)
    print(f"{ci}->{co} @{h}x{w} [{lib.vg_last_conv_kernel().decode()}] us per iteration: " + ", ".join(f"{k} {v:.1f}" for k, v in res.items()))
    print(f"    finalize between the two: +{res['conv,fin,act'] - res['conv,act']:.1f} us; conv + act alone {res['conv'] + res['act']:.1f} "
          f"against chained {res['conv,act']:.1f}", flush=True)


for c in ((64, 64, 64, 256, 256), (64, 32, 32, 512, 512), (64, 128, 128, 128, 128), (64, 256, 256, 64, 64)):
    chain(*c)
