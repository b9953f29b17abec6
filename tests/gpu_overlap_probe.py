"""Manual probe (not collected by pytest): do an MFMA-bound kernel and an HBM-bound kernel overlap when they are put on
two HIP streams?  Pairs the weight gradient of a layer (vg_conv_wgrad) with the BatchNorm backward passes of the layer
in front of it (vg_bn_bwd_reduce + vg_bn_bwd_apply), which do not depend on each other in the backward pass.
PYTHONPATH=.:tests python tests/gpu_overlap_probe.py"""
import ctypes as C, torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc

lib = L.load()
R = L.STATS_REPLICAS


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def probe(n, hw, c, other="wgrad"):
    dt = torch.bfloat16
    d = make_desc(n, hw, hw, c, c, 3, 1, 1, dt)
    x = torch.randn(n, hw, hw, d.xcs, device="cuda").to(dt)
    gy = torch.randn(n, hw, hw, d.ycs, device="cuda").to(dt)
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    dw = torch.empty(c, c, 3, 3, dtype=torch.float32, device="cuda")
    wp = None
    rows = n * hw * hw
    bx = torch.randn(rows, c, device="cuda").to(dt)
    bg = torch.randn(rows, c, device="cuda").to(dt)
    bdx = torch.empty_like(bx)
    vec = torch.rand(4, c, device="cuda") + 0.5
    gamma = torch.ones(c, device="cuda")
    sums = torch.zeros(R * 2 * c * 2 + 3 * c, device="cuda", dtype=torch.float64)
    dg = torch.zeros(2, c, device="cuda")
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def mfma(st):
        L.check(lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, st))

    def hbm(st):
        L.check(lib.vg_bn_bwd_reduce(bg.data_ptr(), None, bx.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                     vec[3].data_ptr(), sums.data_ptr(), rows, c, c, c, 1, 1, st))
        L.check(lib.vg_bn_bwd_apply(bg.data_ptr(), None, bx.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                    vec[3].data_ptr(), gamma.data_ptr(), sums.data_ptr(), bdx.data_ptr(), None, dg[0].data_ptr(),
                                    dg[1].data_ptr(), 1, None, rows, rows, 1.0, c, c, c, 1, 1, st))

    ta = timed(lambda: mfma(main.cuda_stream))
    tb = timed(lambda: hbm(main.cuda_stream))
    ts = timed(lambda: (mfma(main.cuda_stream), hbm(main.cuda_stream)))

    def both():
        side.wait_stream(main)
        mfma(side.cuda_stream)
        hbm(main.cuda_stream)
        main.wait_stream(side)

    tc = timed(both)

    def both_rev():
        side.wait_stream(main)
        hbm(side.cuda_stream)
        mfma(main.cuda_stream)
        main.wait_stream(side)

    tr = timed(both_rev)
    print(f"c={c:4d} hw={hw:4d}: wgrad {ta*1e3:6.1f} us, bn bwd passes {tb*1e3:6.1f} us, serial {ts*1e3:6.1f} us, "
          f"two streams {tc*1e3:6.1f} us (bn on side: {tr*1e3:6.1f}) -> hidden {(ts-min(tc,tr))*1e3:6.1f} us "
          f"of {min(ta,tb)*1e3:6.1f}", flush=True)


for c, hw in ((64, 256), (128, 128), (256, 64), (512, 32)):
    probe(64, hw, c)
