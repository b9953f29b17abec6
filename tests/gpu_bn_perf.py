"""Manual perf script (not collected by pytest): HBM rate of the BatchNorm kernels through the C ABI at the layer
sizes of the VQ step (batch 64).  python tests/gpu_bn_perf.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vae_gan_amd.lib as L
if os.environ.get("VG_LIB_EXP"):      # A/B against another build: libvg_hip_<tag>.so
    _v = os.environ["VG_LIB_EXP"]
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libvg_hip_{'exp' if _v == '1' else _v}.so")

lib = L.load()
dev = "cuda"
R = L.STATS_REPLICAS


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for c, hw in ((64, 256), (128, 128), (256, 64), (512, 32)):
    rows = 64 * hw * hw
    cs = cpad = c
    x = torch.randn(rows, cs, device=dev).bfloat16()
    gy = torch.randn(rows, cs, device=dev).bfloat16()
    y = torch.empty_like(x); dx = torch.empty_like(x); dres = torch.empty_like(x)
    vec = torch.rand(4, c, device=dev) + 0.5
    gamma = torch.ones(c, device=dev)
    sums = torch.zeros(R * 2 * cpad * 2 + 3 * cpad, device=dev, dtype=torch.float64)
    dg = torch.zeros(2, c, device=dev); bsum = torch.zeros(c, device=dev)
    E = rows * cs * 2
    st = L.stream()
    t = timed(lambda: L.check(lib.vg_bn_act_fwd(x.data_ptr(), None, y.data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), rows, cs, c, 1, 1, st)))
    print(f"c={c:4d} bn_act_fwd        {t*1e3:7.1f} us  {2*E/t/1e9:6.2f} TB/s")
    t = timed(lambda: L.check(lib.vg_bn_act_fwd(x.data_ptr(), gy.data_ptr(), y.data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), rows, cs, c, 1, 1, st)))
    print(f"c={c:4d} bn_act_fwd+res    {t*1e3:7.1f} us  {3*E/t/1e9:6.2f} TB/s")
    t = timed(lambda: L.check(lib.vg_bn_bwd_reduce(gy.data_ptr(), None, x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), sums.data_ptr(), rows, cs, c, cpad, 1, 1, st)))
    print(f"c={c:4d} bn_bwd_reduce(x)  {t*1e3:7.1f} us  {2*E/t/1e9:6.2f} TB/s")
    t = timed(lambda: L.check(lib.vg_bn_bwd_reduce(gy.data_ptr(), y.data_ptr(), x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), sums.data_ptr(), rows, cs, c, cpad, 1, 1, st)))
    print(f"c={c:4d} bn_bwd_reduce(y)  {t*1e3:7.1f} us  {3*E/t/1e9:6.2f} TB/s")
    t = timed(lambda: L.check(lib.vg_bn_bwd_apply(gy.data_ptr(), None, x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), gamma.data_ptr(), sums.data_ptr(), dx.data_ptr(), None, dg[0].data_ptr(), dg[1].data_ptr(), 1, bsum.data_ptr(), rows, rows, 1.0, cs, c, cpad, 1, 1, st)))
    print(f"c={c:4d} bn_bwd_apply+bias {t*1e3:7.1f} us  {3*E/t/1e9:6.2f} TB/s")
    t = timed(lambda: L.check(lib.vg_bn_bwd_apply(gy.data_ptr(), None, x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), gamma.data_ptr(), sums.data_ptr(), dx.data_ptr(), None, dg[0].data_ptr(), dg[1].data_ptr(), 1, None, rows, rows, 1.0, cs, c, cpad, 1, 1, st)))
    print(f"c={c:4d} bn_bwd_apply      {t*1e3:7.1f} us  {3*E/t/1e9:6.2f} TB/s")
    t = timed(lambda: L.check(lib.vg_bn_bwd_apply(gy.data_ptr(), y.data_ptr(), x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), gamma.data_ptr(), sums.data_ptr(), dx.data_ptr(), dres.data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(), 1, bsum.data_ptr(), rows, rows, 1.0, cs, c, cpad, 1, 1, st)))
    print(f"c={c:4d} bn_bwd_apply+dres {t*1e3:7.1f} us  {5*E/t/1e9:6.2f} TB/s")
    t = timed(lambda: L.check(lib.vg_act_bwd(gy.data_ptr(), y.data_ptr(), dx.data_ptr(), rows * cs, 1, 1, st)))
    print(f"c={c:4d} act_bwd           {t*1e3:7.1f} us  {3*E/t/1e9:6.2f} TB/s")
    del x, gy, y, dx, dres
