"""Per shape: does taking a residual BatchNorm's backward sums (and storing the masked gradient) in the epilogue of the
input-gradient launch beat the separate vg_bn_bwd_reduce_dz pass?  The five launches of the VQ step that fuse today
(VG_BNSTAT_FUSION=1): the input gradients of the three transposed convs of the decoder (4x4 stride-2 forward gathers) and
of the encoder's 128 -> 256 and 256 -> 512 downsampling convs (2x2-tap phases).  PYTHONPATH=.:tests python tests/gpu_fused_residual_perf.py"""
import ctypes as C, torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc, pack

lib = L.load()
dt = torch.bfloat16


def timed(fn, n=10):
    for _ in range(2):
        L.check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(kind, n, h, w, ci, co):
    # kind "fwd": the launch is vg_conv_fwd of a 4x4 stride-2 conv ci -> co (input gradient of ConvTranspose2d(co, ci)):
    #             gradient tensor [n, h/2, w/2, co];  kind "dgrad": vg_conv_dgrad of a 4x4 stride-2 conv ci -> co: [n, h, w, ci]
    d = make_desc(n, h, w, ci, co, 4, 2, 1, dt)
    g = torch.Generator(device="cuda").manual_seed(3)
    wt = torch.randn(co, ci, 4, 4, device="cuda", generator=g) * 0.05
    if kind == "fwd":
        wp = pack(lib, d, wt, 0, dt)
        src = torch.randn(n, h, w, d.xcs, device="cuda", generator=g).to(dt)
        shape, c, cs = (n, d.oh, d.ow, d.ycs), co, d.ycs
    else:
        wp = pack(lib, d, wt, 1, dt)
        src = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda", generator=g).to(dt)
        shape, c, cs = (n, h, w, d.xcs), ci, d.xcs
    rows = shape[0] * shape[1] * shape[2]
    xraw = torch.randn(shape, device="cuda", generator=g).to(dt)
    ybn = torch.randn(shape, device="cuda", generator=g).to(dt)
    dx = torch.empty(shape, device="cuda", dtype=dt)
    dz = torch.empty(shape, device="cuda", dtype=dt)
    vec = [torch.rand(c, device="cuda") + 0.5 for _ in range(4)]
    cpad = (c + 127) // 128 * 128 if c > 64 else 64
    sums = torch.zeros(L.STATS_REPLICAS * 2 * cpad, dtype=torch.float64, device="cuda")
    bs = L.BnBwdStats(x=xraw.data_ptr(), y=ybn.data_ptr(), mean=vec[0].data_ptr(), invstd=vec[1].data_ptr(), scale=vec[2].data_ptr(),
                      shift=vec[3].data_ptr(), sums=sums.data_ptr(), cpad=cpad, act=L.ACT_LRELU, mask_out=1)
    fused = C.c_int(-1)
    if kind == "fwd":
        plain = lambda: lib.vg_conv_fwd(C.byref(d), src.data_ptr(), wp.data_ptr(), None, dx.data_ptr(), None, L.ACT_NONE, 0, L.stream())
        fuse = lambda: lib.vg_conv_fwd_bnstats(C.byref(d), src.data_ptr(), wp.data_ptr(), dx.data_ptr(), 0, C.byref(bs), C.byref(fused), L.stream())
    else:
        plain = lambda: lib.vg_conv_dgrad(C.byref(d), src.data_ptr(), wp.data_ptr(), None, dx.data_ptr(), None, L.ACT_NONE, 0, L.stream())
        fuse = lambda: lib.vg_conv_dgrad_bnstats(C.byref(d), src.data_ptr(), wp.data_ptr(), dx.data_ptr(), 0, C.byref(bs), C.byref(fused), L.stream())
    red = lambda: lib.vg_bn_bwd_reduce_dz(dx.data_ptr(), ybn.data_ptr(), xraw.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                          vec[3].data_ptr(), sums.data_ptr(), dz.data_ptr(), rows, cs, c, cpad, L.ACT_LRELU, L.VG_BF16, L.stream())
    both = lambda: (plain(), red())[1]
    tp, tr, tb, tf = timed(plain), timed(red), timed(both), timed(fuse)
    print(f"{kind:5s} {h}x{w} {ci}->{co}: plain {tp:6.1f} us [{lib.vg_last_conv_kernel().decode()}]  + reduce_dz {tr:6.1f} = {tb:6.1f} back to back;"
          f"  fused {tf:6.1f} us (fused={fused.value})  -> fusion {'wins' if tf < tb else 'LOSES'} by {abs(tb - tf):5.1f} us", flush=True)


for _ in range(2):
    run("fwd", 64, 256, 256, 64, 128)
    run("fwd", 64, 128, 128, 128, 256)
    run("fwd", 64, 64, 64, 256, 512)
    run("dgrad", 64, 128, 128, 128, 256)
    run("dgrad", 64, 64, 64, 256, 512)
