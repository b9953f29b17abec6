"""Cross-check of the conv kernel generations (im2col-row / spatial-tile / ping-pong) against each other
on the same inputs through the C ABI: fwd and dgrad, both dtypes, many shapes."""
import ctypes as C, itertools, sys, torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc, pack, rel

def run(lib, d, x, wp, wpt, gy, mode, dtype):
    lib.vg_set_conv_kernel(mode)
    y = torch.full((d.n, d.oh, d.ow, d.ycs), float("nan"), dtype=dtype, device="cuda")
    L.check(lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, 0, 0, L.stream()), "fwd")
    dx = torch.full((d.n, d.h, d.w, d.xcs), float("nan"), dtype=dtype, device="cuda")
    L.check(lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), None, 0, 0, L.stream()), "dgrad")
    # epilogue variants: bias + statistics + LeakyReLU forward; accumulate into an existing gradient
    npad = 32 if d.co <= 32 else (64 if d.co <= 64 else (d.co + 127) // 128 * 128)
    stats = torch.zeros(L.STATS_REPLICAS, 2, npad, dtype=torch.float64, device="cuda")
    bias = torch.linspace(-1, 1, d.co, device="cuda")
    y2 = torch.full((d.n, d.oh, d.ow, d.ycs), float("nan"), dtype=dtype, device="cuda")
    L.check(lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y2.data_ptr(), stats.data_ptr(), 1, 0,
                            L.stream()), "fwd2")
    dx2 = (torch.arange(d.n * d.h * d.w * d.xcs, device="cuda").reshape(d.n, d.h, d.w, d.xcs) % 7).to(dtype)
    dx2[..., d.ci:] = 0
    L.check(lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx2.data_ptr(), None, 0, 1, L.stream()), "dgrad2")
    torch.cuda.synchronize()
    return y.float(), dx.float(), y2.float(), dx2.float(), stats.sum(0).float()


def run_wgrad(lib, d, x, gy, mode):
    lib.vg_set_wgrad_kernel(mode)
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    dw = torch.full((d.co, d.ci, d.r, d.s), float("nan"), dtype=torch.float32, device="cuda")
    L.check(lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream()), "wgrad")
    torch.cuda.synchronize()
    return dw

def main(exit=True):
    lib = L.load(); bad = 0
    shapes = []
    for hw in (8, 16, 17, 32, 33, 64):
        for ci, co in ((64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (96, 64), (3, 64), (64, 3), (256, 512)):
            for k, s, pd in ((3, 1, 1), (4, 2, 1), (4, 1, 1), (1, 1, 0)):
                if hw >= 64 and ci * co > 128 * 256: continue
                shapes.append((2, hw, hw + (hw % 2) * 3, ci, co, k, s, pd))
    for dtype, tol in ((torch.float32, 3e-6), (torch.bfloat16, 1e-2)):
        for shp in shapes:
            n, h, w, ci, co, k, s, pd = shp
            if (h + 2 * pd - k) // s + 1 < 1 or (s == 2 and (h % 2 or w % 2)): continue
            d = make_desc(n, h, w, ci, co, k, s, pd, dtype)
            g = torch.Generator(device="cuda").manual_seed(1)
            x = torch.randn(n, h, w, d.xcs, device="cuda", generator=g).to(dtype); x[..., ci:] = 0
            wt = torch.randn(co, ci, k, k, device="cuda", generator=g) / (ci * k * k) ** 0.5
            gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda", generator=g).to(dtype); gy[..., co:] = 0
            wp, wpt = pack(lib, d, wt, 0, dtype), pack(lib, d, wt, 1, dtype)
            ref = run(lib, d, x, wp, wpt, gy, 1, dtype)
            for mode in (2, 3):
                out = run(lib, d, x, wp, wpt, gy, mode, dtype)
                for nm, a, b in (("fwd", out[0], ref[0]), ("dgrad", out[1], ref[1]), ("fwd+bias+lrelu", out[2], ref[2]),
                                 ("dgrad+acc", out[3], ref[3]), ("stats", out[4], ref[4])):
                    e = rel(a, b)
                    if not e <= tol:
                        print(f"MISMATCH v{mode} {nm} {shp} {str(dtype)[6:]} rel={e:.3e}", flush=True); bad += 1
            ref_w = run_wgrad(lib, d, x, gy, 1)
            for wmode in (3, 4):                 # 64 x 64 slabs; 128 x 64 slabs where there are 128 output channels
                e = rel(run_wgrad(lib, d, x, gy, wmode), ref_w)
                if not e <= 2e-5:
                    print(f"MISMATCH wgrad v{wmode} {shp} {str(dtype)[6:]} rel={e:.3e}", flush=True); bad += 1
    lib.vg_set_conv_kernel(3)
    lib.vg_set_wgrad_kernel(4)
    print("mismatches:", bad)
    if exit:
        sys.exit(1 if bad else 0)
    return bad


if __name__ == "__main__":
    main()
