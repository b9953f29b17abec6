"""Where a tile's time goes inside igemm3: per-block s_memtime stamps (ablation build only)."""
import ctypes as C, os, sys, torch
from vae_gan_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libvg_hip_ablate.so")
from gpu_ops_check import make_desc, pack

def run(n, h, w, ci, co, k, stride, pad):
    lib = L.load(); dt = torch.bfloat16
    lib.vg_dbg_set_stamp_buffer.argtypes = [C.c_void_p]
    d = make_desc(n, h, w, ci, co, k, stride, pad, dt)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    wp = pack(lib, d, wt, 0, dt)
    y = torch.empty(n, d.oh, d.ow, d.ycs, dtype=dt, device="cuda")
    nblk = 1 << 16
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
    lib.vg_dbg_set_stamp_buffer(buf.data_ptr())
    lib.vg_set_conv_kernel(3 | (8 << 8))
    for _ in range(2):
        buf.zero_()
        lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, 0, 0, L.stream())
        torch.cuda.synchronize()
    s = buf.view(nblk, 8).cpu()
    s = s[s[:, 0] > 0].double()
    dts = [(s[:, i + 1] - s[:, i]) for i in range(4)]
    names = ["launch->prologue landed", "main loop", "epilogue (stores issued)", "stats+tail"]
    print(f"{h}x{w} ci{ci} co{co} k{k}s{stride}: blocks {s.shape[0]}, s_memtime ticks:")
    for nm, v in zip(names, dts):
        print(f"   {nm:28s} mean {v.mean():9.1f} median {v.median():9.1f}")
    print(f"   block lifetime (stamp0..4) mean {(s[:,4]-s[:,0]).mean():9.1f}; kernel span {(s[:,4].max()-s[:,0].min()):9.1f}")
    lib.vg_dbg_set_stamp_buffer(None)
    lib.vg_set_conv_kernel(3)

for c in [(64, 64, 64, 256, 256, 3, 1, 1), (64, 32, 32, 512, 512, 3, 1, 1)]:
    run(*c)
