"""Where a block's time goes inside igemm4: s_memtime ticks per stage summed over its tiles (ablation build only)."""
import ctypes as C, os, torch
from vae_gan_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libvg_hip_{os.environ.get('VG_LIB_EXP', 'ablate')}.so")      # VG_LIB_EXP=<tag>: another -DVG_ABLATE=1 build
from gpu_ops_check import make_desc, pack

def run(n=64, h=256, w=256):
    lib = L.load(); dt = torch.bfloat16
    lib.vg_dbg_set_stamp_buffer.argtypes = [C.c_void_p]
    d = make_desc(n, h, w, 64, 64, 3, 1, 1, dt)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    wp = pack(lib, d, wt, 0, dt)
    y = torch.empty(n, h, w, d.ycs, dtype=dt, device="cuda")
    stats = torch.zeros(L.STATS_REPLICAS, 2, 64, dtype=torch.float64, device="cuda")
    buf = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
    lib.vg_dbg_set_stamp_buffer(buf.data_ptr())
    import sys
    masks = [int(v) for v in sys.argv[1:]] or [0]
    for dbg, act, st, tag in [(m, a, s_, t) for m in masks for a, s_, t in ((0, None, "plain"), (1, stats, "stats+lrelu"))]:
        lib.vg_set_conv_kernel(3 | ((8 | dbg) << 8))
        tag = f"{tag} dbg={dbg}"
        for _ in range(2):
            buf.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None if st is None else st.data_ptr(), act, 0, L.stream())
            e1.record()
            torch.cuda.synchronize()
        s = buf.view(256, 8).cpu().double()
        names = ["barrier (top)", "K loop", "epilogue", "wait vmcnt"]
        print(f"{tag} [{lib.vg_last_conv_kernel().decode()}] launch {e0.elapsed_time(e1) * 1e3:.0f} us; per block, us (memtime 100 MHz):")
        for i, nm in enumerate(names):
            print(f"   {nm:16s} mean {s[:, i].mean() / 100:8.1f} min {s[:, i].min() / 100:8.1f} max {s[:, i].max() / 100:8.1f}")
        print(f"   block lifetime   mean {s[:, 4].mean() / 100:8.1f} max {s[:, 4].max() / 100:8.1f}; start spread {(s[:, 5].max() - s[:, 5].min()) / 100:.1f}")
    lib.vg_dbg_set_stamp_buffer(None)
    lib.vg_set_conv_kernel(3)

run()
