"""Manual timing of the codebook search, exact-f32 MFMA (vg_vq_assign) against the shortlist path
(vg_vq_assign_shortlist), at BASELINE configs[4] (65,536 latents, K = 8192, D = 256) and at the default codebook:
PYTHONPATH=.:tests python tests/gpu_vq_perf.py"""
import torch
from vae_gan_amd import lib as L

lib = L.load()


def timeit(fn, iters=10):
    for _ in range(2):
        L.check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rows, k, d in [(65536, 8192, 256), (65536, 512, 64), (65536, 4096, 128)]:
    for kind in ("init", "trained"):
        g = torch.Generator(device="cuda").manual_seed(1)
        if kind == "init":
            cb = (torch.rand(k, d, device="cuda", generator=g) * 2 - 1) / k
            z = torch.randn(rows, d, device="cuda", generator=g)
        else:
            cb = torch.randn(k, d, device="cuda", generator=g) * 0.5
            z = cb[torch.randint(0, k, (rows,), device="cuda", generator=g)] + 0.3 * torch.randn(rows, d, device="cuda", generator=g)
        idx = torch.empty(rows, dtype=torch.int64, device="cuda")
        norms = torch.empty(k, device="cuda")
        ws = torch.empty(lib.vg_vq_shortlist_ws_bytes(rows, k, d), dtype=torch.uint8, device="cuda")
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        t0 = timeit(lambda: lib.vg_vq_assign(z.data_ptr(), cb.data_ptr(), rows, k, d, idx.data_ptr(), norms.data_ptr(), L.stream()))
        t1 = timeit(lambda: lib.vg_vq_assign_shortlist(z.data_ptr(), cb.data_ptr(), rows, k, d, idx.data_ptr(), ws.data_ptr(),
                                                       flag.data_ptr(), L.stream()))
        # candidates per row: the count array of the shortlist workspace (layout of vq_short_layout in csrc/vq.hip)
        up = lambda v: (v + 255) & ~255
        off = up(k * 4) + up((k + 31) // 32 * 32 * 4) + up(rows * 4)
        cnt = ws[off:off + rows * 4].view(torch.int32).float()
        print(f"rows {rows} K {k} D {d} {kind:8s}: exact {t0:8.1f} us   shortlist {t1:8.1f} us   (fallback {int(flag.item())}; "
              f"candidates per row: mean {cnt.mean().item():.2f}, max {int(cnt.max().item())})", flush=True)
