import os, sys
sys.path.insert(0, os.getcwd())
import torch
from vae_gan_amd import lib as L
if os.environ.get('VG_LIB_EXP'):
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libvg_hip_{os.environ['VG_LIB_EXP']}.so")
lib = L.load()
N, K, D = 65536, 512, 64
z = torch.randn(N, D, device="cuda"); cb = torch.randn(K, D, device="cuda") * 0.5
for name, idx in (("uniform", torch.randint(0, K, (N,), device="cuda")), ("skewed", (torch.rand(N, device="cuda") ** 6 * K).long().clamp(max=K-1))):
    zq = torch.empty_like(z); sse = torch.zeros(1, dtype=torch.float64, device="cuda")
    cnt = torch.zeros(K, device="cuda"); es = torch.zeros(K, D, device="cuda")
    fn = lambda: L.check(lib.vg_vq_gather(z.data_ptr(), cb.data_ptr(), idx.data_ptr(), N, K, D, zq.data_ptr(), sse.data_ptr(), cnt.data_ptr(), es.data_ptr(), L.stream()))
    fn(); torch.cuda.synchronize()
    ref = torch.zeros(K, D, device="cuda").index_add_(0, idx, z)
    print(name, "esum rel err", float((es - ref).norm() / ref.norm()), "counts ok", bool(torch.equal(cnt, torch.bincount(idx, minlength=K).float())))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1) / 10 * 1e3, "us")
