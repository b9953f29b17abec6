"""Forward launches of two 3x3 layers for a counter pass (VG_LIB_EXP picks the library):
   rocprofv3 --pmc ... --kernel-trace --output-format csv -d out -- python3 tests/gpu_pmc_probe.py"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_gan_amd import lib as L
if os.environ.get("VG_LIB_EXP"):
    _v = os.environ["VG_LIB_EXP"]
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libvg_hip_{'exp' if _v == '1' else _v}.so")
from gpu_ops_check import make_desc, pack

lib = L.load()
dt = torch.bfloat16
shapes = [(64, 64, 64, 256, 256, 3, 1, 1), (64, 32, 32, 512, 512, 3, 1, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (n, h, w, ci, co, k, stride, pad) in shapes:
    d = make_desc(n, h, w, ci, co, k, stride, pad, dt)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    wp = pack(lib, d, wt, 0, dt)
    y = torch.empty(n, d.oh, d.ow, d.ycs, dtype=dt, device="cuda")
    for _ in range(12):
        lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, 0, 0, L.stream())
    torch.cuda.synchronize()
