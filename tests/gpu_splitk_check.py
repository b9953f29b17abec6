"""Split-K forward against the single-kernel forward on PatchGAN-tail shapes (bf16): python tests/gpu_splitk_check.py"""
import ctypes as C
import torch
from vae_gan_amd import lib as L
from gpu_ops_check import make_desc, pack

lib = L.load()
dt = torch.bfloat16
for (n, h, w, ci, co, k, s, p) in [(2, 16, 16, 512, 1, 4, 1, 1), (64, 16, 16, 512, 1, 4, 1, 1), (2, 8, 8, 512, 1, 4, 1, 1), (3, 16, 16, 256, 1, 4, 1, 1)]:
    d = make_desc(n, h, w, ci, co, k, s, p, dt)
    torch.manual_seed(1)
    x = torch.randn(n, h, w, d.xcs, device="cuda").to(dt)
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.02
    bias = torch.randn(co, device="cuda")
    wp = pack(lib, d, wt, 0, dt)
    y0 = torch.full((n, d.oh, d.ow, d.ycs), float("nan"), dtype=dt, device="cuda")
    y1 = torch.full((n, d.oh, d.ow, d.ycs), float("nan"), dtype=dt, device="cuda")
    L.check(lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y0.data_ptr(), None, 0, 0, L.stream()), "fwd")
    k0 = lib.vg_last_conv_kernel()
    ws = torch.empty(lib.vg_conv_splitk_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    L.check(lib.vg_conv_fwd_splitk(C.byref(d), x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y1.data_ptr(), 0, 0, ws.data_ptr(), L.stream()), "splitk")
    k1 = lib.vg_last_conv_kernel()
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x[..., :ci].permute(0, 3, 1, 2).double().cpu(), wt.to(dt).double().cpu(), bias.double().cpu(), stride=s, padding=p)
    a, b = y0[..., :co].permute(0, 3, 1, 2).double().cpu(), y1[..., :co].permute(0, 3, 1, 2).double().cpu()
    print((n, h, w, ci, co), k0, k1, "fwd vs ref", float((a - ref).abs().max()), "splitk vs ref", float((b - ref).abs().max()),
          "pad zero", float(y1[..., co:].float().abs().max()), "ref max", float(ref.abs().max()))
