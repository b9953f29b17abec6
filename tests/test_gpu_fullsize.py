"""GPU tier: the hot path at BASELINE.json's FULL sizes (batch 64 at 256x256, K=8192/D=256 codebooks, batch 32 beta,
batch 16 hierarchical), where the CPU oracle would take minutes to hours.  Checked through size-independent
properties: adjointness of the three conv primitives (<conv(x), dy> = <x, dgrad(dy)> = <W, wgrad(x, dy)>), nearest-code
optimality / idempotence / bookkeeping of the quantiser against the oracle's distance expression, and whole steps
that must stay finite and book the reference's BatchNorm counters."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import ops
    ops.set_compute_dtype(torch.bfloat16)
    yield ops
    ops.set_compute_dtype(torch.bfloat16)


# VQ-VAE-GAN layer shapes at batch 64, 256x256 (BASELINE.json configs[1]): n, h, w, ci, co, k, stride, pad
FULL_SHAPES = [(64, 256, 256, 3, 64, 3, 1, 1), (64, 256, 256, 64, 64, 3, 1, 1), (64, 256, 256, 64, 128, 4, 2, 1),
               (64, 128, 128, 128, 128, 3, 1, 1), (64, 128, 128, 128, 256, 4, 2, 1), (64, 64, 64, 256, 256, 3, 1, 1),
               (64, 64, 64, 256, 512, 4, 2, 1), (64, 32, 32, 512, 512, 3, 1, 1), (64, 32, 32, 512, 64, 1, 1, 0),
               (64, 256, 256, 64, 3, 3, 1, 1), (64, 256, 256, 3, 64, 4, 2, 1), (64, 16, 16, 512, 1, 4, 1, 1)]


@pytest.mark.parametrize("shape", FULL_SHAPES, ids=lambda c: "n%d_%dx%d_ci%d_co%d_k%ds%dp%d" % c)
def test_conv_adjointness_full_size(gpu, shape):
    """The input-gradient and weight-gradient kernels are the exact adjoints of the forward kernel:
    <conv(x; W), dy> = <x, dgrad(dy; W)> = <W, wgrad(x, dy)>.  bf16 storage, f32 accumulation: the three numbers
    agree to 2e-3 of |conv(x)|.|dy| (each output is rounded to bf16 once, 2^-9 relative, with random signs)."""
    from vae_gan_amd import lib as L
    from gpu_ops_check import make_desc, pack
    n, h, w, ci, co, k, stride, pad = shape
    lib, dt = L.load(), torch.bfloat16
    d = make_desc(n, h, w, ci, co, k, stride, pad, dt)
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(n, h, w, d.xcs, device="cuda", generator=g).to(dt)
    x[..., ci:] = 0
    dy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda", generator=g).to(dt)
    dy[..., co:] = 0
    wt = (torch.randn(co, ci, k, k, device="cuda", generator=g) / (ci * k * k) ** 0.5).to(dt).float()   # exactly representable
    wp, wpt = pack(lib, d, wt, 0, dt), pack(lib, d, wt, 1, dt)
    y = torch.empty(n, d.oh, d.ow, d.ycs, dtype=dt, device="cuda")
    dx = torch.empty(n, h, w, d.xcs, dtype=dt, device="cuda")
    dw = torch.empty(co, ci, k, k, device="cuda")
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    L.check(lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, 0, 0, L.stream()), "fwd")
    L.check(lib.vg_conv_dgrad(C.byref(d), dy.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), None, 0, 0, L.stream()), "dgrad")
    L.check(lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream()), "wgrad")
    torch.cuda.synchronize()
    a = float((y.double() * dy.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((wt.double() * dw.double()).sum())
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(a - b) <= 2e-3 * scale and abs(a - c) <= 2e-3 * scale, (a, b, c, scale)
    assert float(y[..., co:].abs().sum()) == 0.0 and float(dx[..., ci:].abs().sum()) == 0.0      # padding lanes stay zero
    # linearity in the input: conv(2x) = 2 conv(x) bit for bit (a power-of-two scale commutes with every rounding)
    y2 = torch.empty_like(y)
    x2 = (x * 2).contiguous()
    L.check(lib.vg_conv_fwd(C.byref(d), x2.data_ptr(), wp.data_ptr(), None, y2.data_ptr(), None, 0, 0, L.stream()), "fwd2")
    torch.cuda.synchronize()
    assert torch.equal(y2, y * 2)


@pytest.mark.parametrize("rows,K,D", [(64 * 32 * 32, 512, 64), (2 * 64 * 64, 8192, 256)], ids=["cfg2_N65536_K512", "cfg5_N8192_K8192"])
def test_quantizer_full_size(gpu, rows, K, D):
    """configs[1] (batch 64: N = 65,536 latents, K=512, D=64) and configs[4] (512x512 crops: 4,096 latents per image,
    K=8192, D=256).  Indices against the oracle's fp32 distance expression (vq_vae_gan/model.py:88-95) with the
    documented near-tie rule; gather, loss, EMA bookkeeping and idempotence as properties."""
    from oracle import vaegan_oracle as O
    from vae_gan_amd.vq_vae_gan.model import VectorQuantizer
    side = int((rows // 2) ** 0.5) if K == 8192 else 32
    b = rows // (side * side)
    g = torch.Generator().manual_seed(3)
    z = 0.5 * torch.randn(b, D, side, side, generator=g)
    vq = VectorQuantizer(K, D, 0.25, 0.99)
    vq.embeddings.weight.data = 0.5 * torch.randn(K, D, generator=g)
    vq.ema_w.copy_(vq.embeddings.weight.data)
    vq.ema_cluster_size.fill_(1.0)
    vq.cuda()
    E0 = vq.embeddings.weight.data.clone()
    zc = z.cuda().requires_grad_(True)
    zq, loss, idx = vq(zc, training=True)
    torch.cuda.synchronize()
    flat = z.permute(0, 2, 3, 1).reshape(-1, D)
    dist = O.vq_distances(flat, E0.cpu())                                  # [N, K] fp32, the reference's op order
    ref = dist.argmin(1)
    mine = idx.reshape(-1).cpu()
    bad = (mine != ref).nonzero().flatten()
    if bad.numel():                                                         # admissible only as fp32 near-ties
        d_ref, d_mine = dist[bad, ref[bad]], dist[bad, mine[bad]]
        ulp = torch.abs(d_ref) * 2.0 ** -23
        assert bool(((d_mine - d_ref).abs() <= 4 * ulp).all()), (bad.numel(), float((d_mine - d_ref).abs().max()))
    assert bad.numel() <= 1e-4 * rows
    q = E0[idx.reshape(-1)].reshape(b, side, side, D).permute(0, 3, 1, 2)
    assert torch.allclose(zq.detach(), zc.detach() + (q - zc.detach()), atol=0, rtol=0)       # straight-through value
    want_loss = 1.25 * float(((q - zc.detach()).double() ** 2).mean())
    assert abs(float(loss.detach()) - want_loss) <= 1e-5 * want_loss
    counts = torch.bincount(mine, minlength=K).float()
    assert torch.allclose(vq.ema_cluster_size.cpu(), 0.99 * torch.ones(K) + 0.01 * counts, rtol=1e-6, atol=1e-6)
    assert abs(float(vq.ema_cluster_size.sum()) - (0.99 * K + 0.01 * rows)) <= 1e-3 * K
    # idempotence: code vectors quantise to themselves with zero loss (against the codebook they came from)
    vq.embeddings.weight.data.copy_(E0)
    zq2, loss2, idx2 = vq(q.contiguous(), training=False)
    assert torch.equal(idx2, idx) and float(loss2) <= 1e-10
    assert torch.equal(zq2, q)


def _finite(d):
    return all(bool(torch.isfinite(torch.as_tensor(v)).all()) for v in d.values())


@pytest.mark.parametrize("arch,batch,size,kw", [("vq", 64, 256, {}), ("vq", 8, 512, {"num_embeddings": 8192, "embedding_dim": 256}),
                                                ("beta", 32, 256, {}), ("hier", 16, 256, {})],
                         ids=["cfg2_vq_b64", "cfg5_vq_k8192_512px", "cfg3_beta_b32_per_gpu", "cfg4_hier_b16_per_gpu"])
def test_full_size_steps(gpu, arch, batch, size, kw):
    """Two whole train steps at the per-GPU sizes of BASELINE.json configs[1], [4], [2], [3] in bf16: losses stay
    finite, the reconstruction is a sigmoid output, BatchNorm counters advance like the reference's (+2 generator
    (+1 for beta), +3 discriminator per iteration), parameters move and stay finite."""
    from vae_gan_amd import train as T
    torch.manual_seed(0)
    if arch == "vq":
        from vae_gan_amd.vq_vae_gan.config import Config
        from vae_gan_amd.vq_vae_gan.model import VQVAEGAN as M
        step, gen_inc = T.vq_train_step, 2
    elif arch == "beta":
        from vae_gan_amd.beta_vae_gans.config import Config
        from vae_gan_amd.beta_vae_gans.model import BetaVAEGAN as M
        step, gen_inc = T.beta_train_step, 1
    else:
        from vae_gan_amd.hierarchical_vae_gan.config import Config
        from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalVAEGAN as M
        step, gen_inc = T.hier_train_step, 2
    cfg = Config()
    cfg.recon_target_policy, cfg.hier_decoder_policy = "area", "nearest"
    for k, v in kw.items():
        setattr(cfg, k, v)
    model = M(cfg).cuda().train()
    opt_g, opt_d, _, _ = T.build_optimizers(model, cfg)
    before = opt_g.flat_p.clone()
    images = torch.rand(batch, 3, size, size, device="cuda")
    for it in range(2):
        out = step(model, images, opt_g, opt_d)
        torch.cuda.synchronize()
        assert _finite({k: v.detach() if torch.is_tensor(v) else v for k, v in out[0].items()}), out[0]
        assert _finite({k: v.detach() if torch.is_tensor(v) else v for k, v in out[1].items()}), out[1]
    sd = model.state_dict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            want = 2 * (3 if k.startswith("discriminator") else gen_inc)
            assert int(v) == want, (k, int(v), want)
        else:
            assert bool(torch.isfinite(v).all()), k
    assert float((opt_g.flat_p - before).abs().max()) > 0
    if arch == "vq":
        assert out[2].dtype == torch.int64 and int(out[2].max()) < cfg.num_embeddings and int(out[2].min()) >= 0
        assert out[2].shape == (batch, size // 8, size // 8)
