"""GPU tier: hierarchical VAE-GAN blocks (SelfAttention 128/256, encoder, discriminator, each decoder sub-block,
output conv) against goldens captured from hierarchical_vae_gan/model.py; the composed decoder raises like the
reference, the documented 'nearest' composition (shim D2) matches the oracle's restatement of it."""
import pytest
import torch

from conftest import load_golden, as_t, close_l2, check_digests

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import ops
    ops.set_compute_dtype(torch.float32)
    yield ops
    ops.set_compute_dtype(torch.bfloat16)


class _Wrap(torch.nn.Module):
    def __init__(self, inner, fn):
        super().__init__()
        self.inner, self.fn = inner, fn

    def forward(self, x):
        return self.fn(self.inner, x)


def _check(name, module, params_of, fill_keys=None, key_prefix=""):
    """params_of: module whose named_parameters() carry the golden's gradient names (after key_prefix)."""
    from vae_gan_amd.testing import fill_state_dict
    g = load_golden(name)
    module.cuda().train()
    x = as_t(g["x"]).cuda().requires_grad_(True)
    y = module(x)
    ys = []
    for t in (y if isinstance(y, (tuple, list)) else [y]):
        ys.extend(t if isinstance(t, (tuple, list)) else [t])
    sum((t * as_t(g[f"cot{i}"]).cuda()).sum() for i, t in enumerate(ys)).backward()
    torch.cuda.synchronize()
    for i, t in enumerate(ys):
        close_l2(t, g[f"y{i}"], g[f"noise:y{i}"], what=f"{name}.y{i}")
    close_l2(x.grad, g["dx"], g["noise:dx"], what=name + ".dx")
    check_digests(g, {key_prefix + k: p.grad for k, p in params_of.named_parameters()})


def _filled(m):
    from vae_gan_amd.testing import fill_state_dict
    m.load_state_dict(fill_state_dict(m.state_dict()))
    return m


@pytest.mark.parametrize("c", [128, 256])
def test_self_attention(vg, c):
    from vae_gan_amd.hierarchical_vae_gan.model import SelfAttention
    sa = _filled(SelfAttention(c))
    _check(f"hier_attention{c}", _Wrap(sa, lambda m, x: vg.to_nchw(m(vg.to_nhwc(x)), c)), sa)


def test_hier_encoder(vg):
    from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalEncoder
    enc = _filled(HierarchicalEncoder(3, 64, [32, 64, 128], 3))
    _check("hier_encoder", _Wrap(enc, lambda m, x: tuple(a for pair in zip(*m(x)) for a in pair) if False else m(x)), enc)


def test_hier_discriminator(vg):
    from vae_gan_amd.hierarchical_vae_gan.model import Discriminator
    d = _filled(Discriminator(3, 64, 3))
    _check("hier_discriminator", d, d)


@pytest.mark.parametrize("j", [0, 1, 2])
def test_hier_decoder_block(vg, j):
    from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalDecoder
    dec = _filled(HierarchicalDecoder(3, 64, [32, 64, 128], 3))
    _check(f"hier_dec_block{j}", _Wrap(dec, lambda m, x: m.run_block(j, x)), dec.decoders[j], key_prefix=f"decoders.{j}.")


def test_hier_output_conv(vg):
    from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalDecoder
    dec = _filled(HierarchicalDecoder(3, 64, [32, 64, 128], 3))
    _check("hier_output_conv", _Wrap(dec, lambda m, x: m.run_output_conv(x)), dec.output_conv, key_prefix="output_conv.")


def test_composed_decoder_raises_like_reference_and_nearest_matches_oracle(vg):
    from oracle import vaegan_oracle as O
    from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalDecoder
    from vae_gan_amd.testing import fill_state_dict, synthetic_eps
    dec = HierarchicalDecoder(3, 64, [32, 64, 128], 3)
    sd = fill_state_dict(dec.state_dict())
    dec.load_state_dict(sd)
    dec.cuda().train()
    zs = [0.5 * synthetic_eps((2, 32, 32, 32), 1), 0.5 * synthetic_eps((2, 64, 16, 16), 2), 0.5 * synthetic_eps((2, 128, 8, 8), 3)]
    with pytest.raises(RuntimeError, match="channels"):
        dec([z.cuda() for z in zs])
    dec.policy = "nearest"
    y = dec([z.cuda() for z in zs])
    torch.cuda.synchronize()
    ref = O.hier_decoder_nearest({"decoder." + k: v.double() if v.dtype.is_floating_point else v.clone() for k, v in sd.items()},
                                 [z.double() for z in zs])
    close_l2(y, ref, rtol=1e-4, what="nearest composition")


@pytest.mark.parametrize("share", [True, False])
def test_hier_full_step_matches_reference(vg, share):
    """Two iterations of hierarchical_vae_gan/main.py:75-111 against the golden made by running the REFERENCE
    modules + loss_function with shims D2 (nearest composition) + D1 (area target); both the shared-encoder
    schedule (default) and the literal two-forward schedule."""
    from conftest import check_step, check_state
    from vae_gan_amd.hierarchical_vae_gan.config import Config
    from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalVAEGAN
    from vae_gan_amd.testing import fill_state_dict, hier_step_eps
    from vae_gan_amd.train import build_optimizers, hier_train_step
    g = load_golden("hier_step_128")
    cfg = Config()
    cfg.recon_target_policy, cfg.hier_decoder_policy = "area", "nearest"
    model = HierarchicalVAEGAN(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.cuda().train()
    opt_g, opt_d, _, _ = build_optimizers(model, cfg)
    images = as_t(g["images"]).cuda()
    eps = hier_step_eps(2, 128)
    for it in range(2):
        pair = [[e.cuda() for e in fwd] for fwd in eps[it]]
        dl, gl, recon = hier_train_step(model, images, opt_g, opt_d, eps_lists=pair, elide_d_wgrad=(it == 1),
                                        share_encoder=share, return_recon=True)
        torch.cuda.synchronize()
        r = {"d_loss": float(dl["total"].detach()), "d_kl": float(dl["kl"]), "g_loss": float(gl["total"].detach()),
             "recon": float(gl["recon"]), "kl": float(gl["kl"]), "adv": float(gl["adv"]), "recon_img": recon.detach().cpu()}
        check_step(g, it, r, ("d_loss", "d_kl", "g_loss", "recon", "kl", "adv"))
    check_state(g, {k: v.detach().cpu() for k, v in model.state_dict().items()}, max(float(g["noise:it1_recon_img"]), 1e-4))


@pytest.mark.parametrize("c,hw,b", [(128, 512, 2), (256, 256, 2), (128, 4096, 1)])
def test_bf16_attention_kernels_match_exact_f32_kernels(vg, c, hw, b):
    """vg_attn_fwd_bf16 / vg_attn_bwd_bf16 (bf16 MFMA, transposed LDS reads) against the exact-f32 MFMA kernels on the
    SAME bf16-representable q, k, v: scores are exact products either way, so only P and dS are rounded (2^-9)."""
    from vae_gan_amd import lib as L
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(c + hw)
    d = 32
    width = c // 8
    q = torch.zeros(b, hw, d, device="cuda")
    k = torch.zeros(b, hw, d, device="cuda")
    q[..., :width] = 0.6 * torch.randn(b, hw, width, device="cuda", generator=g)
    k[..., :width] = 0.6 * torch.randn(b, hw, width, device="cuda", generator=g)
    v = torch.randn(b, hw, c, device="cuda", generator=g)
    d_o = torch.randn(b, hw, c, device="cuda", generator=g)
    qb, kb, vb = q.bfloat16(), k.bfloat16(), v.bfloat16()
    qf, kf, vf = qb.float(), kb.float(), vb.float()
    assert lib.vg_attn_supported_bf16(hw, d, c)

    def run(fwd, bwd, qq, kk, vv):
        o = torch.empty(b, hw, c, device="cuda")
        lse = torch.empty(b * hw, device="cuda")
        L.check(fwd(qq.data_ptr(), kk.data_ptr(), vv.data_ptr(), o.data_ptr(), lse.data_ptr(), b, hw, d, c, L.stream()), "fwd")
        dq, dk, dv = torch.empty(b, hw, d, device="cuda"), torch.empty(b, hw, d, device="cuda"), torch.empty(b, hw, c, device="cuda")
        ws = torch.empty(b * hw, device="cuda")
        L.check(bwd(qq.data_ptr(), kk.data_ptr(), vv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dq.data_ptr(),
                    dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), b, hw, d, c, L.stream()), "bwd")
        torch.cuda.synchronize()
        return o, lse, dq, dk, dv
    fast = run(lib.vg_attn_fwd_bf16, lib.vg_attn_bwd_bf16, qb, kb, vb)
    ref = run(lib.vg_attn_fwd, lib.vg_attn_bwd, qf, kf, vf)
    for name, a, r, tol in zip(("o", "lse", "dq", "dk", "dv"), fast, ref, (4e-3, 1e-5, 1e-2, 1e-2, 6e-3)):
        close_l2(a, r.cpu(), rtol=tol, what=f"attn_bf16.{name}")
    if width < d:      # zero-padded q/k channels get exactly zero gradients
        assert float(fast[2][..., width:].abs().max()) == 0.0 and float(fast[3][..., width:].abs().max()) == 0.0


def test_self_attention_module_bf16_mode(vg):
    """SelfAttention(128) in the bf16 compute mode (bf16 MFMA attention kernels) against the reference golden:
    north_star's bf16 tolerance, 2e-2."""
    from vae_gan_amd.hierarchical_vae_gan.model import SelfAttention
    vg.set_compute_dtype(torch.bfloat16)
    try:
        g = load_golden("hier_attention128")
        sa = _filled(SelfAttention(128)).cuda().train()
        x = as_t(g["x"]).cuda().requires_grad_(True)
        y = vg.to_nchw(sa(vg.to_nhwc(x)), 128)
        (y * as_t(g["cot0"]).cuda()).sum().backward()
        torch.cuda.synchronize()
        close_l2(y, g["y0"], rtol=2e-2, what="attn128.bf16.y")
        close_l2(x.grad, g["dx"], rtol=2e-2, what="attn128.bf16.dx")
        from vae_gan_amd.testing import digest
        for key, ref in g.items():
            if key.startswith("gd:") and key.endswith(("weight", "gamma")):
                mine = digest(key[3:], dict(sa.named_parameters())[key[3:]].grad)
                assert abs(mine[0] - ref[0]) <= 3e-2 * ref[0], (key, mine[0], ref[0])
    finally:
        vg.set_compute_dtype(torch.float32)


def test_hier_step_bf16_close_to_reference(vg):
    """bf16 mode (bf16 MFMA convs and attention): first-iteration generator losses within 2e-2 of the fp32 reference.
    The discriminator loss is held to 6e-2: with the parity fill (SelfAttention.gamma = 0.7, unit-variance
    activations) the unscaled attention logits of hierarchical_vae_gan/model.py:37 reach |s| ~ 50, the softmax is
    nearly one-hot, and bf16 STORAGE of q/k/v decides which key a query locks on to.  The hinge loss then moves in
    steps with nothing but the f32 summation ORDER inside the conv kernels (tests/gpu_hier_bf16_diag.py, one fixture,
    same bf16 tensors everywhere): -0.25 % (row-major tap walk), +2.2 % (igemm2 generation), +3.2 % (exact-f32
    attention kernels, VG_ATTN_F32=1), +3.9 % (im2col-row generation, and with the weights-stationary kernels off),
    +4.2 % (column-major tap walk, today's default) -- every other loss of the step stays within 0.3 %.  It is the
    activation storage, not any one kernel's arithmetic; the per-layer teacher-forced tests (test_gpu_layer_grads.py)
    hold 2e-2 where no such amplifier sits behind the layer."""
    from vae_gan_amd.hierarchical_vae_gan.config import Config
    from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalVAEGAN
    from vae_gan_amd.testing import fill_state_dict, hier_step_eps
    from vae_gan_amd.train import build_optimizers, hier_train_step
    from test_gpu_step import bf16_bound
    vg.set_compute_dtype(torch.bfloat16)
    try:
        g = load_golden("hier_step_128")
        cfg = Config()
        cfg.recon_target_policy, cfg.hier_decoder_policy = "area", "nearest"
        model = HierarchicalVAEGAN(cfg)
        model.load_state_dict(fill_state_dict(model.state_dict()))
        model.cuda().train()
        opt_g, opt_d, _, _ = build_optimizers(model, cfg)
        eps = hier_step_eps(2, 128)
        pair = [[e.cuda() for e in fwd] for fwd in eps[0]]
        dl, gl = hier_train_step(model, as_t(g["images"]).cuda(), opt_g, opt_d, eps_lists=pair)
        torch.cuda.synchronize()
        for k, v in {"d_loss": dl["total"], "d_kl": dl["kl"], "g_loss": gl["total"], "recon": gl["recon"], "kl": gl["kl"]}.items():
            ref = float(g[f"it0_{k}"])
            tol = bf16_bound(g, k)       # max(2e-2, 1.5 x the largest of the reference's own 8 bf16-storage samples): d_loss 5.5e-2
            assert abs(float(v.detach()) - ref) <= tol * max(abs(ref), 1e-3), (k, float(v.detach()), ref)
        assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    finally:
        vg.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("c,h,w", [(128, 5, 7), (128, 4, 4), (256, 3, 3), (128, 9, 8)])
def test_self_attention_ragged_sizes_match_oracle(vg, c, h, w):
    """HW that is not a multiple of the 32-wide key / query tiles (small training crops: 64x64 inputs give the
    discriminator a 4x4 attention map): the last tile is masked, results against the CPU oracle (torch autograd)."""
    from oracle import vaegan_oracle as O
    from vae_gan_amd.hierarchical_vae_gan.model import SelfAttention
    from vae_gan_amd.testing import synthetic_eps, cotangent
    sa = _filled(SelfAttention(c)).cuda().train()
    sd = {"a." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in sa.state_dict().items()}
    x = synthetic_eps((2, c, h, w), 77)
    cot = cotangent(f"ragged:{c}:{h}:{w}", (2, c, h, w))
    xc = x.clone().requires_grad_(True)
    want = O.self_attention(sd, "a", xc)
    (want * cot).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = vg.to_nchw(sa(vg.to_nhwc(xg)), c)
    (y * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    close_l2(y, want.detach(), rtol=1e-4, what="ragged attn y")
    close_l2(xg.grad, xc.grad, rtol=1e-4, what="ragged attn dx")
    for name, p in sa.named_parameters():
        if name == "key.bias":     # q.(k + b) shifts every key's score of a query alike: softmax-invariant, the gradient is rounding noise
            continue
        close_l2(p.grad, sd["a." + name].grad, rtol=2e-4, what=name)
