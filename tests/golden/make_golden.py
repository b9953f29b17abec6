#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the REFERENCE modules.

Runs only in the build container (needs /root/reference); the resulting ``*.npz`` files are
data (inputs, expected outputs, gradient digests) and are committed.  No reference source is
copied: the modules are imported by path and executed on CPU.  Weights are not stored —
both sides regenerate them with ``vae_gan_amd.testing.fill_state_dict`` from key names.

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py vq_step    # one group
"""
import contextlib
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from vae_gan_amd.testing import (fill_state_dict, synthetic_images, synthetic_eps,  # noqa: E402
                                 cotangent, digest, hier_step_eps)

REF = os.environ.get("VG_REFERENCE", "/root/reference")


def load_ref(sub, name):
    spec = importlib.util.spec_from_file_location(f"ref_{sub}_{name}", os.path.join(REF, sub, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def filled(module, salt=0):
    module.load_state_dict(fill_state_dict(module.state_dict(), salt=salt))
    return module


def grads_digest(module, prefix=""):
    out = {}
    for k, p in module.named_parameters():
        if p.grad is not None:
            out["gd:" + prefix + k] = digest(prefix + k, p.grad)
    return out


def bn_buffers(module, prefix=""):
    out = {}
    for k, b in module.named_buffers():
        out["buf:" + prefix + k] = b.detach().clone().numpy()
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


# ----------------------------------------------------------------------------------------
def g_vq_quantizer():
    m = load_ref("vq_vae_gan", "model")
    for tag, K, D, shape, init in [("a", 512, 64, (2, 64, 8, 8), "normal"),
                                   ("b", 512, 64, (2, 64, 8, 8), "default"),
                                   ("c", 8192, 256, (1, 256, 8, 8), "normal")]:
        vq = m.VectorQuantizer(K, D, 0.25, 0.99)
        sd = fill_state_dict(vq.state_dict())
        if init == "default":  # reference default: U(-1/K, 1/K), ema_w = copy, cluster sizes 0
            g = torch.Generator().manual_seed(5)
            sd["embeddings.weight"] = (torch.rand(K, D, generator=g) * 2 - 1) / K
            sd["ema_w"] = sd["embeddings.weight"].clone()
            sd["ema_cluster_size"] = torch.zeros(K)
        vq.load_state_dict(sd)
        scale = 0.5 if init == "normal" else 1.0 / K
        z = (synthetic_eps(shape, 100 + ord(tag)) * scale).requires_grad_(True)
        cot = cotangent("vq:" + tag, shape)
        zq, loss, idx = vq(z, training=True)
        ((zq * cot).sum() + loss).backward()
        extra = {}
        if init == "default":   # not reproducible from key names: store the starting state
            extra = dict(codebook_before=sd["embeddings.weight"], ema_cluster_before=sd["ema_cluster_size"],
                         ema_w_before=sd["ema_w"])
        if K * D > 100000:      # big codebook: digests only (fixtures stay small)
            after = {"gd:ema_w": digest("ema_w", vq.ema_w), "gd:codebook_after": digest("codebook_after", vq.embeddings.weight)}
        else:
            after = dict(ema_w=vq.ema_w, codebook_after=vq.embeddings.weight)
        save("vq_quantizer_" + tag, z=z, cot=cot, zq=zq, vq_loss=loss, idx=idx, dz=z.grad,
             ema_cluster_size=vq.ema_cluster_size, **after, **extra)


def rel_l2(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def _fwd_bwd_once(name, mod, x, dtype):
    mod = mod.to(dtype).train()
    x = x.detach().clone().to(dtype).requires_grad_(True)
    y = mod(x)
    ys = y if isinstance(y, (tuple, list)) else [y]
    flat = []
    for t in ys:
        flat.extend(t if isinstance(t, (tuple, list)) else [t])
    cots = [cotangent(f"{name}:{i}", t.shape).to(dtype) for i, t in enumerate(flat)]
    sum((t * c).sum() for t, c in zip(flat, cots)).backward()
    grads = {k: p.grad.detach().clone() for k, p in mod.named_parameters() if p.grad is not None}
    bufs = {k: b.detach().clone() for k, b in mod.named_buffers()}
    return [t.detach() for t in flat], cots, x.grad.detach(), grads, bufs


def _module_fwd_bwd(name, make_mod, x, prefix):
    """Golden = the reference module run in float64 (stored rounded to fp32: the 'truth'),
    plus 'noise:*' = relative L2 distance of the reference's own fp32 run from that truth.
    A piecewise-linear net has LeakyReLU sign flips under fp32 rounding, so some gradients
    are only reproducible to that noise floor (e.g. vq_decoder dx: 9e-3), not to 1e-4."""
    y64, cots, dx64, g64, b64 = _fwd_bwd_once(name, make_mod(), x, torch.float64)
    y32, _, dx32, g32, b32 = _fwd_bwd_once(name, make_mod(), x, torch.float32)
    out = {"x": x, "dx": dx64.float(), "noise:dx": rel_l2(dx32, dx64)}
    for i, (a, b) in enumerate(zip(y32, y64)):
        out[f"y{i}"] = b.float()
        out[f"cot{i}"] = cots[i].float()
        out[f"noise:y{i}"] = rel_l2(a, b)
    for k in g64:
        out["gd:" + prefix + k] = digest(prefix + k, g64[k])
        out["noise:gd:" + prefix + k] = rel_l2(g32[k], g64[k])
    for k in b64:
        out["buf:" + prefix + k] = b64[k].float() if b64[k].dtype.is_floating_point else b64[k]
    save(name, **out)


def g_resblock():
    m = load_ref("vq_vae_gan", "model")
    _module_fwd_bwd("resblock64", lambda: filled(m.ResidualBlock(64, 64)), synthetic_eps((2, 64, 16, 16), 7), "")


def g_vq_modules():
    m = load_ref("vq_vae_gan", "model")
    img = synthetic_images(2, 64, seed=42)
    _module_fwd_bwd("vq_encoder", lambda: filled(m.Encoder(3, [64, 128, 256, 512], 64)), img, "")
    _module_fwd_bwd("vq_decoder", lambda: filled(m.Decoder(64, [512, 256, 128, 64], 3)),
                    0.5 * synthetic_eps((2, 64, 8, 8), 11), "")
    _module_fwd_bwd("vq_discriminator", lambda: filled(m.Discriminator(3, [64, 128, 256, 512])), img, "")


def _ref_vq_step(model, images, opt_g, opt_d):
    """Re-composition of vq_vae_gan/main.py:75-111 around the reference model object."""
    opt_d.zero_grad()
    recon, _, _, vq_loss, idx1 = model(images)
    d_real = model.discriminator(images)
    d_fake = model.discriminator(recon.detach())
    dl = model.loss_function(images, recon, vq_loss, d_real, d_fake, train_generator=False)
    dl["total"].backward()
    opt_d.step()
    opt_g.zero_grad()
    recon, _, _, vq_loss, idx2 = model(images)
    d_fake = model.discriminator(recon)
    gl = model.loss_function(images, recon, vq_loss, d_real, d_fake, train_generator=True)
    gl["total"].backward()
    opt_g.step()
    return dl, gl, idx1, idx2, recon


class _Bf16Storage:
    """The reference with bf16 STORAGE (float64 arithmetic): what an ideal bf16 engine computes.  Every tensor the HIP
    engine keeps in bf16 is rounded where the engine rounds it: the output of every convolution, transposed convolution,
    LeakyReLU and Sigmoid (forward hooks; their backward rounds nothing extra, like the casts' own backward in
    ``_unit_run``), and the conv / Linear weights the kernels read (``weights()``: the float masters are put back before
    the optimizer steps, as the engine keeps f32 masters and re-packs bf16 images after each step)."""

    def __init__(self, model, sample=0):
        self.model, self.hooks = model, []
        kinds = (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.LeakyReLU, torch.nn.Sigmoid)
        # sample > 0: another, equally legitimate bf16 engine.  Two engines that differ only in the ORDER of their f32
        # accumulation differ by ~2^-23 relative before the rounding to bf16, which decides the rounding (and, behind a
        # LeakyReLU, the mask) of the few values that sit on a boundary: a relative jitter of that size before every
        # rounding samples exactly that.
        gen = torch.Generator().manual_seed(1000 + sample) if sample else None

        def rnd(t):
            if gen is not None and t.dtype.is_floating_point:
                t = t * (1.0 + 2.0 ** -23 * torch.randn(t.shape, generator=gen, dtype=t.dtype))
            return t.to(torch.bfloat16).to(t.dtype)
        for leaf in model.modules():
            if isinstance(leaf, kinds):
                self.hooks.append(leaf.register_forward_hook(lambda _m, _i, o: rnd(o)))
            if isinstance(leaf, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                # the engine's activations are bf16 tensors: whatever enters a convolution from outside the stack (the
                # quantised latents, a reparameterised z, a concatenation with one) is rounded on entry (ops.to_nhwc);
                # a no-op for what a hooked layer produced
                self.hooks.append(leaf.register_forward_pre_hook(lambda _m, args: tuple(rnd(a) for a in args)))

    def weights(self):
        model = self.model

        class _W:
            def __enter__(w):
                w.saved = [(leaf.weight, leaf.weight.data) for leaf in model.modules()
                           if isinstance(leaf, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.Linear))]
                for p_, d in w.saved:
                    p_.data = d.to(torch.bfloat16).to(d.dtype)

            def __exit__(w, *exc):
                for p_, d in w.saved:
                    p_.data = d
                return False
        return _W()

    def close(self):
        for h in self.hooks:
            h.remove()


def _ref_vq_step_bf16(model, images, opt_g, opt_d, st):
    """``_ref_vq_step`` with bf16 storage (see _Bf16Storage); the optimizers step the float masters."""
    rnd = lambda t: t.to(torch.bfloat16).to(t.dtype)
    images = rnd(images)
    opt_d.zero_grad()
    with st.weights():
        recon, _, _, vq_loss, idx1 = model(images)
        d_real = model.discriminator(images)
        d_fake = model.discriminator(recon.detach())
        dl = model.loss_function(images, recon, vq_loss, d_real, d_fake, train_generator=False)
        dl["total"].backward()
    opt_d.step()
    opt_g.zero_grad()
    with st.weights():
        recon, _, _, vq_loss, idx2 = model(images)
        d_fake = model.discriminator(recon)
        gl = model.loss_function(images, recon, vq_loss, d_real, d_fake, train_generator=True)
        gl["total"].backward()
    opt_g.step()
    return dl, gl, idx1, idx2, recon


N_BF16_SAMPLES = 8


def _bf16_floor(run, truth):
    """'noise_bf16:it0_*': how far the first iteration's scalars of the reference run with bf16 STORAGE are from the
    exact run, the floor any bf16 engine has on them.  The discriminator's hinge loss sits behind a stack of LeakyReLU
    masks (and, in the hierarchical model, a nearly one-hot softmax) and moves in steps with the rounding points, so one
    run is one sample of a distribution: N_BF16_SAMPLES equally legitimate bf16 engines are run (sample 0: plain
    round-to-nearest of fp64 values; the others: the same behind a 2^-23 relative jitter, i.e. another f32 accumulation
    order, see _Bf16Storage) and the largest distance is recorded ('noise_bf16v:*' keeps them all)."""
    out = {}
    runs = [run(sample) for sample in range(N_BF16_SAMPLES)]
    for k in runs[0]:
        if k.startswith("it0_") and not k.endswith(("idx1", "idx2", "recon_img")):
            d = [abs(float(r[k]) - float(truth[k])) / max(abs(float(truth[k])), 1e-12) for r in runs]
            out["noise_bf16:" + k] = max(d)
            out["noise_bf16v:" + k] = np.array(d)
        if k == "it0_idx2":
            out["idxeq_bf16:it0_idx2"] = float((runs[0][k] == truth[k]).float().mean())
    return out


def _vq_step_run(dtype, iters=2, size=64, bf16_storage=False, sample=0):
    m = load_ref("vq_vae_gan", "model")
    cfg = load_ref("vq_vae_gan", "config").Config()
    model = filled(m.VQVAEGAN(cfg)).to(dtype).train()
    opt_g = torch.optim.Adam(list(model.encoder.parameters()) + list(model.decoder.parameters()),
                             lr=cfg.lr, betas=(cfg.beta1, cfg.beta2))
    opt_d = torch.optim.Adam(model.discriminator.parameters(), lr=cfg.lr / 2, betas=(cfg.beta1, cfg.beta2))
    images = synthetic_images(2, size, seed=42).to(dtype)
    out = {"images": images.float()}
    st = _Bf16Storage(model, sample) if bf16_storage else None
    for it in range(iters):
        if st is not None:
            dl, gl, idx1, idx2, recon = _ref_vq_step_bf16(model, images, opt_g, opt_d, st)
        else:
            dl, gl, idx1, idx2, recon = _ref_vq_step(model, images, opt_g, opt_d)
        out.update({f"it{it}_d_loss": dl["total"].item(), f"it{it}_d_recon": dl["recon"],
                    f"it{it}_g_loss": gl["total"].item(), f"it{it}_recon": gl["recon"],
                    f"it{it}_vq": gl["vq"], f"it{it}_adv": gl["adv"],
                    f"it{it}_idx1": idx1, f"it{it}_idx2": idx2, f"it{it}_recon_img": recon.detach().clone().float()})
    return out, model.state_dict()


def g_vq_step(name="vq_step_64", iters=2, size=64):
    """Two iterations of the reference step in fp32 (the golden) and in fp64 (to record how far
    the reference's own fp32 run is from exact arithmetic: 'noise:*').  Adam's first step is
    lr*sign(g), so parameters whose gradient is rounding noise move by +-lr in a direction that
    depends on summation order; scalars downstream are reproducible only to that floor.
    'noise_bf16:it0_*' = the first iteration's scalars of the reference run with bf16 STORAGE (_Bf16Storage) against
    the fp64 run: the floor a bf16 engine has on them (the bounds of the bf16 step tests derive from it)."""
    out, sd = _vq_step_run(torch.float32, iters, size)
    out64, sd64 = _vq_step_run(torch.float64, iters, size)
    # how far the reference's own fp32 run is from its fp64 run in the final state: Adam's lr * sign(g) on gradients that
    # are rounding noise moves an element by +-lr whatever the arithmetic, so the digests carry a floor of their own
    dn = [(digest(k_, sd[k_]), digest(k_, sd64[k_])) for k_ in sd if sd[k_].dtype != torch.int64]
    out["noise:state"] = max(max(abs(a[0] - b[0]), abs(a[2] - b[2]) / 4) / max(float(b[0]), 1e-6) for a, b in dn)
    out.update(_bf16_floor(lambda smp: _vq_step_run(torch.float64, 1, size, bf16_storage=True, sample=smp)[0], out64))
    for k, v in list(out.items()):
        if k.startswith("it") and not k.endswith(("idx1", "idx2", "recon_img")):
            out["noise:" + k] = abs(v - out64[k]) / max(abs(out64[k]), 1e-12)
        if k.startswith("it") and k.endswith(("idx1", "idx2")):
            out["idxeq64:" + k] = float((v == out64[k]).float().mean())
        if k.startswith("it") and k.endswith("recon_img"):
            out["noise:" + k] = rel_l2(v, out64[k])
    for k, v in sd.items():
        if v.dtype == torch.int64 or k.startswith("vector_quantizer"):
            out["sd:" + k] = v.clone()
        else:
            out["sdd:" + k] = digest(k, v)
    save(name, **out)
    print({k: v for k, v in out.items() if k.startswith(("noise:", "noise_bf16:", "idxeq"))})


def g_vq_step_256():
    """one iteration of the reference VQ step at the reference's own image_size = 256 (B = 2):
    the shapes at which the benchmarked kernels (igemm3, igemm3s2, igemm4, igemm4c, igemm4p, thin, wgrad3) are selected."""
    g_vq_step("vq_step_256", iters=1, size=256)


def g_losses():
    mu = synthetic_eps((4, 128), 21)
    lv = torch.clamp(2.0 * synthetic_eps((4, 128), 22), -20, 20)
    eps = synthetic_eps((4, 128), 23)
    mb = load_ref("beta_vae_gans", "model")

    class C:  # minimal config object for the constructor (beta_vae_gans/config.py names)
        latent_dim = 128; beta = 4.0; channels = 3
        enc_channels = [8, 8]; dec_channels = [8, 8]; lambda_adv = 0.1
    # reparameterize draws randn_like internally: reproduce by seeding and recording the draw
    torch.manual_seed(1234)
    eps_drawn = torch.randn_like(mu)
    torch.manual_seed(1234)
    z = mb.BetaVAEGAN.reparameterize(None, mu, lv)
    d_real = synthetic_eps((2, 1, 15, 15), 24)
    d_fake = synthetic_eps((2, 1, 7, 7), 25)
    kl = -0.5 * torch.mean(1 + lv - mu.pow(2) - lv.exp())           # beta model.py:258 evaluated
    hinge = torch.mean(F.relu(1.0 - d_real)) + torch.mean(F.relu(1.0 + d_fake))
    adv = -torch.mean(d_fake)
    save("losses", mu=mu, logvar=lv, eps=eps_drawn, z=z, d_real=d_real, d_fake=d_fake, kl=kl, hinge=hinge, adv=adv)


def g_beta_modules():
    m = load_ref("beta_vae_gans", "model")
    img = synthetic_images(2, 256, seed=43)
    _module_fwd_bwd("beta_encoder", lambda: filled(m.Encoder(3, [64, 128, 256, 512], 128)), img, "")
    _module_fwd_bwd("beta_decoder", lambda: filled(m.Decoder(128, [512, 256, 128, 64], 3)), synthetic_eps((2, 128), 31), "")
    _module_fwd_bwd("beta_discriminator_128", lambda: filled(m.Discriminator(3, [64, 128, 256, 512])),
                    synthetic_images(2, 128, seed=44), "")


def g_hier_modules():
    m = load_ref("hierarchical_vae_gan", "model")
    _module_fwd_bwd("hier_attention128", lambda: filled(m.SelfAttention(128)), synthetic_eps((2, 128, 16, 16), 51), "")
    _module_fwd_bwd("hier_attention256", lambda: filled(m.SelfAttention(256)), synthetic_eps((1, 256, 8, 8), 52), "")
    img = synthetic_images(2, 64, seed=45)
    _module_fwd_bwd("hier_encoder", lambda: filled(m.HierarchicalEncoder(3, 64, [32, 64, 128], 3)), img, "")
    _module_fwd_bwd("hier_discriminator", lambda: filled(m.Discriminator(3, 64, 3)), img, "")
    dec = lambda: filled(m.HierarchicalDecoder(3, 64, [32, 64, 128], 3))
    _module_fwd_bwd("hier_dec_block0", lambda: dec().decoders[0], 0.5 * synthetic_eps((2, 128, 8, 8), 53), "decoders.0.")
    _module_fwd_bwd("hier_dec_block1", lambda: dec().decoders[1], 0.5 * synthetic_eps((2, 320, 16, 16), 54), "decoders.1.")
    _module_fwd_bwd("hier_dec_block2", lambda: dec().decoders[2], 0.5 * synthetic_eps((2, 160, 32, 32), 55), "decoders.2.")
    _module_fwd_bwd("hier_output_conv", lambda: dec().output_conv, 0.5 * synthetic_eps((2, 64, 32, 32), 56), "output_conv.")


# ----------------------------------------------------------------------------------------
# Whole-step goldens for the two architectures whose reference step raises.
# The REFERENCE modules and the REFERENCE loss_function do all the arithmetic; the harness only
# (D1) hands loss_function the 2x2-area-downsampled original so that mse_loss is defined on the
# 128x128 reconstruction, (D2) chains the reference decoder sub-blocks with nearest x2 upsampling
# instead of the mis-indexed ConvTranspose list, and injects the N(0,1) draws of reparameterize.
# ----------------------------------------------------------------------------------------
class _InjectedNoise:
    """Replace torch.randn_like by a queue of prepared draws (same values in the fp32 and fp64 runs)."""

    def __init__(self, draws):
        self.draws = list(draws)

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda t, **kw: self.draws.pop(0).to(t.dtype)
        return self

    def __exit__(self, *exc):
        torch.randn_like = self.orig
        assert not self.draws, "unused noise draws"
        return False


def _step_digest(out, out64, sd):
    for k, v in list(out.items()):
        if k.startswith("it") and not k.endswith("recon_img"):
            out["noise:" + k] = abs(v - out64[k]) / max(abs(out64[k]), 1e-12)
        if k.endswith("recon_img"):
            out["noise:" + k] = rel_l2(v, out64[k])
    for k, v in sd.items():
        if v.dtype == torch.int64:
            out["sd:" + k] = v.clone()
        else:
            out["sdd:" + k] = digest(k, v)
    return out


def _adam_pair(model, cfg):
    og = torch.optim.Adam(list(model.encoder.parameters()) + list(model.decoder.parameters()), lr=cfg.lr,
                          betas=(cfg.beta1, cfg.beta2))
    od = torch.optim.Adam(model.discriminator.parameters(), lr=cfg.lr / 2, betas=(cfg.beta1, cfg.beta2))
    return og, od


def _beta_step_run(dtype, iters, images, eps, bf16=None):
    """beta_vae_gans/main.py:75-108 around the reference model; D1: target = avg_pool2d(original, 2)."""
    m = load_ref("beta_vae_gans", "model")
    cfg = load_ref("beta_vae_gans", "config").Config()
    model = filled(m.BetaVAEGAN(cfg)).to(dtype).train()
    opt_g, opt_d = _adam_pair(model, cfg)
    images = images.to(dtype)
    st = _Bf16Storage(model, bf16 - 1) if bf16 else None       # bf16 = 1 + sample number
    wctx = st.weights if st else contextlib.nullcontext
    if st:
        images = images.to(torch.bfloat16).to(dtype)
    target = F.avg_pool2d(images, 2)
    out = {}
    for it in range(iters):
        opt_d.zero_grad()
        with wctx():
            with _InjectedNoise([eps[it]]):
                recon, _, mu, logvar = model(images)
            d_real = model.discriminator(images)
            d_fake = model.discriminator(recon.detach())
            dl = model.loss_function(target, recon, mu, logvar, d_real, d_fake, train_generator=False)
            dl["total"].backward()
        opt_d.step()
        opt_g.zero_grad()
        with wctx():
            d_fake = model.discriminator(recon)
            gl = model.loss_function(target, recon, mu, logvar, d_real, d_fake, train_generator=True)
            gl["total"].backward()
        opt_g.step()
        out.update({f"it{it}_d_loss": dl["total"].item(), f"it{it}_g_loss": gl["total"].item(), f"it{it}_recon": gl["recon"],
                    f"it{it}_kl": gl["kl"], f"it{it}_adv": gl["adv"], f"it{it}_recon_img": recon.detach().clone().float()})
    return out, model.state_dict()


def g_beta_step():
    images = synthetic_images(2, 256, seed=46)
    eps = [synthetic_eps((2, 128), 60 + it) for it in range(2)]
    out, sd = _beta_step_run(torch.float32, 2, images, eps)
    out64, _ = _beta_step_run(torch.float64, 2, images, eps)
    out = _step_digest(out, out64, sd)
    save("beta_step_256", images_seed=46, eps0=eps[0], eps1=eps[1], **out)
    print({k: v for k, v in out.items() if k.startswith("noise:")})


def _nearest_decoder(dec):
    """D2: HierarchicalDecoder sub-blocks chained with nearest x2 upsampling (decoder.upsample.* unused)."""
    def forward(z_list):
        x = None
        for j, z in enumerate(reversed(z_list)):
            x = dec.decoders[j](z if j == 0 else torch.cat([z, F.interpolate(x, scale_factor=2, mode="nearest")], dim=1))
        return dec.output_conv(x)
    return forward


def _hier_step_run(dtype, iters, images, eps, bf16=None):
    """hierarchical_vae_gan/main.py:75-111 around the reference model with D2 (decoder) and D1 (target)."""
    m = load_ref("hierarchical_vae_gan", "model")
    cfg = load_ref("hierarchical_vae_gan", "config").Config()
    model = filled(m.HierarchicalVAEGAN(cfg)).to(dtype).train()
    model.decoder.forward = _nearest_decoder(model.decoder)
    opt_g, opt_d = _adam_pair(model, cfg)
    images = images.to(dtype)
    st = _Bf16Storage(model, bf16 - 1) if bf16 else None       # bf16 = 1 + sample number
    wctx = st.weights if st else contextlib.nullcontext
    if st:
        images = images.to(torch.bfloat16).to(dtype)
    target = F.avg_pool2d(images, 2)
    out = {}
    for it in range(iters):
        opt_d.zero_grad()
        with wctx():
            with _InjectedNoise(eps[it][0]):
                recon, _, mu_l, lv_l = model(images)
            d_real = model.discriminator(images)
            d_fake = model.discriminator(recon.detach())
            dl = model.loss_function(target, recon, mu_l, lv_l, d_real, d_fake, train_generator=False)
            dl["total"].backward()
        opt_d.step()
        opt_g.zero_grad()
        with wctx():
            with _InjectedNoise(eps[it][1]):
                recon, _, mu_l, lv_l = model(images)
            d_fake = model.discriminator(recon)
            gl = model.loss_function(target, recon, mu_l, lv_l, d_real, d_fake, train_generator=True)
            gl["total"].backward()
        opt_g.step()
        out.update({f"it{it}_d_loss": dl["total"].item(), f"it{it}_d_kl": dl["kl"], f"it{it}_g_loss": gl["total"].item(),
                    f"it{it}_recon": gl["recon"], f"it{it}_kl": gl["kl"], f"it{it}_adv": gl["adv"],
                    f"it{it}_recon_img": recon.detach().clone().float()})
    return out, model.state_dict()


def g_hier_step():
    images = synthetic_images(2, 128, seed=47)
    out, sd = _hier_step_run(torch.float32, 2, images, hier_step_eps(2, 128))
    out64, _ = _hier_step_run(torch.float64, 2, images, hier_step_eps(2, 128))
    out = _step_digest(out, out64, sd)
    save("hier_step_128", images=images, **out)
    print({k: v for k, v in out.items() if k.startswith("noise:")})



def g_vq_codec():
    """VQ index entropy coder (vq_vae_gan/model.py:337-576): bitstreams of the REFERENCE ArithmeticEncoder.encode.
    The reference module never imports numpy (model.py:5-8 vs :378: NameError when called); the harness injects it
    into the loaded module's namespace, nothing else is touched.  Also records whether the reference's own decode
    returns the encoded indices (it re-uses the symbol counter as its bit pointer, model.py:491,530-532)."""
    m = load_ref("vq_vae_gan", "model")
    m.np = np
    out = {}
    cases = [("k512_skewed", 512, (1, 32, 32), 1.5, True), ("k512_uniform_probs", 512, (2, 8, 8), 0.0, False),
             ("k8192", 8192, (1, 16, 16), 1.2, True), ("k16_tiny", 16, (1, 4, 4), 1.0, True),
             ("k16_empty", 16, (1, 0, 4), 0.0, False), ("k512_one_symbol", 512, (1, 4, 4), -1.0, True)]
    for tag, K, shape, zipf, use_hist in cases:
        g = torch.Generator().manual_seed(len(tag) * 131 + K)
        if zipf < 0:     # a collapsed codebook: every latent picks the same code (probability exactly 1)
            idx = torch.full(shape, 37, dtype=torch.int64)
        elif zipf > 0:     # skewed usage, like a trained codebook
            w = 1.0 / torch.arange(1, K + 1, dtype=torch.float64) ** zipf
            idx = torch.multinomial(w / w.sum(), int(np.prod(shape)), replacement=True, generator=g).reshape(shape)
            idx = idx[..., torch.randperm(idx.shape[-1], generator=g)]
        else:
            idx = torch.randint(0, K, shape, generator=g)
        enc = m.ArithmeticEncoder(K)
        if use_hist:
            probs = m.calculate_codebook_usage(idx, K)
            data = enc.encode(idx, probs)
            out[f"{tag}:probs"] = probs
        else:
            probs = None
            data = enc.encode(idx)
        try:
            back = enc.decode(data, tuple(shape), probs)
            rt = bool((back == idx).all())
        except Exception:
            rt = False
        out[f"{tag}:idx"] = idx
        out[f"{tag}:bytes"] = np.frombuffer(bytes(data), dtype=np.uint8)
        out[f"{tag}:K"] = K
        out[f"{tag}:ref_decode_roundtrips"] = rt
        print(tag, "bytes", len(data), "reference decode round-trips:", rt)
    save("vq_codec", **out)


def g_augment():
    """Input transform (beta_vae_gans/data.py:106-119): the pixel arithmetic belongs to Pillow (the reference resizes
    PIL images through torchvision, which is not installed here).  Vectors made by PILLOW ITSELF: flips
    (Image.transpose), crop (Image.crop), Image.resize((S, S), BILINEAR); ToTensor = uint8 / 255 is left to the test."""
    from PIL import Image
    import PIL
    rng = np.random.default_rng(2024)
    out = {"pillow_version": np.array([int(x) for x in PIL.__version__.split(".")[:2]])}
    #        H    W   S  fh fv  (i, j, h, w) or None = whole image (eval transform)
    cases = [(97, 131, 64, 0, 0, (5, 7, 83, 113)), (97, 131, 64, 1, 0, (5, 7, 83, 113)), (97, 131, 64, 0, 1, (0, 0, 97, 120)),
             (97, 131, 64, 1, 1, (14, 18, 80, 90)), (200, 300, 96, 1, 1, (10, 20, 170, 255)), (40, 50, 64, 0, 1, (3, 2, 33, 45)),
             (64, 64, 64, 1, 0, (0, 0, 64, 64)), (301, 77, 32, 0, 0, None), (128, 192, 64, 0, 0, None), (64, 90, 64, 1, 1, (0, 11, 64, 64)),
             (90, 64, 64, 0, 0, (13, 0, 64, 64)), (33, 29, 96, 1, 0, (2, 3, 30, 25))]
    for n, (H, W, S, fh, fv, box) in enumerate(cases):
        if n % 2:
            yy, xx = np.mgrid[0:H, 0:W]
            img = np.stack([(np.sin(yy / 6.0) + 1) * 127.5, (np.cos(xx / 4.0) + 1) * 127.5, (yy * 3 + xx * 5) % 256], -1).astype(np.uint8)
        else:
            img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        pil = Image.fromarray(img)
        if fh:
            pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
        if fv:
            pil = pil.transpose(Image.FLIP_TOP_BOTTOM)
        i, j, h, w = box if box else (0, 0, H, W)
        res = pil.crop((j, i, j + w, i + h)).resize((S, S), Image.BILINEAR)
        out[f"c{n}:img"] = img
        out[f"c{n}:params"] = np.array([S, fh, fv, i, j, h, w])
        out[f"c{n}:out"] = np.asarray(res)
    out["n_cases"] = len(cases)
    save("augment", **out)


def _unit_run(u, index, dtype, bf16_storage=False):
    """One teacher-forced run of unit ``u`` on the reference modules.  bf16_storage: the same float64 arithmetic, but
    every tensor the HIP engine keeps in bf16 is rounded to bf16 where the engine rounds it: the input, the weights of
    the convolutions, the output of every convolution and of every activation (NOT of a BatchNorm: the engine fuses
    normalise (+ residual) + LeakyReLU and rounds once, after the activation; rounding the BatchNorm output before the
    residual add would move the LeakyReLU mask of ~0.06 % of a ResidualBlock's outputs), and (through the casts' own
    backward) every gradient that crosses such a boundary.  That is the reference itself with bf16 storage: its distance from the float64 run is
    the floor any bf16 engine has (LeakyReLU masks of pre-activations within bf16 rounding of zero flip)."""
    from layer_units import unit_input
    m = load_ref("vq_vae_gan" if u.arch == "vq" else "hierarchical_vae_gan", "model")
    mod = u.build(m)
    mod.load_state_dict(fill_state_dict(mod.state_dict(), salt=100 + index))
    mod = mod.to(dtype).train()
    rnd = lambda t: t.to(torch.bfloat16).to(t.dtype)
    hooks = []
    if bf16_storage:
        with torch.no_grad():
            for k, p_ in mod.named_parameters():
                if p_.dim() >= 2:
                    p_.copy_(rnd(p_))
        for leaf in mod.modules():
            if isinstance(leaf, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.LeakyReLU, torch.nn.Sigmoid)):
                hooks.append(leaf.register_forward_hook(lambda _m, _i, o: rnd(o)))
    x = unit_input(u, index).to(dtype)
    if bf16_storage:
        x = rnd(x)
    x.requires_grad_(True)
    y = mod(rnd(x) if bf16_storage else x)
    if bf16_storage:
        y = rnd(y)
    cot = cotangent("unit:" + u.name, y.shape).to(dtype)
    (y * cot).sum().backward()
    grads = {k: p_.grad.detach().clone() for k, p_ in mod.named_parameters() if p_.grad is not None}
    bufs = {k: b.detach().clone() for k, b in mod.named_buffers()}
    for h in hooks:
        h.remove()
    return y.detach(), x.grad.detach(), grads, bufs


def g_layer_units():
    """Teacher-forced per-layer fixtures (tests/layer_units.py): for every distinct fused unit the reference's float64
    output and input gradient (whole-tensor digest + a seeded 16 K-element sample) and the digest of every parameter
    gradient; 'noise:*' = the reference's own fp32 run against that, 'noise_bf16:*' = the reference with bf16 storage
    against it, and 'b:*' = that bf16-storage run itself (same rounding points as the engine's bf16 mode, so the
    LeakyReLU masks coincide and the engine can be held to a tight bound against it)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from layer_units import UNITS, sample_index
    out = {}
    for index, u in enumerate(UNITS):
        y64, dx64, g64, b64 = _unit_run(u, index, torch.float64)
        y32, dx32, g32, _ = _unit_run(u, index, torch.float32)
        yb, dxb, gb, _ = _unit_run(u, index, torch.float64, bf16_storage=True)
        n = u.name
        out[f"{n}:yshape"] = np.array(y64.shape)
        for tag, t64, t32, tb in (("y", y64, y32, yb), ("dx", dx64, dx32, dxb)):
            idx = sample_index(n + ":" + tag, t64.numel())
            out[f"{n}:{tag}:sample"] = t64.flatten()[idx].float()
            out[f"{n}:{tag}:dig"] = digest(n + ":" + tag, t64)
            out[f"{n}:noise:{tag}"] = rel_l2(t32, t64)
            out[f"{n}:noise_bf16:{tag}"] = rel_l2(tb, t64)
            out[f"{n}:b:{tag}:sample"] = tb.flatten()[idx].float()       # the bf16-storage reference run itself
            out[f"{n}:b:{tag}:dig"] = digest(n + ":" + tag, tb)
        for k in g64:
            out[f"{n}:gd:{k}"] = digest(k, g64[k])
            out[f"{n}:noise:gd:{k}"] = rel_l2(g32[k], g64[k])
            out[f"{n}:noise_bf16:gd:{k}"] = rel_l2(gb[k], g64[k])
            out[f"{n}:b:gd:{k}"] = digest(k, gb[k])
        for k, v in b64.items():
            out[f"{n}:buf:{k}"] = v.float() if v.dtype.is_floating_point else v
        print(n, tuple(y64.shape), {k.split(":", 1)[1]: f"{float(v):.1e}" for k, v in out.items()
                                    if k.startswith(n + ":noise") and ("dx" in k or "weight" in k)})
    save("layer_units", **out)


def g_bf16_floors():
    """Add 'noise_bf16:it0_*' (see _bf16_floor) to the three committed whole-step goldens; every other entry stays as it
    is.  The truth is the golden's own first-iteration scalars (the reference's fp32 run: 1e-7 from its fp64 run)."""
    def merge(name, run):
        path = os.path.join(HERE, name + ".npz")
        z = np.load(path, allow_pickle=False)
        old = {k: z[k] for k in z.files if not k.startswith(("noise_bf16", "idxeq_bf16"))}
        truth = {k: torch.as_tensor(v) if k.endswith("idx2") else float(v) for k, v in old.items()
                 if k.startswith("it0_") and not k.endswith(("idx1", "recon_img"))}
        add = _bf16_floor(run, truth)
        np.savez_compressed(path, **old, **{k: np.asarray(v) for k, v in add.items()})
        print(name, {k: (v if np.ndim(v) == 0 else list(v)) for k, v in add.items()})
    merge("vq_step_64", lambda smp: _vq_step_run(torch.float64, 1, 64, bf16_storage=True, sample=smp)[0])
    img_h = synthetic_images(2, 128, seed=47)
    merge("hier_step_128", lambda smp: _hier_step_run(torch.float64, 1, img_h, hier_step_eps(2, 128), bf16=1 + smp)[0])
    img_b = synthetic_images(2, 256, seed=46)
    merge("beta_step_256", lambda smp: _beta_step_run(torch.float64, 1, img_b, [synthetic_eps((2, 128), 60)], bf16=1 + smp)[0])
    merge("vq_step_256", lambda smp: _vq_step_run(torch.float64, 1, 256, bf16_storage=True, sample=smp)[0])


GROUPS = {"layer_units": g_layer_units, "vq_quantizer": g_vq_quantizer, "resblock": g_resblock, "vq_modules": g_vq_modules,
          "vq_step": g_vq_step, "vq_step_256": g_vq_step_256, "bf16_floors": g_bf16_floors, "losses": g_losses, "beta_modules": g_beta_modules,
          "hier_modules": g_hier_modules, "beta_step": g_beta_step, "hier_step": g_hier_step, "vq_codec": g_vq_codec, "augment": g_augment}

if __name__ == "__main__":
    torch.set_num_threads(8)
    for name in (sys.argv[1:] or list(GROUPS)):
        GROUPS[name]()
