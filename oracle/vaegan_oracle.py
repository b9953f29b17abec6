"""CPU oracle for the VAE-GAN training hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``vae_gan_amd``) never imports it and has no CPU
fallback: it fails loudly when the HIP library is missing.

What this is: a *functional* fp32 restatement (pure functions over a flat ``dict`` of
state_dict tensors, torch CPU ops, explicit BatchNorm / Adam / VQ-EMA arithmetic) of the
reference's training hot path.  Each function cites the reference lines it follows
(paths relative to the upstream repository root).  Gradients come from torch autograd on
CPU leaves, which is the same arithmetic the reference relies on (SURVEY.md §8(c):
"third-party arithmetic: all of it torch").

Pinned by: ``tests/golden/*.npz`` — outputs of the reference's own ``*/model.py`` modules
executed in the build container by ``tests/golden/make_golden.py`` (weights regenerated
on both sides by ``vae_gan_amd.testing.fill_state_dict``; the reference has no tests or
fixtures of its own).  ``tests/test_oracle_golden.py`` checks every function
below against those vectors.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

BN_EPS = 1e-5        # nn.BatchNorm2d default
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default
LRELU_SLOPE = 0.2    # every LeakyReLU in the three model.py files


# --------------------------------------------------------------------------------------
# primitive layers
# --------------------------------------------------------------------------------------
def conv2d(sd: SD, p: str, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def conv_transpose2d(sd: SD, p: str, x, stride=2, padding=1):
    return F.conv_transpose2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def batchnorm_train(sd: SD, p: str, x, update_stats: bool = True):
    """Training-mode BatchNorm2d: biased batch variance for normalisation, unbiased for the
    running estimate, momentum 0.1 (torch semantics relied on by every */model.py)."""
    n = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.mean(dim=(0, 2, 3))
    var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    if update_stats:
        with torch.no_grad():
            unbiased = var * (n / max(n - 1, 1))
            sd[p + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            sd[p + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unbiased.detach())
            sd[p + ".num_batches_tracked"].add_(1)
    xhat = (x - mean[None, :, None, None]) * torch.rsqrt(var + BN_EPS)[None, :, None, None]
    return xhat * sd[p + ".weight"][None, :, None, None] + sd[p + ".bias"][None, :, None, None]


def batchnorm_eval(sd: SD, p: str, x):
    inv = torch.rsqrt(sd[p + ".running_var"] + BN_EPS) * sd[p + ".weight"]
    return (x - sd[p + ".running_mean"][None, :, None, None]) * inv[None, :, None, None] \
        + sd[p + ".bias"][None, :, None, None]


def bn(sd, p, x, train=True):
    return batchnorm_train(sd, p, x) if train else batchnorm_eval(sd, p, x)


def lrelu(x):
    return torch.where(x > 0, x, x * LRELU_SLOPE)


def residual_block(sd: SD, p: str, x, train=True):
    """beta_vae_gans/model.py:27-40 (identical copies vq_vae_gan/model.py:27-40,
    hierarchical_vae_gan/model.py:67-80); skip is Identity at every call site."""
    out = lrelu(bn(sd, p + ".bn1", conv2d(sd, p + ".conv1", x, 1, 1), train))
    out = bn(sd, p + ".bn2", conv2d(sd, p + ".conv2", out, 1, 1), train)
    if (p + ".skip.weight") in sd:
        x = conv2d(sd, p + ".skip", x, 1, 0)
    return lrelu(out + x)


def self_attention(sd: SD, p: str, x):
    """hierarchical_vae_gan/model.py:24-47 — softmax over keys, no 1/sqrt(d) scale."""
    b, c, h, w = x.shape
    q = conv2d(sd, p + ".query", x).reshape(b, -1, h * w).transpose(1, 2)   # [B,HW,C/8]
    k = conv2d(sd, p + ".key", x).reshape(b, -1, h * w)                      # [B,C/8,HW]
    v = conv2d(sd, p + ".value", x).reshape(b, -1, h * w)                    # [B,C,HW]
    att = torch.softmax(torch.bmm(q, k), dim=-1)                             # [B,HW,HW]
    out = torch.bmm(v, att.transpose(1, 2)).reshape(b, c, h, w)
    return sd[p + ".gamma"] * out + x


# --------------------------------------------------------------------------------------
# VQ-VAE-GAN  (vq_vae_gan/model.py)
# --------------------------------------------------------------------------------------
def _down_stack(sd, p, x, n_levels, train):
    """conv3+BN+LReLU, then n_levels-1 × [Res, conv4s2p1, BN, LReLU], then Res.
    Shared by beta_vae_gans/model.py:49-66 and vq_vae_gan/model.py:143-158."""
    x = lrelu(bn(sd, f"{p}.1", conv2d(sd, f"{p}.0", x, 1, 1), train))
    i = 3
    for _ in range(n_levels - 1):
        x = residual_block(sd, f"{p}.{i}", x, train)
        x = lrelu(bn(sd, f"{p}.{i+2}", conv2d(sd, f"{p}.{i+1}", x, 2, 1), train))
        i += 4
    x = residual_block(sd, f"{p}.{i}", x, train)
    return x, i + 1


def _up_stack(sd, p, x, i, n_levels, train):
    """n_levels-1 × [Res, ConvT4s2p1, BN, LReLU], Res, conv3, Sigmoid.
    beta_vae_gans/model.py:121-131 and vq_vae_gan/model.py:184-194."""
    for _ in range(n_levels - 1):
        x = residual_block(sd, f"{p}.{i}", x, train)
        x = lrelu(bn(sd, f"{p}.{i+2}", conv_transpose2d(sd, f"{p}.{i+1}", x, 2, 1), train))
        i += 4
    x = residual_block(sd, f"{p}.{i}", x, train)
    return torch.sigmoid(conv2d(sd, f"{p}.{i+1}", x, 1, 1))


def vq_encoder(sd: SD, x, n_levels=4, train=True, p="encoder.encoder"):
    """vq_vae_gan/model.py:137-166."""
    x, i = _down_stack(sd, p, x, n_levels, train)
    return conv2d(sd, f"{p}.{i}", x, 1, 0)


def vq_decoder(sd: SD, zq, n_levels=4, train=True, p="decoder.decoder"):
    """vq_vae_gan/model.py:169-199."""
    x = lrelu(bn(sd, f"{p}.1", conv2d(sd, f"{p}.0", zq, 1, 1), train))
    return _up_stack(sd, p, x, 3, n_levels, train)


def patch_discriminator(sd: SD, x, n_levels=4, train=True, p="discriminator.model"):
    """beta_vae_gans/model.py:148-176 == vq_vae_gan/model.py:202-226."""
    x = lrelu(conv2d(sd, f"{p}.0", x, 2, 1))
    i = 2
    for _ in range(n_levels - 1):
        x = lrelu(bn(sd, f"{p}.{i+1}", conv2d(sd, f"{p}.{i}", x, 2, 1), train))
        i += 3
    return conv2d(sd, f"{p}.{i}", x, 1, 1)


def vq_distances(flat, codebook):
    """vq_vae_gan/model.py:88-92, same op order: (|z|^2 + |E|^2) - 2 z.E^T in fp32."""
    return (flat ** 2).sum(1, keepdim=True) + (codebook ** 2).sum(1) - 2 * flat @ codebook.t()


def vector_quantizer(sd: SD, z, commitment_cost=0.25, decay=0.99, training=True,
                     p="vector_quantizer"):
    """vq_vae_gan/model.py:68-134.  EMA uses the scatter form Σ_i onehot_i ⊗ z_i ==
    index_add (SURVEY.md §3.2 verified identity) instead of the [N,K,D] intermediate."""
    b, d, h, w = z.shape
    E = sd[p + ".embeddings.weight"]
    K = E.shape[0]
    flat = z.permute(0, 2, 3, 1).reshape(-1, d)
    idx = torch.argmin(vq_distances(flat, E), dim=1)              # first-min tie rule
    q = E[idx].reshape(b, h, w, d).permute(0, 3, 1, 2)
    e_latent = F.mse_loss(q.detach(), z)
    q_latent = F.mse_loss(q, z.detach())
    vq_loss = q_latent + commitment_cost * e_latent
    if training:
        with torch.no_grad():
            counts = torch.bincount(idx, minlength=K).to(flat.dtype)
            sd[p + ".ema_cluster_size"] = sd[p + ".ema_cluster_size"] * decay + (1 - decay) * counts
            esum = torch.zeros_like(E).index_add_(0, idx, flat.detach())
            sd[p + ".ema_w"] = sd[p + ".ema_w"] * decay + (1 - decay) * esum
            n = sd[p + ".ema_cluster_size"].sum()
            cs = (sd[p + ".ema_cluster_size"] + 1e-5) / (n + K * 1e-5) * n
            # reference rebinds .data: the graph built above keeps the *old* codebook values
            sd[p + ".embeddings.weight"] = (sd[p + ".ema_w"] / cs[:, None]).requires_grad_(E.requires_grad)
    zq = z + (q - z).detach()
    return zq, vq_loss, idx.reshape(b, h, w)


def vq_forward(sd: SD, x, cfg, training=True, train_bn=True):
    """VQVAEGAN.forward, vq_vae_gan/model.py:280-288."""
    nl = len(cfg["enc_channels"])
    z = vq_encoder(sd, x, nl, train_bn)
    zq, vq_loss, idx = vector_quantizer(sd, z, cfg["commitment_cost"], cfg["decay"], training)
    recon = vq_decoder(sd, zq, nl, train_bn)
    return recon, z, zq, vq_loss, idx


def hinge_d_loss(disc_real, disc_fake):
    """*/model.py loss_function, discriminator branch (vq_vae_gan/model.py:325)."""
    return F.relu(1.0 - disc_real).mean() + F.relu(1.0 + disc_fake).mean()


def vq_g_loss(original, recon, vq_loss, disc_fake, lambda_adv):
    """vq_vae_gan/model.py:306-314."""
    recon_loss = F.mse_loss(recon, original)
    adv = -disc_fake.mean()
    return recon_loss + vq_loss + lambda_adv * adv, recon_loss, adv


# --------------------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults: eps 1e-8, no weight decay, bias-corrected) and the step
# --------------------------------------------------------------------------------------
def adam_step(params: Sequence[torch.Tensor], grads: Sequence[Optional[torch.Tensor]], state: dict,
              lr: float, betas=(0.5, 0.999), eps=1e-8):
    """torch.optim.Adam single-tensor arithmetic; */main.py:34-44 construct it with
    lr=1e-4 (G) / 5e-5 (D), betas=(0.5, 0.999).  Params with grad None are skipped and do
    not advance their step counter (torch behaviour)."""
    b1, b2 = betas
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            if g is None:
                continue
            st = state.setdefault(i, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
            st["step"] += 1
            t = st["step"]
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** t
            bc2 = 1 - b2 ** t
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(st["m"], denom, value=-lr / bc1)


def _names(sd, prefixes, learnable_only=True):
    out = []
    for k in sd:
        if any(k.startswith(pf) for pf in prefixes):
            leaf = k.rsplit(".", 1)[-1]
            if leaf in ("running_mean", "running_var", "num_batches_tracked"):
                continue
            out.append(k)
    return out


def g_param_names(sd):
    return _names(sd, ("encoder.", "decoder."))


def d_param_names(sd):
    return _names(sd, ("discriminator.",))


def _set_requires(sd, names, flag):
    for k in names:
        sd[k] = sd[k].detach().requires_grad_(flag)


def vq_train_step(sd: SD, images, cfg, opt, lr_g=1e-4, lr_d=5e-5):
    """One iteration of vq_vae_gan/main.py:75-111: D step (forward, D(real), D(fake.detach()),
    hinge, Adam_D) then G step (second full forward incl. EMA + BN stat updates, D(fake) with
    the updated D, recon + vq + λ·adv, Adam_G).  ``opt`` = {"g": {}, "d": {}} Adam states.
    Returns a dict of python floats + the indices of both forwards."""
    gn, dn = g_param_names(sd), d_param_names(sd)
    _set_requires(sd, gn, True)
    _set_requires(sd, dn, True)
    # ---- D step
    recon, _, _, vq_loss, idx1 = vq_forward(sd, images, cfg)
    d_real = patch_discriminator(sd, images)
    d_fake = patch_discriminator(sd, recon.detach())
    d_loss = hinge_d_loss(d_real, d_fake)
    d_recon = F.mse_loss(recon, images).item()
    grads = torch.autograd.grad(d_loss, [sd[k] for k in dn])
    adam_step([sd[k] for k in dn], grads, opt["d"], lr_d)
    # ---- G step
    recon, _, _, vq_loss, idx2 = vq_forward(sd, images, cfg)
    d_fake = patch_discriminator(sd, recon)
    g_loss, recon_loss, adv = vq_g_loss(images, recon, vq_loss, d_fake, cfg["lambda_adv"])
    grads = torch.autograd.grad(g_loss, [sd[k] for k in gn], allow_unused=True)
    adam_step([sd[k] for k in gn], grads, opt["g"], lr_g)
    return {"d_loss": d_loss.item(), "d_recon": d_recon, "g_loss": g_loss.item(),
            "recon": recon_loss.item(), "vq": vq_loss.item(), "adv": adv.item(),
            "idx1": idx1, "idx2": idx2, "recon_img": recon.detach()}


VQ_DEFAULT_CFG = dict(embedding_dim=64, num_embeddings=512, commitment_cost=0.25, decay=0.99,
                      channels=3, enc_channels=[64, 128, 256, 512], dec_channels=[512, 256, 128, 64],
                      lambda_adv=0.1)


# --------------------------------------------------------------------------------------
# β-VAE-GAN  (beta_vae_gans/model.py)
# --------------------------------------------------------------------------------------
def beta_encoder(sd: SD, x, n_levels=4, train=True):
    """beta_vae_gans/model.py:77-93: conv features, flatten (NCHW order), fc_mu / fc_logvar,
    clamp(logvar, -20, 20)."""
    f, _ = _down_stack(sd, "encoder.features", x, n_levels, train)
    f = f.reshape(x.shape[0], -1)
    mu = F.linear(f, sd["encoder.fc_mu.weight"], sd["encoder.fc_mu.bias"])
    logvar = F.linear(f, sd["encoder.fc_logvar.weight"], sd["encoder.fc_logvar.bias"])
    return mu, torch.clamp(logvar, -20, 20)


def beta_decoder(sd: SD, z, n_levels=4, train=True, init_size=16):
    """beta_vae_gans/model.py:135-145: fc → view [B,512,16,16] → BN+LReLU → up stack."""
    p = "decoder.decoder"
    x = F.linear(z, sd["decoder.fc.weight"], sd["decoder.fc.bias"]).reshape(z.shape[0], -1, init_size, init_size)
    x = lrelu(bn(sd, f"{p}.0", x, train))
    return _up_stack(sd, p, x, 2, n_levels, train)


def reparameterize(mu, logvar, eps):
    """beta_vae_gans/model.py:211-216 / hierarchical_vae_gan/model.py:356-361 with the
    N(0,1) draw injected."""
    return mu + eps * torch.exp(0.5 * logvar)


def kl_loss(mu, logvar):
    """beta_vae_gans/model.py:258."""
    return -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())


def hier_kl_loss(mu_list, logvar_list, weights):
    """hierarchical_vae_gan/model.py:406-409."""
    tot = 0.0
    for mu, lv, w in zip(mu_list, logvar_list, weights):
        tot = tot + w * kl_loss(mu, lv)
    return tot


# --------------------------------------------------------------------------------------
# Hierarchical VAE-GAN  (hierarchical_vae_gan/model.py)
# --------------------------------------------------------------------------------------
def _hier_level_block(sd, p, x, has_attn, train):
    q = p + ".0" if has_attn else p
    x = lrelu(bn(sd, q + ".1", conv2d(sd, q + ".0", x, 2, 1), train))
    x = residual_block(sd, q + ".3", x, train)
    x = residual_block(sd, q + ".4", x, train)
    if has_attn:
        x = self_attention(sd, p + ".1", x)
    return x


def hier_encoder(sd: SD, x, num_hierarchies=3, train=True):
    """hierarchical_vae_gan/model.py:141-173."""
    x = lrelu(bn(sd, "encoder.init_conv.1", conv2d(sd, "encoder.init_conv.0", x, 1, 1), train))
    mus, lvs = [], []
    for i in range(num_hierarchies):
        x = _hier_level_block(sd, f"encoder.encoders.{i}", x, i >= 1, train)
        mus.append(conv2d(sd, f"encoder.mu_layers.{i}", x))
        lvs.append(torch.clamp(conv2d(sd, f"encoder.logvar_layers.{i}", x), -20, 20))
    return mus, lvs


def hier_decoder_block(sd: SD, j, x, has_attn, train=True):
    """One entry of HierarchicalDecoder.decoders (hierarchical_vae_gan/model.py:196-222):
    conv3+BN+LReLU+Res [+SelfAttention]."""
    p = f"decoder.decoders.{j}"
    q = p + ".0" if has_attn else p
    x = lrelu(bn(sd, q + ".1", conv2d(sd, q + ".0", x, 1, 1), train))
    x = residual_block(sd, q + ".3", x, train)
    if has_attn:
        x = self_attention(sd, p + ".1", x)
    return x


def hier_output_conv(sd: SD, x, train=True):
    """hierarchical_vae_gan/model.py:225-231."""
    p = "decoder.output_conv"
    x = lrelu(bn(sd, p + ".1", conv2d(sd, p + ".0", x, 1, 1), train))
    return torch.sigmoid(conv2d(sd, p + ".3", x, 1, 1))


def hier_decoder_nearest(sd: SD, z_list, num_hierarchies=3, train=True):
    """NON-REFERENCE composition D2: the reference's composed forward
    raises (model.py:261), so the sub-blocks are chained with nearest ×2 upsampling."""
    x = None
    for j, z in enumerate(reversed(z_list)):
        level = num_hierarchies - 1 - j
        if j > 0:
            x = torch.cat([z, F.interpolate(x, scale_factor=2, mode="nearest")], dim=1)
        else:
            x = z
        x = hier_decoder_block(sd, j, x, level >= 1, train)
    return hier_output_conv(sd, x, train)


def hier_discriminator(sd: SD, x, num_hierarchies=3, train=True):
    """hierarchical_vae_gan/model.py:275-313."""
    p = "discriminator.model"
    x = lrelu(conv2d(sd, f"{p}.0", x, 2, 1))
    i = 2
    for _ in range(1, num_hierarchies):
        x = lrelu(bn(sd, f"{p}.{i+1}", conv2d(sd, f"{p}.{i}", x, 2, 1), train))
        x = self_attention(sd, f"{p}.{i+3}", x)
        i += 4
    x = lrelu(bn(sd, f"{p}.{i+1}", conv2d(sd, f"{p}.{i}", x, 1, 1), train))
    return conv2d(sd, f"{p}.{i+3}", x, 1, 1)


# --------------------------------------------------------------------------------------
# whole steps of the two architectures whose reference step raises (documented shims D1, D2)
# --------------------------------------------------------------------------------------
BETA_DEFAULT_CFG = dict(beta=1.0, lambda_adv=0.1)
HIER_DEFAULT_CFG = dict(beta=1.0, lambda_adv=0.1, level_weights=[1.0, 0.75, 0.5], num_hierarchies=3)


def _vae_g_loss(target, recon, kl, d_fake, beta, lambda_adv):
    """beta_vae_gans/model.py:255-266 == hierarchical_vae_gan/model.py:403-417."""
    recon_loss = F.mse_loss(recon, target)
    adv = -d_fake.mean()
    return recon_loss + beta * kl + lambda_adv * adv, recon_loss, adv


def beta_train_step(sd: SD, images, eps, cfg, opt, lr_g=1e-4, lr_d=5e-5):
    """One iteration of beta_vae_gans/main.py:75-108: ONE generator forward; the G step reuses
    recon/mu/logvar and re-runs D(recon) with the updated D (BatchNorm statistics of D update a third
    time).  NON-REFERENCE shim D1: the reconstruction is 128x128, so the MSE target is
    avg_pool2d(original, 2) — with the original itself model.py:255 raises."""
    gn, dn = g_param_names(sd), d_param_names(sd)
    _set_requires(sd, gn, True)
    _set_requires(sd, dn, True)
    target = F.avg_pool2d(images, 2)
    mu, logvar = beta_encoder(sd, images)
    recon = beta_decoder(sd, reparameterize(mu, logvar, eps))
    d_real = patch_discriminator(sd, images)
    d_fake = patch_discriminator(sd, recon.detach())
    d_loss = hinge_d_loss(d_real, d_fake)
    grads = torch.autograd.grad(d_loss, [sd[k] for k in dn])
    adam_step([sd[k] for k in dn], grads, opt["d"], lr_d)
    d_fake = patch_discriminator(sd, recon)
    kl = kl_loss(mu, logvar)
    g_loss, recon_loss, adv = _vae_g_loss(target, recon, kl, d_fake, cfg["beta"], cfg["lambda_adv"])
    grads = torch.autograd.grad(g_loss, [sd[k] for k in gn], allow_unused=True)
    adam_step([sd[k] for k in gn], grads, opt["g"], lr_g)
    return {"d_loss": d_loss.item(), "g_loss": g_loss.item(), "recon": recon_loss.item(), "kl": kl.item(),
            "adv": adv.item(), "recon_img": recon.detach()}


def hier_train_step(sd: SD, images, eps_pair, cfg, opt, lr_g=1e-4, lr_d=5e-5):
    """One iteration of hierarchical_vae_gan/main.py:75-111: two generator forwards, each drawing fresh
    N(0,1) noise per level (``eps_pair`` = [draws of forward 1, draws of forward 2], level order 0,1,2).
    NON-REFERENCE shims: D2 (decoder sub-blocks chained with nearest x2 upsampling, model.py:261 raises)
    and D1 (target = avg_pool2d(original, 2): the reconstruction has half the input size)."""
    gn, dn = g_param_names(sd), d_param_names(sd)
    _set_requires(sd, gn, True)
    _set_requires(sd, dn, True)
    nh, w = cfg["num_hierarchies"], cfg["level_weights"]
    target = F.avg_pool2d(images, 2)

    def generator(eps_list):
        mus, lvs = hier_encoder(sd, images, nh)
        zs = [reparameterize(m, v, e) for m, v, e in zip(mus, lvs, eps_list)]
        return hier_decoder_nearest(sd, zs, nh), mus, lvs

    recon, mus, lvs = generator(eps_pair[0])
    d_real = hier_discriminator(sd, images, nh)
    d_fake = hier_discriminator(sd, recon.detach(), nh)
    d_loss = hinge_d_loss(d_real, d_fake)
    d_kl = hier_kl_loss(mus, lvs, w).item()
    grads = torch.autograd.grad(d_loss, [sd[k] for k in dn])
    adam_step([sd[k] for k in dn], grads, opt["d"], lr_d)
    recon, mus, lvs = generator(eps_pair[1])
    d_fake = hier_discriminator(sd, recon, nh)
    kl = hier_kl_loss(mus, lvs, w)
    g_loss, recon_loss, adv = _vae_g_loss(target, recon, kl, d_fake, cfg["beta"], cfg["lambda_adv"])
    grads = torch.autograd.grad(g_loss, [sd[k] for k in gn], allow_unused=True)
    adam_step([sd[k] for k in gn], grads, opt["g"], lr_g)
    return {"d_loss": d_loss.item(), "d_kl": d_kl, "g_loss": g_loss.item(), "recon": recon_loss.item(), "kl": kl.item(),
            "adv": adv.item(), "recon_img": recon.detach()}


# --------------------------------------------------------------------------------------
# state_dict templates (key → zero tensor of the right shape/dtype); SURVEY.md §8(b) contract
# --------------------------------------------------------------------------------------
def _t_conv(t, p, cin, cout, k):
    t[p + ".weight"] = torch.zeros(cout, cin, k, k)
    t[p + ".bias"] = torch.zeros(cout)


def _t_convT(t, p, cin, cout, k):
    t[p + ".weight"] = torch.zeros(cin, cout, k, k)
    t[p + ".bias"] = torch.zeros(cout)


def _t_bn(t, p, c):
    t[p + ".weight"] = torch.zeros(c)
    t[p + ".bias"] = torch.zeros(c)
    t[p + ".running_mean"] = torch.zeros(c)
    t[p + ".running_var"] = torch.zeros(c)
    t[p + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)


def _t_res(t, p, c):
    _t_conv(t, p + ".conv1", c, c, 3)
    _t_conv(t, p + ".conv2", c, c, 3)
    _t_bn(t, p + ".bn1", c)
    _t_bn(t, p + ".bn2", c)


def _t_attn(t, p, c):
    _t_conv(t, p + ".query", c, c // 8, 1)
    _t_conv(t, p + ".key", c, c // 8, 1)
    _t_conv(t, p + ".value", c, c, 1)
    t[p + ".gamma"] = torch.zeros(1)


def _t_down(t, p, cin, ch):
    _t_conv(t, f"{p}.0", cin, ch[0], 3)
    _t_bn(t, f"{p}.1", ch[0])
    i = 3
    for a, b in zip(ch[:-1], ch[1:]):
        _t_res(t, f"{p}.{i}", a)
        _t_conv(t, f"{p}.{i+1}", a, b, 4)
        _t_bn(t, f"{p}.{i+2}", b)
        i += 4
    _t_res(t, f"{p}.{i}", ch[-1])
    return i + 1


def _t_up(t, p, i, ch, cout):
    for a, b in zip(ch[:-1], ch[1:]):
        _t_res(t, f"{p}.{i}", a)
        _t_convT(t, f"{p}.{i+1}", a, b, 4)
        _t_bn(t, f"{p}.{i+2}", b)
        i += 4
    _t_res(t, f"{p}.{i}", ch[-1])
    _t_conv(t, f"{p}.{i+1}", ch[-1], cout, 3)


def _t_patch_disc(t, p, cin, ch):
    _t_conv(t, f"{p}.0", cin, ch[0], 4)
    i = 2
    for a, b in zip(ch[:-1], ch[1:]):
        _t_conv(t, f"{p}.{i}", a, b, 4)
        _t_bn(t, f"{p}.{i+1}", b)
        i += 3
    _t_conv(t, f"{p}.{i}", ch[-1], 1, 4)


def vq_encoder_template(cin=3, ch=(64, 128, 256, 512), d=64, p="encoder"):
    t = {}
    i = _t_down(t, p, cin, list(ch))
    _t_conv(t, f"{p}.{i}", ch[-1], d, 1)
    return t


def vq_decoder_template(d=64, ch=(512, 256, 128, 64), cout=3, p="decoder"):
    t = {}
    _t_conv(t, f"{p}.0", d, ch[0], 3)
    _t_bn(t, f"{p}.1", ch[0])
    _t_up(t, p, 3, list(ch), cout)
    return t


def patch_disc_template(cin=3, ch=(64, 128, 256, 512), p="model"):
    t = {}
    _t_patch_disc(t, p, cin, list(ch))
    return t


def vq_template(cfg=None):
    """VQVAEGAN.state_dict() layout (200 tensors at the default config)."""
    cfg = cfg or VQ_DEFAULT_CFG
    t = vq_encoder_template(cfg["channels"], cfg["enc_channels"], cfg["embedding_dim"], "encoder.encoder")
    K, D = cfg["num_embeddings"], cfg["embedding_dim"]
    t["vector_quantizer.ema_cluster_size"] = torch.zeros(K)
    t["vector_quantizer.ema_w"] = torch.zeros(K, D)
    t["vector_quantizer.embeddings.weight"] = torch.zeros(K, D)
    t.update(vq_decoder_template(D, cfg["dec_channels"], cfg["channels"], "decoder.decoder"))
    t.update(patch_disc_template(cfg["channels"], cfg["enc_channels"], "discriminator.model"))
    return t


def beta_template(latent_dim=128, ch=(64, 128, 256, 512)):
    """BetaVAEGAN.state_dict() layout (199 tensors)."""
    t = {}
    _t_down(t, "encoder.features", 3, list(ch))
    fin = ch[-1] * (256 // 2 ** (len(ch) - 1)) ** 2
    for n in ("fc_mu", "fc_logvar"):
        t[f"encoder.{n}.weight"] = torch.zeros(latent_dim, fin)
        t[f"encoder.{n}.bias"] = torch.zeros(latent_dim)
    rch = list(reversed(ch))
    fs = 256 // 2 ** len(ch)
    t["decoder.fc.weight"] = torch.zeros(rch[0] * fs * fs, latent_dim)
    t["decoder.fc.bias"] = torch.zeros(rch[0] * fs * fs)
    _t_bn(t, "decoder.decoder.0", rch[0])
    _t_up(t, "decoder.decoder", 2, rch, 3)
    _t_patch_disc(t, "discriminator.model", 3, list(ch))
    return t


def hier_template(base=64, latent_dims=(32, 64, 128), nh=3):
    """HierarchicalVAEGAN.state_dict() layout (267 tensors at the default config)."""
    t = {}
    _t_conv(t, "encoder.init_conv.0", 3, base, 3)
    _t_bn(t, "encoder.init_conv.1", base)
    cur = base
    for i in range(nh):
        lc = base * 2 ** i
        p = f"encoder.encoders.{i}" + (".0" if i >= 1 else "")
        _t_conv(t, p + ".0", cur, lc, 4)
        _t_bn(t, p + ".1", lc)
        _t_res(t, p + ".3", lc)
        _t_res(t, p + ".4", lc)
        if i >= 1:
            _t_attn(t, f"encoder.encoders.{i}.1", lc)
        cur = lc
    for i in range(nh):
        _t_conv(t, f"encoder.mu_layers.{i}", base * 2 ** i, latent_dims[i], 1)
    for i in range(nh):
        _t_conv(t, f"encoder.logvar_layers.{i}", base * 2 ** i, latent_dims[i], 1)
    for j, i in enumerate(range(nh - 1, -1, -1)):
        lc = base * 2 ** i
        cin = latent_dims[i] if i == nh - 1 else latent_dims[i] + base * 2 ** (i + 1)
        p = f"decoder.decoders.{j}" + (".0" if i >= 1 else "")
        _t_conv(t, p + ".0", cin, lc, 3)
        _t_bn(t, p + ".1", lc)
        _t_res(t, p + ".3", lc)
        if i >= 1:
            _t_attn(t, f"decoder.decoders.{j}.1", lc)
    _t_conv(t, "decoder.output_conv.0", base, base, 3)
    _t_bn(t, "decoder.output_conv.1", base)
    _t_conv(t, "decoder.output_conv.3", base, 3, 3)
    for u, i in enumerate(range(1, nh)):
        _t_convT(t, f"decoder.upsample.{u}", base * 2 ** i, base * 2 ** (i - 1), 4)
    p = "discriminator.model"
    _t_conv(t, f"{p}.0", 3, base, 4)
    cur, k = base, 2
    for i in range(1, nh):
        nc = base * 2 ** i
        _t_conv(t, f"{p}.{k}", cur, nc, 4)
        _t_bn(t, f"{p}.{k+1}", nc)
        _t_attn(t, f"{p}.{k+3}", nc)
        cur, k = nc, k + 4
    _t_conv(t, f"{p}.{k}", cur, cur * 2, 4)
    _t_bn(t, f"{p}.{k+1}", cur * 2)
    _t_conv(t, f"{p}.{k+3}", cur * 2, 1, 4)
    return t


def sub_dict(sd: SD, prefix: str) -> SD:
    """Entries under ``prefix`` with the prefix stripped (views, not copies)."""
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
