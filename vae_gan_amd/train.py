"""The GAN training step of <arch>/main.py:68-111 on the HIP engine, single GPU or data parallel.

One process per GPU; gradients of each optimizer live in one flat buffer (``optim.FlatAdam``)
that is all-reduced in a few large slices on a side stream while nothing else waits on the host.
"""
from __future__ import annotations

import contextlib
from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from .optim import FlatAdam, StepLR


def build_optimizers(model, config):
    """<arch>/main.py:34-48: Adam(encoder+decoder, lr) and Adam(discriminator, lr/2), betas from the config."""
    opt_g = FlatAdam(list(model.encoder.parameters()) + list(model.decoder.parameters()), lr=config.lr,
                     betas=(config.beta1, config.beta2))
    opt_d = FlatAdam(model.discriminator.parameters(), lr=config.lr / 2, betas=(config.beta1, config.beta2))
    return opt_g, opt_d, StepLR(opt_g, 30, 0.5), StepLR(opt_d, 30, 0.5)


class DataParallel:
    """Gradient exchange for one process per GPU.  ``reduce(opt)`` averages ``opt.flat_g`` over the
    group in ``n_slices`` contiguous all-reduces (RCCL over xGMI on GPUs, gloo in the CPU tests)."""

    def __init__(self, group=None, slice_bytes: int = 64 << 20):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.slice_elems = max(1, slice_bytes // 4)

    def broadcast_model(self, model):
        if self.world == 1:
            return
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0, group=self.group)

    def reduce(self, opt: FlatAdam):
        if self.world == 1:
            return 1.0
        flat = opt.flat_g
        handles = []
        for s in range(0, flat.numel(), self.slice_elems):
            handles.append(dist.all_reduce(flat[s:s + self.slice_elems], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        return 1.0 / self.world      # folded into the Adam kernel as grad_scale


@contextlib.contextmanager
def frozen(module):
    """Disable parameter gradients of ``module`` (the discriminator during the generator step: the
    reference computes those weight gradients and throws them away at the next zero_grad)."""
    flags = [p.requires_grad for p in module.parameters()]
    for p in module.parameters():
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, f in zip(module.parameters(), flags):
            p.requires_grad_(f)


def vq_train_step(model, images, opt_g: FlatAdam, opt_d: FlatAdam, dp: Optional[DataParallel] = None,
                  elide_d_wgrad: bool = True, share_encoder: bool = True):
    """vq_vae_gan/main.py:75-111.  Returns (d_loss_dict, g_loss_dict, indices) with device scalars.

    The reference runs the generator twice per iteration (:78 and :99).  Between the two passes only the
    discriminator changes, so the two encoder passes are bit-identical (same images, same weights, train-mode
    BatchNorm): with ``share_encoder`` the encoder runs once, taped, in the D phase, its BatchNorm buffers book
    both updates (``ops.bn_updates(2)``), and the G phase starts from that ``z``.  The quantiser (EMA moved the
    codebook in between) and the decoder run twice like the reference.  ``share_encoder=False`` is the literal
    schedule."""
    # ---- discriminator step
    opt_d.zero_grad()
    z = None
    with ops.step_scope(images.device):
        if share_encoder:
            with ops.bn_updates(2):
                z = model.encoder(images)                       # taped: its backward runs in the G phase
            with torch.no_grad():  # the D loss only sees recon.detach(): no tape for quantiser + decoder
                z_q, vq_loss, _ = model.vector_quantizer(z.detach(), training=True)
                recon = model.decoder(z_q)
        else:
            with torch.no_grad():
                recon, _, _, vq_loss, _ = model(images)
        disc_real = model.discriminator(images)
        disc_fake = model.discriminator(recon)
        d_loss = model.loss_function(images, recon, vq_loss, disc_real, disc_fake, train_generator=False)
        d_loss['total'].backward()
    opt_d.step(dp.reduce(opt_d) if dp else 1.0)
    # ---- generator step (second forward: EMA + BatchNorm statistics update again)
    opt_g.zero_grad()
    with ops.step_scope(images.device):
        if share_encoder:
            z_q, vq_loss, indices = model.vector_quantizer(z, training=True)
            recon = model.decoder(z_q)
        else:
            recon, _, _, vq_loss, indices = model(images)
        with (frozen(model.discriminator) if elide_d_wgrad else contextlib.nullcontext()):
            disc_fake = model.discriminator(recon)
            g_loss = model.loss_function(images, recon, vq_loss, disc_real, disc_fake, train_generator=True)
            g_loss['total'].backward()
    opt_g.step(dp.reduce(opt_g) if dp else 1.0)
    return d_loss, g_loss, indices


def beta_train_step(model, images, opt_g: FlatAdam, opt_d: FlatAdam, dp: Optional[DataParallel] = None, eps=None,
                    elide_d_wgrad: bool = True, return_recon: bool = False):
    """beta_vae_gans/main.py:75-108: ONE generator forward per iteration; the G step reuses recon/mu/logvar and
    only re-runs D(recon).  With the reference's shapes loss_function raises (128 vs 256) unless the model was
    built with recon_target_policy='area' (shim D1)."""
    opt_d.zero_grad()
    with ops.step_scope(images.device):
        recon, _, mu, logvar = model(images, eps)
        disc_real = model.discriminator(images)
        disc_fake = model.discriminator(recon.detach())
        d_loss = model.loss_function(images, recon, mu, logvar, disc_real, disc_fake, train_generator=False)
        d_loss['total'].backward()
        opt_d.step(dp.reduce(opt_d) if dp else 1.0)
        opt_g.zero_grad()
        with (frozen(model.discriminator) if elide_d_wgrad else contextlib.nullcontext()):
            disc_fake = model.discriminator(recon)
            g_loss = model.loss_function(images, recon, mu, logvar, disc_real, disc_fake, train_generator=True)
            g_loss['total'].backward()
    opt_g.step(dp.reduce(opt_g) if dp else 1.0)
    return (d_loss, g_loss, recon) if return_recon else (d_loss, g_loss)


def hier_train_step(model, images, opt_g: FlatAdam, opt_d: FlatAdam, dp: Optional[DataParallel] = None, eps_lists=(None, None),
                    elide_d_wgrad: bool = True, share_encoder: bool = True, return_recon: bool = False):
    """hierarchical_vae_gan/main.py:75-111: two generator forwards per iteration (fresh N(0,1) draws for each of the
    three levels in both).  Needs hier_decoder_policy='nearest' + recon_target_policy='area' (shims D2 + D1): with the
    reference's own composition the decoder raises.  ``share_encoder``: as in ``vq_train_step`` (the two encoder
    passes are bit-identical; the reparameterisation draws and the decoder are not and run twice)."""
    opt_d.zero_grad()
    mu_list = logvar_list = None
    with ops.step_scope(images.device):
        if share_encoder:
            with ops.bn_updates(2):
                mu_list, logvar_list = model.encoder(images)    # taped: its backward runs in the G phase
            with torch.no_grad():
                mu_d, lv_d = [m.detach() for m in mu_list], [v.detach() for v in logvar_list]
                eps = eps_lists[0] or [None] * len(mu_d)
                recon = model.decode([model.reparameterize(m, v, e) for m, v, e in zip(mu_d, lv_d, eps)])
        else:
            with torch.no_grad():  # the D loss only sees recon.detach(): no tape for the generator pass
                recon, _, mu_d, lv_d = model(images, eps_lists[0])
        disc_real = model.discriminator(images)
        disc_fake = model.discriminator(recon)
        d_loss = model.loss_function(images, recon, mu_d, lv_d, disc_real, disc_fake, train_generator=False)
        d_loss['total'].backward()
    opt_d.step(dp.reduce(opt_d) if dp else 1.0)
    opt_g.zero_grad()
    with ops.step_scope(images.device):
        if share_encoder:
            eps = eps_lists[1] or [None] * len(mu_list)
            recon = model.decode([model.reparameterize(m, v, e) for m, v, e in zip(mu_list, logvar_list, eps)])
        else:
            recon, _, mu_list, logvar_list = model(images, eps_lists[1])
        with (frozen(model.discriminator) if elide_d_wgrad else contextlib.nullcontext()):
            disc_fake = model.discriminator(recon)
            g_loss = model.loss_function(images, recon, mu_list, logvar_list, disc_real, disc_fake, train_generator=True)
            g_loss['total'].backward()
    opt_g.step(dp.reduce(opt_g) if dp else 1.0)
    return (d_loss, g_loss, recon) if return_recon else (d_loss, g_loss)
