"""The GAN training step of <arch>/main.py:68-111 on the HIP engine, single GPU or data parallel.

One process per GPU; gradients of each optimizer live in one flat buffer (``optim.FlatAdam``)
that is all-reduced in a few large slices on a side stream while nothing else waits on the host.
"""
from __future__ import annotations

import contextlib
import os
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
    """Data parallelism for one process per GPU (RCCL over xGMI on GPUs, gloo in the CPU tests).

    Gradient exchange.  Every optimizer keeps its gradients in one flat buffer (``FlatAdam.flat_g``), so a bucket is
    a contiguous slice of it: no packing copies.  ``mark(x, opt, params)`` puts a boundary on the tensor that enters
    a group of layers; when the backward pass reaches it, the gradient kernels of that group have all been enqueued
    and the slice that holds ``params`` is all-reduced on the communication stream, in ``bucket_bytes`` pieces,
    while the rest of the backward pass still runs on the compute stream.  ``reduce(opt)`` exchanges whatever has
    not been sent, makes the compute stream wait for all of it and returns 1/world, which the Adam kernel applies
    (the sum is never divided in memory).

    Statistics.  ``bn_sync='local'``: each rank normalises with the BatchNorm statistics of its own images and
    feeds its own VQ-EMA counts (DistributedDataParallel semantics).  ``bn_sync='global'``: the statistics of the
    whole batch are used on every rank (the sums are added over the ranks before they are finalised, forward and
    backward), which is what the single-device reference computes at the global batch size."""

    def __init__(self, group=None, bucket_bytes: int = 32 << 20, bn_sync: str = "local", slice_bytes: Optional[int] = None,
                 exchange: Optional[str] = None):
        if bn_sync not in ("local", "global"):
            raise ValueError("bn_sync must be 'local' or 'global'")
        exchange = exchange or os.environ.get("VG_DP_EXCHANGE", "allreduce")
        if exchange not in ("allreduce", "rsag"):
            raise ValueError("exchange (VG_DP_EXCHANGE) must be 'allreduce' or 'rsag'")
        # 'allreduce': one all-reduce per bucket (RCCL picks ring / tree).  'rsag': reduce-scatter + all-gather on the
        # same bucket, in place: over the 7-link xGMI mesh every rank then reduces its own 1/world shard and a link
        # carries bucket/world bytes per phase instead of the ring's 2*(world-1)/world * bucket through one link
        # (SURVEY.md §5).  Both give the same sums; the first 8-GPU run A/Bs them with this switch.
        self.exchange = exchange
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._comm_world, self._comm_rank = self.world, self.rank      # the communicator's own size (tests overwrite .world)
        self.bucket_elems = max(1, (slice_bytes or bucket_bytes) // 4)
        self.bn_sync = bn_sync
        self._sent = {}          # id(opt) -> list of (lo, hi) element ranges already exchanged in this phase
        self._work = []
        self._comm = None        # side stream for the collectives (GPU only)
        if self.world > 1 and bn_sync == "global":
            ops.STAT_SYNC[0] = self
        self.n_async = 0         # buckets that were launched from a backward boundary (tests / diagnostics)
        self.stat_bytes = 0      # bytes of statistics exchanged (bn_sync='global'): tests hold them to the SURVEY figure
        self.stat_calls = 0
        # gloo (CPU tests, or several ranks sharing one GPU in the GPU test) moves device tensors through the host
        self._staged = dist.is_initialized() and dist.get_backend(group) == "gloo"

    # ---- global-batch statistics (ops.STAT_SYNC interface)
    def sum_stats(self, stats: torch.Tensor, nslots: int) -> torch.Tensor:
        """Sum a [replicas][nslots] statistics buffer (BatchNorm forward / backward sums) over the ranks, in place.
        What travels is ONE f64 per accumulator (``vg_stats_collapse``; nslots = 2 * cpad: the SURVEY section 5 figure
        of 2 x C values per layer and direction), not the 32 replicas / 22 fixed-point bins it is spread over on the
        device (32x the bytes in round 2); ``vg_stats_load`` puts the summed vector back in the buffer's own format."""
        from . import lib as L
        lib = L.load()
        vec = torch.empty(nslots, dtype=torch.float64, device=stats.device)
        L.check(lib.vg_stats_collapse(stats.data_ptr(), nslots, vec.data_ptr(), L.stream()), "stats_collapse")
        self.sum_(vec)
        L.check(lib.vg_stats_load(vec.data_ptr(), nslots, stats.data_ptr(), L.stream()), "stats_load")
        return stats

    def sum_(self, t: torch.Tensor) -> torch.Tensor:
        self.stat_bytes += t.numel() * t.element_size()
        self.stat_calls += 1
        if self.world > 1:
            if self._staged and t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def close(self):
        if ops.STAT_SYNC[0] is self:
            ops.STAT_SYNC[0] = None

    # ---- parameters
    def broadcast_model(self, model):
        if self.world == 1:
            return
        for t in list(model.parameters()) + list(model.buffers()):
            if self._staged and t.is_cuda:
                h = t.data.cpu()
                dist.broadcast(h, src=0, group=self.group)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src=0, group=self.group)
        # the copies went through .data: neither torch's version counter nor the optimizers' epoch cells saw them, so
        # packed weight images cached by an earlier forward would be stale
        ops.PARAM_EPOCH[0] += 1
        for p in model.parameters():
            cell = getattr(p, "_vg_epoch", None)
            if cell is not None and cell is not ops.PARAM_EPOCH:
                cell[0] += 1

    # ---- gradient buckets
    @staticmethod
    def span(opt: FlatAdam, params) -> tuple:
        """Element range [lo, hi) of ``opt.flat_g`` that holds exactly the gradients of ``params``
        (they must be adjacent in the optimizer's parameter order)."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(opt.params) if id(p) in ids]
        if not idx:
            return (0, 0)
        if idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError("bucket parameters are not contiguous in the optimizer's flat buffer")
        lo = opt.offsets[idx[0]]
        hi = opt.offsets[idx[-1] + 1] if idx[-1] + 1 < len(opt.params) else opt.numel
        return (lo, hi)

    def _exchange(self, t: torch.Tensor):
        """Sum the contiguous 1-D bucket ``t`` over the ranks, in place; appends the pending work handles."""
        w, r = self._comm_world, self._comm_rank
        n = t.numel()
        m = n // w * w if self.exchange == "rsag" else 0
        if m:
            c = m // w
            shard = t[r * c:(r + 1) * c]
            if dist.get_backend(self.group) == "gloo":      # gloo has no reduce-scatter: every rank reduces its own shard
                for dst in range(w):
                    dist.reduce(t[dst * c:(dst + 1) * c], dst=dist.get_global_rank(self.group, dst) if self.group else dst,
                                op=dist.ReduceOp.SUM, group=self.group)
                self._work.append(dist.all_gather_into_tensor(t[:m], shard.clone(), group=self.group, async_op=True))
            else:                                            # RCCL: both in place (shard = its slot of the bucket)
                self._work.append(dist.reduce_scatter_tensor(shard, t[:m], op=dist.ReduceOp.SUM, group=self.group,
                                                             async_op=True))
                self._work.append(dist.all_gather_into_tensor(t[:m], shard, group=self.group, async_op=True))
        if m < n:
            self._work.append(dist.all_reduce(t[m:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _send(self, opt, lo: int, hi: int):
        if hi <= lo or self.world == 1:
            return
        flat = opt.flat_g
        self._sent.setdefault(id(opt), []).append((lo, hi))
        if flat.is_cuda and self._staged:
            h = flat[lo:hi].cpu()                # several ranks on one GPU (tests): through the host
            for s in range(0, hi - lo, self.bucket_elems):
                self._exchange(h[s:min(s + self.bucket_elems, hi - lo)])
            for w in self._work:
                w.wait()
            self._work.clear()
            flat[lo:hi].copy_(h)
        elif flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record()                          # every gradient kernel enqueued so far on the compute stream
            self._comm.wait_event(ev)
            with torch.cuda.stream(self._comm):
                for s in range(lo, hi, self.bucket_elems):
                    self._exchange(flat[s:min(s + self.bucket_elems, hi)])
        else:
            for s in range(lo, hi, self.bucket_elems):
                self._exchange(flat[s:min(s + self.bucket_elems, hi)])

    def mark(self, x: torch.Tensor, opt: FlatAdam, params) -> torch.Tensor:
        """Boundary on ``x``, the input of the layers that own ``params``: their slice is exchanged as soon as the
        backward pass gets here."""
        if self.world == 1:
            return x
        lo, hi = self.span(opt, list(params))

        def fire():
            self.n_async += 1
            self._send(opt, lo, hi)
        return ops.grad_boundary(x, fire)

    def reduce(self, opt: FlatAdam) -> float:
        if self.world == 1:
            return 1.0
        done = sorted(self._sent.get(id(opt), []))
        pos = 0
        for lo, hi in done + [(opt.flat_g.numel(), opt.flat_g.numel())]:
            if lo > pos:
                self._send(opt, pos, lo)
            pos = max(pos, hi)
        for w in self._work:
            w.wait()                              # GPU: the compute stream waits for the collective, the host does not
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)
        self._work.clear()
        self._sent[id(opt)] = []
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
            if dp:      # decoder gradients are complete when the backward pass reaches z_q: exchange them under the
                z_q = dp.mark(z_q, opt_g, model.decoder.parameters())      # encoder's backward
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
        if dp:      # G-phase backward order: decoder, the two 268 MB Linear layers, the conv features.  Each slice is
            # exchanged as soon as it is complete, under the backward pass of what comes before it in the network.
            enc = model.encoder
            enc.grad_boundary = lambda f: dp.mark(f, opt_g, list(enc.fc_mu.parameters()) + list(enc.fc_logvar.parameters()))
            try:
                z, mu, logvar = model.encode(images, eps)
            finally:
                enc.grad_boundary = None        # the closure holds dp and opt_g: never leave it on the module
            recon = model.decode(dp.mark(z, opt_g, model.decoder.parameters()))
        else:
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
            z_list = [model.reparameterize(m, v, e) for m, v, e in zip(mu_list, logvar_list, eps)]
            if dp:      # the deepest latent enters the decoder first, so its gradient is the decoder's last
                z_list[-1] = dp.mark(z_list[-1], opt_g, model.decoder.parameters())
            recon = model.decode(z_list)
        else:
            recon, _, mu_list, logvar_list = model(images, eps_lists[1])
        with (frozen(model.discriminator) if elide_d_wgrad else contextlib.nullcontext()):
            disc_fake = model.discriminator(recon)
            g_loss = model.loss_function(images, recon, mu_list, logvar_list, disc_real, disc_fake, train_generator=True)
            g_loss['total'].backward()
    opt_g.step(dp.reduce(opt_g) if dp else 1.0)
    return (d_loss, g_loss, recon) if return_recon else (d_loss, g_loss)
