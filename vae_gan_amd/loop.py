"""The rest of ``train()`` around the step: epoch loop, StepLR, PSNR evaluation on the
device, and checkpoints in the reference's layout.

Mirrors ``<arch>/main.py``: Adam x2 + StepLR(30, 0.5) (:34-48), one scheduler step per epoch (:156-157), evaluation
and checkpoints every ``save_interval`` epochs and after the last one (:169-201): ``best_model.pth`` is the bare
``state_dict`` of the best-PSNR model, ``latest.pth`` holds ``epoch / model_state_dict / optimizer_{G,D}_state_dict
/ scheduler_{G,D}_state_dict / best_psnr``.  ``--resume`` in the reference restores ONLY ``model_state_dict``
(:320-328: optimizers, schedulers and the epoch counter are saved but never reloaded); ``load_checkpoint`` does the
same by default and restores everything with ``full=True``.

Bits per pixel (VQ model) are logged like vq_vae_gan/main.py:113-131: every 50th iteration the first image of the
batch goes through ``compress_image`` (encoder + quantiser in eval mode + the HIP arithmetic coder, vq_vae_gan/codec.py)
and its true ``metadata['bpp']`` is taken; the iterations in between use the closed-form ``log2(K)`` estimate; the
evaluation reports the true BPP of every validation image (:249).

Out of scope here, as in DESIGN.md: SSIM / LPIPS / FID (third-party networks) and TensorBoard.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Iterable, Optional

import torch

from . import ops
from . import train as T


def psnr(original: torch.Tensor, reconstructed: torch.Tensor) -> float:
    """beta_vae_gans/utils.py:41-55 (skimage peak_signal_noise_ratio, data_range 1): both clipped to [0,1],
    10*log10(1/mse) per image, mean over the batch.  The squared-error sums are the HIP reduction kernel."""
    if original.shape != reconstructed.shape:
        raise RuntimeError(f"psnr: shapes differ {tuple(original.shape)} vs {tuple(reconstructed.shape)}")
    a, b = ops.clamp(original.detach(), 0.0, 1.0), ops.clamp(reconstructed.detach(), 0.0, 1.0)
    vals = [ops.mse(a[i:i + 1], b[i:i + 1]) for i in range(a.shape[0])]
    mses = torch.stack(vals).double().cpu()
    return float((10.0 * torch.log10(1.0 / mses.clamp_min(1e-20))).mean())


def vq_bpp_estimate(indices: torch.Tensor, num_embeddings: int, image_size: int) -> float:
    """vq_vae_gan/main.py:121-131: log2(K) bits per index over the pixels*3 of one image."""
    return (indices.shape[1] * indices.shape[2] * math.log2(num_embeddings)) / (image_size * image_size * 3)


def vq_true_bpp(model, images: torch.Tensor, device="cuda") -> float:
    """``compress_image(model, images)['bpp']`` (vq_vae_gan/main.py:115-118): real bytes out of the HIP arithmetic
    coder.  compress_image switches the model to eval mode (model.py:613); the caller's mode is restored."""
    from .vq_vae_gan.codec import compress_image
    was_training = model.training
    try:
        with torch.no_grad():
            _, metadata = compress_image(model, images, device)
    finally:
        model.train(was_training)
    return float(metadata['bpp'])


def evaluate(model, loader: Iterable[torch.Tensor], device="cuda", with_bpp: bool = False):
    """Eval-mode forward (running-statistics BatchNorm, no EMA update) + mean PSNR over the loader
    (<arch>/main.py evaluate(), PSNR only).  For the architectures whose reconstruction is half-size the target
    is the model's documented 'area' target.  ``with_bpp``: returns (psnr, bpp) with the mean true BPP of the VQ
    model (vq_vae_gan/main.py:249-250; 0.0 for the other architectures)."""
    was_training = model.training
    model.eval()
    vals, bpps = [], []
    is_vq = hasattr(model, "vector_quantizer")
    with torch.no_grad():
        for images in loader:
            images = images.to(device)
            out = model(images, training=False) if is_vq else model(images)
            recon = out[0]
            target = images
            if recon.shape != images.shape and getattr(model, "recon_target_policy", "error") == "area":
                target = ops.avg_pool2(images)
            vals.append(psnr(target, recon))
            if with_bpp and is_vq:
                bpps.append(vq_true_bpp(model, images, device))
    model.train(was_training)
    value = float(sum(vals) / max(len(vals), 1))
    return (value, float(sum(bpps) / max(len(bpps), 1))) if with_bpp else value


def save_checkpoint(path, epoch, model, opt_g, opt_d, sch_g, sch_d, best_psnr):
    """``latest.pth`` of <arch>/main.py:193-201 (same keys; optimizer states in torch.optim.Adam's layout)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({
        "epoch": epoch,
        "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
        "optimizer_G_state_dict": opt_g.state_dict(),
        "optimizer_D_state_dict": opt_d.state_dict(),
        "scheduler_G_state_dict": sch_g.state_dict(),
        "scheduler_D_state_dict": sch_d.state_dict(),
        "best_psnr": best_psnr,
    }, path)


def load_checkpoint(path, model, opt_g=None, opt_d=None, sch_g=None, sch_d=None, full: bool = False):
    """``--resume`` (<arch>/main.py:320-328): a bare state_dict or a ``latest.pth`` dict; only the model is restored
    unless ``full``.  Returns (epoch, best_psnr) of the file (0, 0.0 for a bare state_dict)."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if "model_state_dict" not in ck:
        model.load_state_dict(ck)
        return 0, 0.0
    model.load_state_dict(ck["model_state_dict"])
    if full:
        for obj, key in ((opt_g, "optimizer_G_state_dict"), (opt_d, "optimizer_D_state_dict"),
                         (sch_g, "scheduler_G_state_dict"), (sch_d, "scheduler_D_state_dict")):
            if obj is not None:
                obj.load_state_dict(ck[key])
    return int(ck["epoch"]), float(ck["best_psnr"])


def train(model, train_loader, val_loader, config, logger=None, step_fn: Optional[Callable] = None, dp=None,
          device="cuda", max_iters: Optional[int] = None, bpp_interval: int = 50):
    """<arch>/main.py train(): returns the trained model; side effects: ``checkpoints/best_model.pth`` and
    ``checkpoints/latest.pth`` under ``logger.log_dir`` (or ``config.output_dir``).  ``dp``: a ``train.DataParallel``;
    the replicas start from rank 0's parameters and buffers."""
    model = model.to(device)
    if dp is not None:
        dp.broadcast_model(model)
    opt_g, opt_d, sch_g, sch_d = T.build_optimizers(model, config)
    is_vq = hasattr(model, "vector_quantizer")
    if step_fn is None:
        step_fn = (T.vq_train_step if hasattr(model, "vector_quantizer") else
                   T.hier_train_step if hasattr(model, "num_hierarchies") else T.beta_train_step)
    out_dir = os.path.join(getattr(logger, "log_dir", None) or config.output_dir, "checkpoints")
    best = 0.0
    for epoch in range(config.epochs):
        model.train()
        if hasattr(train_loader, "set_epoch"):      # multi-rank runs: a fresh shuffle of the shards every epoch
            train_loader.set_epoch(epoch)
        sums, n, bpps = {}, 0, []
        for i, images in enumerate(train_loader):
            if max_iters is not None and i >= max_iters:
                break
            images = images.to(device)
            out = step_fn(model, images, opt_g, opt_d, dp)
            d_loss, g_loss = out[0], out[1]
            for k, v in list(g_loss.items()) + [("disc", d_loss["disc"])]:        # device scalars: no host sync per step
                if torch.is_tensor(v):
                    sums[k] = sums.get(k, 0.0) + v.detach()
            n += 1
            bpp = 0.0
            if is_vq:                                   # vq_vae_gan/main.py:113-131
                if i % bpp_interval == 0 and images.shape[0] > 0:
                    bpp = vq_true_bpp(model, images[:1], device)
                else:
                    bpp = vq_bpp_estimate(out[2], config.num_embeddings, config.image_size)
                bpps.append(bpp)
            if logger is not None and i % config.log_interval == 0 and hasattr(logger, "log_train_metrics"):
                logger.log_train_metrics(epoch, i, {k: float(v) if torch.is_tensor(v) else v for k, v in g_loss.items()}, bpp)
        sch_g.step()
        sch_d.step()
        avg = {k: float(v) / max(n, 1) for k, v in sums.items()}
        print(f"Epoch {epoch + 1}/{config.epochs} - G Loss: {avg.get('total', float('nan')):.4f}, "
              f"D Loss: {avg.get('disc', float('nan')):.4f}" + (f", BPP: {sum(bpps) / len(bpps):.4f}" if bpps else ""))
        if (epoch + 1) % config.save_interval == 0 or epoch == config.epochs - 1:
            value, eval_bpp = evaluate(model, val_loader, device, with_bpp=True) if val_loader is not None else (0.0, 0.0)
            print(f"Validation - PSNR: {value:.2f}" + (f", BPP: {eval_bpp:.4f}" if is_vq else ""))
            if logger is not None and hasattr(logger, "log_eval_metrics"):
                logger.log_eval_metrics(epoch, value, 0.0, 0.0, 0.0, eval_bpp)
            if (not dp or dp.rank == 0):
                os.makedirs(out_dir, exist_ok=True)
                if value > best:
                    best = value
                    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(out_dir, "best_model.pth"))
                save_checkpoint(os.path.join(out_dir, "latest.pth"), epoch, model, opt_g, opt_d, sch_g, sch_d, best)
    if logger is not None and hasattr(logger, "save_metrics"):
        logger.save_metrics()
    return model
