"""The ``main.py`` flag surface of the three architectures (beta_vae_gans/main.py:270-336, vq_vae_gan/main.py:293-363,
hierarchical_vae_gan/main.py:280-347) in front of the HIP training engine.

Every reference flag keeps its name, type, default and meaning: ``--output_dir --data_dir --batch_size --epochs
--lambda_adv --resume`` everywhere, ``--beta --latent_dim`` (beta), ``--embedding_dim --num_embeddings
--commitment_cost`` (VQ), ``--num_hierarchies --beta`` (hierarchical).  The flags below are additions of this engine
(SURVEY.md §5, §8c); their defaults reproduce the reference's behaviour:

  --dtype {bf16,f32}                 compute dtype of the conv / attention kernels (f32 = exact-fp32 parity mode)
  --bn_sync {local,global}           data parallel only: BatchNorm / VQ-EMA statistics per rank, or of the global batch
  --recon_target_policy {error,area} 'error' raises like beta_vae_gans/model.py:255; 'area' is shim D1
  --hier_decoder_policy {error,nearest}  'error' raises like hierarchical_vae_gan/model.py:261; 'nearest' is shim D2
  --image_size N                     Config.image_size (the reference has no flag for it)
  --num_workers N                    Config.num_workers (PNG-decoding loader processes; the reference has no flag)
  --max_iters N                      stop every epoch after N iterations (smoke runs)

One process per GPU: under ``torchrun`` (WORLD_SIZE > 1) the process group (RCCL) is initialised BEFORE anything
touches the GPU, ``--batch_size`` is the per-GPU batch, every rank reads its own shard of the dataset and gradients
are exchanged by ``train.DataParallel``.
"""
from __future__ import annotations

import argparse
import json
import os
from typing import List, Optional

ARCHS = ("beta", "vq", "hier")

_DESCRIPTION = {"beta": "Train β-VAE-GAN model", "vq": "Train VQ-VAE-GAN model", "hier": "Train Hierarchical VAE-GAN model"}
_OUTPUT_DIR = {"beta": "./output/beta_vae_gan", "vq": "./output/vq_vae_gan", "hier": "./output/hierarchical_vae_gan"}


def build_parser(arch: str) -> argparse.ArgumentParser:
    if arch not in ARCHS:
        raise ValueError(f"arch must be one of {ARCHS}")
    p = argparse.ArgumentParser(description=_DESCRIPTION[arch])
    # ---- the reference's flags (same names, types, defaults)
    p.add_argument('--output_dir', type=str, default=_OUTPUT_DIR[arch], help='Output directory')
    p.add_argument('--data_dir', type=str, default='./data', help='Data directory')
    p.add_argument('--batch_size', type=int, default=16, help='Batch size (per GPU)')
    p.add_argument('--epochs', type=int, default=100, help='Number of epochs')
    p.add_argument('--lambda_adv', type=float, default=0.1, help='Adversarial loss weight')
    p.add_argument('--resume', action='store_true', help='Resume training from latest checkpoint')
    if arch == "beta":
        p.add_argument('--beta', type=float, default=1.0, help='KL divergence weight (β parameter)')
        p.add_argument('--latent_dim', type=int, default=128, help='Latent dimension')
    elif arch == "vq":
        p.add_argument('--embedding_dim', type=int, default=64, help='Embedding dimension')
        p.add_argument('--num_embeddings', type=int, default=512, help='Number of embeddings (codebook size)')
        p.add_argument('--commitment_cost', type=float, default=0.25, help='Commitment cost for VQ-VAE')
    else:
        p.add_argument('--num_hierarchies', type=int, default=3, help='Number of hierarchical levels')
        p.add_argument('--beta', type=float, default=1.0, help='KL divergence weight (β parameter)')
    # ---- additions of this engine
    p.add_argument('--dtype', choices=('bf16', 'f32'), default='bf16', help='compute dtype of the MFMA kernels')
    p.add_argument('--bn_sync', choices=('local', 'global'), default='local',
                   help='data parallel: BatchNorm / VQ-EMA statistics per rank or of the global batch')
    p.add_argument('--recon_target_policy', choices=('error', 'area'), default='error',
                   help="half-size reconstructions: 'error' raises like the reference, 'area' compares with avg_pool2d(x, 2)")
    if arch == "hier":
        p.add_argument('--hier_decoder_policy', choices=('error', 'nearest'), default='error',
                       help="'error' raises like the reference's decoder, 'nearest' chains the sub-blocks with nearest x2")
    p.add_argument('--image_size', type=int, default=None, help='Config.image_size (default: the config value, 256)')
    p.add_argument('--num_workers', type=int, default=None, help='Config.num_workers (default: the config value, 4)')
    p.add_argument('--max_iters', type=int, default=None, help='stop each epoch after this many iterations')
    return p


def config_from_args(arch: str, args):
    """The reference's ``Config()`` + ``config.<x> = args.<x>`` block (e.g. vq_vae_gan/main.py:318-328)."""
    if arch == "beta":
        from .beta_vae_gans.config import Config
    elif arch == "vq":
        from .vq_vae_gan.config import Config
    else:
        from .hierarchical_vae_gan.config import Config
    config = Config()
    names = {"beta": ("beta", "latent_dim"), "vq": ("embedding_dim", "num_embeddings", "commitment_cost"),
             "hier": ("num_hierarchies", "beta")}[arch]
    for n in ("output_dir", "data_dir", "batch_size", "epochs", "lambda_adv") + names:
        setattr(config, n, getattr(args, n))
    config.recon_target_policy = args.recon_target_policy
    if arch == "hier":
        config.hier_decoder_policy = args.hier_decoder_policy
    if args.image_size is not None:
        config.image_size = args.image_size
    if args.num_workers is not None:
        config.num_workers = args.num_workers
    return config


def build_model(arch: str, config):
    if arch == "beta":
        from .beta_vae_gans.model import BetaVAEGAN
        return BetaVAEGAN(config)
    if arch == "vq":
        from .vq_vae_gan.model import VQVAEGAN
        return VQVAEGAN(config)
    from .hierarchical_vae_gan.model import HierarchicalVAEGAN
    return HierarchicalVAEGAN(config)


class ConsoleLogger:
    """Stand-in for beta_vae_gans/utils.py MetricLogger (TensorBoard, out of scope): same methods and ``log_dir``;
    keeps the scalars and writes them to ``<log_dir>/train_metrics.json`` / ``eval_metrics.json``."""

    def __init__(self, log_dir, config=None, quiet=False):
        self.log_dir, self.quiet = log_dir, quiet
        self.train_metrics, self.eval_metrics = [], []
        os.makedirs(os.path.join(log_dir, 'checkpoints'), exist_ok=True)

    def log_train_metrics(self, epoch, iteration, loss_dict, bpp, images=None, reconstructed=None):
        row = {"epoch": epoch, "iteration": iteration, "bpp": float(bpp)}
        row.update({k: float(v) for k, v in loss_dict.items()})
        self.train_metrics.append(row)
        if not self.quiet:
            print(f"[{epoch}:{iteration}] " + " ".join(f"{k}={v:.4f}" for k, v in row.items() if k not in ("epoch", "iteration")))

    def log_eval_metrics(self, epoch, psnr, ssim, lpips, fid, bpp, images=None, reconstructed=None):
        self.eval_metrics.append({"epoch": epoch, "psnr": float(psnr), "ssim": float(ssim), "lpips": float(lpips),
                                  "fid": float(fid), "bpp": float(bpp)})

    def save_metrics(self):
        for name, rows in (("train_metrics.json", self.train_metrics), ("eval_metrics.json", self.eval_metrics)):
            with open(os.path.join(self.log_dir, name), "w") as f:
                json.dump(rows, f)

    def close(self):
        pass


def _init_distributed():
    """torchrun: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  Must run before any GPU call."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    import torch
    import torch.distributed as dist
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        dist.init_process_group(os.environ.get("VG_DIST_BACKEND", "nccl"), rank=rank, world_size=world,
                                device_id=torch.device("cuda", local) if os.environ.get("VG_DIST_BACKEND", "nccl") == "nccl" else None)
    return rank, world, local


def main(arch: str, argv: Optional[List[str]] = None):
    args = build_parser(arch).parse_args(argv)
    rank, world, local = _init_distributed()
    import torch
    from . import loop, ops
    from . import data as D
    from .train import DataParallel
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP training engine needs an MI355X (there is no CPU fallback)")
    torch.cuda.set_device(local)
    ops.set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    config = config_from_args(arch, args)
    os.makedirs(os.path.join(config.output_dir, 'checkpoints'), exist_ok=True)
    D.set_seed(config.seed)
    logger = ConsoleLogger(config.output_dir, config, quiet=rank != 0)
    train_loader = D.get_dataloader(config, split='train', rank=rank, world=world)
    val_loader = D.get_kodak_dataset(config)
    model = build_model(arch, config)
    if args.resume:
        latest = os.path.join(config.output_dir, 'checkpoints', 'latest.pth')
        if os.path.isfile(latest):
            epoch, _ = loop.load_checkpoint(latest, model)        # model only, like <arch>/main.py:320-328
            print(f"Resuming from epoch {epoch + 1}")
        else:
            print("No checkpoint found, starting from scratch")
    dp = DataParallel(bn_sync=args.bn_sync) if world > 1 else None
    try:
        model = loop.train(model, train_loader, val_loader, config, logger, dp=dp, device=f"cuda:{local}",
                           max_iters=args.max_iters)
    finally:
        if dp is not None:
            dp.close()
    logger.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    print("Training completed!")
    return model
