"""Input pipeline with the transform on the MI355X — drop-in for beta_vae_gans/data.py.

Same names as the reference: ``set_seed``, ``ImageDataset``, ``get_dataloader(config, split)``,
``get_kodak_dataset(config)``.  What changes is where the work happens: loader workers only DECODE the PNG (Pillow,
as in the reference) and hand over the uint8 HWC image; flips, RandomResizedCrop, the Pillow-bilinear resize and
ToTensor (data.py:107-119) run as two HIP kernels on a whole batch of differently-sized images
(csrc/augment.hip, ``vg_augment_batch``), bit-identical to Pillow's resize + ToTensor.  Host->device traffic is the
decoded bytes (3 B/pixel) instead of float32 tensors.  There is no CPU fallback for the transform.

torchvision is not needed.  The random draws restate torchvision's RandomHorizontalFlip / RandomVerticalFlip /
RandomResizedCrop.get_params (same distributions, same order, torch's global RNG); the reference draws them inside
its 4 worker processes, whose RNG streams are not reproducible across runs either.
"""
from __future__ import annotations

import ctypes as C
import glob
import math
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import lib as L
from .ops import _require_cuda


def set_seed(seed):
    """data.py:14-21."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


# ---- the random draws (torchvision.transforms semantics) ---------------------------------------------------------
def random_resized_crop_params(height, width, scale=(0.8, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """RandomResizedCrop.get_params -> (i, j, h, w)."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1], generator=generator).item()
        aspect = math.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=generator).item())
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,), generator=generator).item()
            j = torch.randint(0, width - w + 1, size=(1,), generator=generator).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def draw_train_params(height, width, generator=None):
    """data.py:108-110 in Compose order: (flip_h, flip_v, i, j, h, w)."""
    flip_h = bool(torch.rand(1, generator=generator) < 0.5)
    flip_v = bool(torch.rand(1, generator=generator) < 0.5)
    return (flip_h, flip_v) + tuple(random_resized_crop_params(height, width, generator=generator))


# ---- the device transform -----------------------------------------------------------------------------------------
class DeviceTransform:
    """Batch transform on the GPU.  ``train=True``: data.py:107-112; ``train=False``: data.py:116-119
    (Resize((S, S)) + ToTensor).  ``image_size=None``: ToTensor only (get_kodak_dataset, data.py:146-148: the output
    has the image's own size, and the images of a batch must share it)."""

    def __init__(self, image_size, train=True, generator=None):
        self.image_size, self.train, self.generator = image_size, train, generator
        self._ws = None

    def __call__(self, images, params=None):
        """images: list of uint8 HWC tensors (device or host: host images are uploaded as bytes).  params: optional
        list of (flip_h, flip_v, i, j, h, w) overriding the random draws.  Returns float32 [B, 3, S, S] on the device."""
        lib = L.load()
        dev = [im.cuda(non_blocking=True) if not im.is_cuda else im for im in images]
        n = len(dev)
        items = (L.AugItem * n)()
        oh = ow = self.image_size
        for b, im in enumerate(dev):
            _require_cuda(im, "DeviceTransform")
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous():
                raise ValueError(f"image {b}: expected a contiguous uint8 HxWx3 tensor, got {im.dtype} {tuple(im.shape)}")
            h, w = int(im.shape[0]), int(im.shape[1])
            if params is not None:
                p = params[b]
            elif self.train:
                p = draw_train_params(h, w, self.generator)
            else:
                p = (False, False, 0, 0, h, w)
            if self.image_size is None:
                oh, ow = h, w
                if any(tuple(t.shape) != tuple(im.shape) for t in dev):
                    raise ValueError("image_size=None (ToTensor only) needs the images of a batch to share one size")
            items[b] = L.AugItem(im.data_ptr(), h, w, int(p[0]), int(p[1]), int(p[2]), int(p[3]), int(p[4]), int(p[5]))
        need = int(lib.vg_augment_ws_bytes(items, n, oh, ow))
        if need == 0:
            raise RuntimeError(f"libvg_hip augment: {L.last_error()}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev[0].device)
        out = torch.empty(n, 3, oh, ow, dtype=torch.float32, device=dev[0].device)
        L.check(lib.vg_augment_batch(items, n, oh, ow, self._ws.data_ptr(), self._ws.numel(), out.data_ptr(), L.stream()),
                "augment_batch")
        self._keep = dev            # the sources must outlive the enqueued kernels
        return out


# ---- dataset / loader (decode only on the host) ---------------------------------------------------------------------
class ImageDataset(Dataset):
    """data.py:23-96 with the transform moved to the device: __getitem__ returns the decoded uint8 HWC image."""

    def __init__(self, root_dir, dataset_name, image_size=256, split='train', transform=None):
        self.root_dir, self.dataset_name, self.image_size, self.split = root_dir, dataset_name, image_size, split
        self.transform = transform          # optional HOST transform on the uint8 tensor (None in get_dataloader)
        self.file_paths = self._get_file_paths()

    def _get_file_paths(self):
        if self.dataset_name == 'kodak':
            return sorted(glob.glob(os.path.join(self.root_dir, 'kodak', '*.png')))
        if self.dataset_name == 'clic':
            sub = {'train': 'train', 'val': 'val'}.get(self.split, 'test')
            return sorted(glob.glob(os.path.join(self.root_dir, 'clic', sub, '*.png')))
        if self.dataset_name == 'div2k':
            sub = {'train': 'DIV2K_train_HR', 'val': 'DIV2K_valid_HR'}.get(self.split, 'DIV2K_test_HR')
            return sorted(glob.glob(os.path.join(self.root_dir, 'div2k', sub, '*.png')))
        raise ValueError(f"Dataset {self.dataset_name} not supported")

    def __len__(self):
        return len(self.file_paths)

    def __getitem__(self, idx):
        from PIL import Image
        image = torch.from_numpy(np.asarray(Image.open(self.file_paths[idx]).convert('RGB')).copy())
        return self.transform(image) if self.transform else image


class DeviceLoader:
    """Iterates a DataLoader of decoded images and yields transformed float32 batches on the device."""

    def __init__(self, loader, transform):
        self.loader, self.transform = loader, transform
        self.dataset = loader.dataset

    def __len__(self):
        return len(self.loader)

    def set_epoch(self, epoch: int) -> None:
        """Under a DistributedSampler the shuffle is a function of (seed, epoch): without this every epoch of a
        multi-rank run would walk the shards in the same order, unlike the single-process loader (shuffle=True draws a
        fresh permutation per epoch, as the reference's does, data.py:133-139)."""
        sampler = getattr(self.loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)

    def __iter__(self):
        for images in self.loader:
            yield self.transform(images)


def _collate_list(batch):
    return list(batch)


def get_dataloader(config, split='train', rank=0, world=1):
    """data.py:99-141.  ``world > 1`` (one process per GPU): every rank reads its own shard of the (shuffled) dataset,
    ``config.batch_size`` images per rank."""
    set_seed(config.seed)
    train = split == 'train'
    dataset = ImageDataset(config.data_dir, config.dataset, config.image_size, split)
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=train, seed=config.seed)
    loader = DataLoader(dataset, batch_size=config.batch_size if train else config.eval_batch_size,
                        shuffle=train and sampler is None, sampler=sampler,
                        num_workers=config.num_workers, pin_memory=True, collate_fn=_collate_list)
    return DeviceLoader(loader, DeviceTransform(config.image_size, train=train))


def get_kodak_dataset(config):
    """data.py:144-166: ToTensor only, one image at a time."""
    dataset = ImageDataset(config.data_dir, 'kodak', None, 'test')
    loader = DataLoader(dataset, batch_size=1, shuffle=False, num_workers=config.num_workers, pin_memory=True,
                        collate_fn=_collate_list)
    return DeviceLoader(loader, DeviceTransform(None, train=False))
