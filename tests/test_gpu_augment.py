"""GPU tier: the HIP input transform (csrc/augment.hip via vg_augment_batch and vae_gan_amd/data.py) — bit-exact
against vectors made by Pillow itself (tests/golden/augment.npz) and against the oracle on random batches of
differently-sized images; size-independent properties at dataset sizes."""
import numpy as np
import pytest
import torch

from oracle import augment_oracle as A
from test_augment_oracle import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import vae_gan_amd.lib as L
    L.load()
    from vae_gan_amd import data as D
    return D


def test_pillow_vectors(data):
    by_size = {}
    for n, img, S, params, want in cases():
        by_size.setdefault(S, []).append((n, img, params, want))
    for S, group in by_size.items():       # one launch per output size, images of different sizes in one batch
        tf = data.DeviceTransform(S, train=True)
        out = tf([torch.from_numpy(img).cuda() for _, img, _, _ in group], params=[p for _, _, p, _ in group]).cpu()
        for b, (n, _, _, want) in enumerate(group):
            assert torch.equal(out[b], A.to_tensor(want)), f"case {n}"


@pytest.mark.parametrize("S", [64, 256])
def test_random_batches_against_oracle(data, S):
    rng = np.random.default_rng(S)
    g = torch.Generator().manual_seed(S)
    imgs, params = [], []
    for b in range(6):
        h, w = int(rng.integers(S // 2, 3 * S)), int(rng.integers(S // 2, 3 * S))
        imgs.append(rng.integers(0, 256, (h, w, 3)).astype(np.uint8))
        params.append(A.draw_train_params(h, w, generator=g))
    out = data.DeviceTransform(S)([torch.from_numpy(a).cuda() for a in imgs], params=params).cpu()
    for b in range(6):
        assert torch.equal(out[b], A.train_transform(imgs[b], S, params[b])), (b, imgs[b].shape, params[b])
    ev = data.DeviceTransform(S, train=False)([torch.from_numpy(a).cuda() for a in imgs]).cpu()
    for b in range(6):
        assert torch.equal(ev[b], A.eval_transform(imgs[b], S)), b


def test_dataset_sizes_properties(data):
    """Kodak-sized (512x768) and DIV2K-sized (1356x2040) images to 256x256: flips commute with the transform of a
    full-image box, the 1:1 box is ToTensor exactly, and the output is a convex combination of the inputs."""
    rng = np.random.default_rng(0)
    for h, w in ((512, 768), (1356, 2040)):
        a = torch.from_numpy(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)).cuda()
        tf = data.DeviceTransform(256)
        full = (0, 0, h, w)
        o = tf([a, a, a, a], params=[(False, False) + full, (True, False) + full, (False, True) + full, (True, True) + full])
        assert torch.equal(o[1], o[0].flip(2)) and torch.equal(o[2], o[0].flip(1)) and torch.equal(o[3], o[0].flip(1, 2))
        assert float(o.min()) >= 0.0 and float(o.max()) <= 1.0
        # smooth downscale: the mean is preserved to rounding (normalised coefficients)
        assert abs(float(o[0].mean()) - float(a.float().mean()) / 255) < 2e-3
        ident = data.DeviceTransform(None, train=False)([a])
        assert torch.equal(ident[0].cpu(), a.cpu().permute(2, 0, 1).float().div(255))      # ToTensor divides on the CPU
        # a box in the flipped image = the mirrored box in the original
        i, j, bh, bw = 17, 40, h - 100, w - 90
        x = tf([a], params=[(True, True, i, j, bh, bw)])[0]
        sub = a[h - i - bh:h - i, w - j - bw:w - j].contiguous()
        y = tf([sub], params=[(True, True, 0, 0, bh, bw)])[0]
        assert torch.equal(x, y)


def test_bad_boxes_are_rejected_on_the_host(data):
    a = torch.zeros(32, 40, 3, dtype=torch.uint8).cuda()
    with pytest.raises(RuntimeError, match="outside"):
        data.DeviceTransform(16)([a], params=[(False, False, 0, 0, 33, 40)])
    with pytest.raises(RuntimeError, match="outside"):
        data.DeviceTransform(16)([a], params=[(False, False, 5, 30, 20, 20)])
    with pytest.raises(ValueError):
        data.DeviceTransform(16)([a.float()])


def test_get_dataloader_on_png_files(data, tmp_path):
    """get_dataloader / get_kodak_dataset (data.py:99-166) over PNG files written here: eval split = Resize + ToTensor
    of every file, bit-exact against the oracle; train split yields [B,3,S,S] batches in [0,1]."""
    from PIL import Image
    rng = np.random.default_rng(5)
    d = tmp_path / "kodak"
    d.mkdir()
    raw = []
    for n in range(5):
        a = rng.integers(0, 256, (48 + 8 * n, 64, 3)).astype(np.uint8)
        Image.fromarray(a).save(d / f"kodim{n:02d}.png")
        raw.append(a)

    class Cfg:
        seed, data_dir, dataset, image_size, batch_size, eval_batch_size, num_workers = 42, str(tmp_path), "kodak", 32, 2, 5, 0

    ev = list(data.get_dataloader(Cfg, "val"))
    assert len(ev) == 1 and ev[0].shape == (5, 3, 32, 32)
    for n in range(5):
        assert torch.equal(ev[0][n].cpu(), A.eval_transform(raw[n], 32))
    tr = list(data.get_dataloader(Cfg, "train"))
    assert [tuple(b.shape) for b in tr] == [(2, 3, 32, 32), (2, 3, 32, 32), (1, 3, 32, 32)]
    assert all(float(b.min()) >= 0 and float(b.max()) <= 1 for b in tr)
    kd = list(data.get_kodak_dataset(Cfg))
    assert len(kd) == 5 and torch.equal(kd[2][0].cpu(), A.to_tensor(raw[2]))
