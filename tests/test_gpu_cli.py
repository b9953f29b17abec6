"""GPU tier: `python -m vae_gan_amd.<arch>.main` end to end on a synthetic PNG directory: two iterations of the VQ
and hierarchical steps through the flag surface, true BPP from the HIP arithmetic coder logged like
vq_vae_gan/main.py:115-118, checkpoints in the reference's layout."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _png_dir(tmp_path, n=2, h=64, w=96):
    from PIL import Image
    d = tmp_path / "data" / "kodak"
    d.mkdir(parents=True)
    rng = np.random.default_rng(7)
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(d / f"kodim{i:02d}.png")
    return str(tmp_path / "data")


def test_vq_main_two_iterations_logs_true_bpp(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import ops
    from vae_gan_amd.vq_vae_gan import main as M
    out = str(tmp_path / "out")
    try:
        model = M.main(["--data_dir", _png_dir(tmp_path), "--output_dir", out, "--epochs", "2", "--batch_size", "2",
                        "--image_size", "64", "--num_workers", "0", "--dtype", "f32", "--num_embeddings", "64"])
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    rows = json.load(open(os.path.join(out, "train_metrics.json")))
    assert len(rows) == 2 and all(np.isfinite(r["total"]) for r in rows)
    # iteration 0 of every epoch: bytes out of the arithmetic coder for one 64x64 image with 8x8 indices
    bpp = rows[0]["bpp"]
    assert 0 < bpp <= (64 * 6 / 8 + 8) * 8 / (64 * 64) + 1e-9, bpp
    assert bpp * 64 * 64 / 8 == round(bpp * 64 * 64 / 8)            # a whole number of bytes
    ev = json.load(open(os.path.join(out, "eval_metrics.json")))
    assert len(ev) == 1 and ev[0]["bpp"] > 0 and np.isfinite(ev[0]["psnr"])     # save_interval 5: only the last epoch evaluates
    ck = torch.load(os.path.join(out, "checkpoints", "latest.pth"), map_location="cpu", weights_only=True)
    assert ck["epoch"] == 1 and set(ck["model_state_dict"]) == set(model.state_dict())
    assert model.vector_quantizer.embeddings.weight.shape == (64, 64)
    # --resume restores the model only (vq_vae_gan/main.py:348-355)
    try:
        M.main(["--data_dir", str(tmp_path / "data"), "--output_dir", out, "--epochs", "1", "--batch_size", "2",
                "--image_size", "64", "--num_workers", "0", "--dtype", "f32", "--num_embeddings", "64", "--resume",
                "--max_iters", "1"])
    finally:
        ops.set_compute_dtype(torch.bfloat16)


def test_hier_main_policies(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import ops
    from vae_gan_amd.hierarchical_vae_gan import main as M
    data = _png_dir(tmp_path, h=128, w=128)
    common = ["--data_dir", data, "--epochs", "1", "--batch_size", "2", "--image_size", "128", "--num_workers", "0",
              "--max_iters", "2", "--dtype", "f32"]
    try:
        with pytest.raises(RuntimeError, match="channels"):                # the reference's decoder failure, by default
            M.main(common + ["--output_dir", str(tmp_path / "o1")])
        M.main(common + ["--output_dir", str(tmp_path / "o2"), "--hier_decoder_policy", "nearest",
                         "--recon_target_policy", "area"])
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    rows = json.load(open(os.path.join(str(tmp_path / "o2"), "train_metrics.json")))
    assert len(rows) == 1 and np.isfinite(rows[0]["total"]) and rows[0]["bpp"] == 0.0
