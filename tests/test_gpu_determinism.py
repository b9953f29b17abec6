"""GPU tier: reproducibility of the f32 (parity) mode — ``vg_set_deterministic`` (include/vg_api.h).

Two runs of the same VQ train step (same weights, same batch; discriminator and generator phases, optimizer updates,
BatchNorm running statistics, codebook EMA) must give the same BITS everywhere: losses, code indices, both flat gradient
buffers and every entry of the state_dict.  The fixed-point statistics accumulators are also checked on their own
against the f64-atomic ones."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, as_t  # noqa: E402
from vae_gan_amd.testing import fill_state_dict  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _vq_step(images, cfg):
    from vae_gan_amd.train import build_optimizers, vq_train_step
    from vae_gan_amd.vq_vae_gan.model import VQVAEGAN
    model = VQVAEGAN(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.cuda().train()
    opt_g, opt_d, _, _ = build_optimizers(model, cfg)
    out = {}
    for it in range(2):                    # the second iteration runs on the first one's updated weights and statistics
        dl, gl, idx = vq_train_step(model, images, opt_g, opt_d, None)
        torch.cuda.synchronize()
        out[f"it{it}:d_loss"], out[f"it{it}:g_loss"] = dl["total"].detach().cpu(), gl["total"].detach().cpu()
        out[f"it{it}:idx"] = idx.cpu()
        out[f"it{it}:flat_g"], out[f"it{it}:flat_d"] = opt_g.flat_g.clone().cpu(), opt_d.flat_g.clone().cpu()
    for k, v in model.state_dict().items():
        out["sd:" + k] = v.detach().clone().cpu()
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16_det"])
def test_two_identical_steps_give_identical_bits(gpu, dtype, monkeypatch):
    from vae_gan_amd import ops
    from vae_gan_amd.vq_vae_gan.config import Config
    if dtype == torch.bfloat16:
        monkeypatch.setenv("VG_DETERMINISTIC", "1")
    try:
        ops.set_compute_dtype(dtype)
        assert ops.deterministic()
        g = load_golden("vq_step_64")
        images = as_t(g["images"]).cuda()
        a, b = _vq_step(images, Config()), _vq_step(images, Config())
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        assert not diff, diff[:10]
    finally:
        monkeypatch.delenv("VG_DETERMINISTIC", raising=False)
        ops.set_compute_dtype(torch.bfloat16)
        assert not ops.deterministic()


def _arch_step(arch):
    """Two iterations of the beta (256x256, Linear layers: split-K forward) or hierarchical (128x128: attention, three
    discriminator scales) train step on their golden inputs."""
    from vae_gan_amd.testing import synthetic_images, hier_step_eps
    from vae_gan_amd.train import build_optimizers, beta_train_step, hier_train_step
    if arch == "beta":
        from vae_gan_amd.beta_vae_gans.config import Config
        from vae_gan_amd.beta_vae_gans.model import BetaVAEGAN as Model
        g = load_golden("beta_step_256")
        images = synthetic_images(2, 256, seed=int(g["images_seed"])).cuda()
    else:
        from vae_gan_amd.hierarchical_vae_gan.config import Config
        from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalVAEGAN as Model
        g = load_golden("hier_step_128")
        images = as_t(g["images"]).cuda()
        eps = hier_step_eps(2, 128)
    cfg = Config()
    cfg.recon_target_policy, cfg.hier_decoder_policy = "area", "nearest"
    model = Model(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.cuda().train()
    opt_g, opt_d, _, _ = build_optimizers(model, cfg)
    out = {}
    for it in range(2):
        if arch == "beta":
            dl, gl = beta_train_step(model, images, opt_g, opt_d, None, eps=as_t(g[f"eps{it}"]).cuda())
        else:
            dl, gl = hier_train_step(model, images, opt_g, opt_d, None, eps_lists=[[e.cuda() for e in fwd] for fwd in eps[it]])
        torch.cuda.synchronize()
        out[f"it{it}:d_loss"], out[f"it{it}:g_loss"] = dl["total"].detach().cpu(), gl["total"].detach().cpu()
        out[f"it{it}:flat_g"], out[f"it{it}:flat_d"] = opt_g.flat_g.clone().cpu(), opt_d.flat_g.clone().cpu()
    for k, v in model.state_dict().items():
        out["sd:" + k] = v.detach().clone().cpu()
    return out


@pytest.mark.parametrize("arch", ["beta", "hier"])
def test_beta_and_hier_f32_steps_are_reproducible(gpu, arch):
    from vae_gan_amd import ops
    try:
        ops.set_compute_dtype(torch.float32)
        a, b = _arch_step(arch), _arch_step(arch)
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        assert not diff, diff[:10]
    finally:
        ops.set_compute_dtype(torch.bfloat16)


def test_fixed_point_statistics_match_f64_atomics(gpu):
    """vg_channel_stats + vg_bn_finalize through both kinds of accumulator: same mean / invstd to f32 rounding, over
    magnitudes from 1e-12 to 1e+6 in one tensor (bins of every exponent range), and the fixed-point result twice."""
    from vae_gan_amd import lib as L
    lib = L.load()
    rows, c, cpad = 40000, 48, 64
    gen = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(rows, cpad, device="cuda", generator=gen)
    x *= (10.0 ** torch.linspace(-12, 6, cpad, device="cuda"))[None, :]
    x[:, c:] = 0
    res = {}
    try:
        for det in (0, 1, 1):
            lib.vg_set_deterministic(det)
            stats = torch.zeros(L.STATS_REPLICAS * 2 * cpad, dtype=torch.float64, device="cuda")
            L.check(lib.vg_channel_stats(x.data_ptr(), stats.data_ptr(), rows, cpad, c, cpad, L.VG_F32, L.stream()), "channel_stats")
            vec = torch.zeros(4, c, device="cuda")
            ones, zeros = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
            L.check(lib.vg_bn_finalize(stats.data_ptr(), c, cpad, rows, ones.data_ptr(), zeros.data_ptr(), 0.0, 0.1, None, None, None, 1,
                                       vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), L.stream()), "bn_finalize")
            torch.cuda.synchronize()
            res.setdefault(det, []).append(vec.cpu())
    finally:
        lib.vg_set_deterministic(0)
    assert torch.equal(res[1][0], res[1][1])
    ref = x[:, :c].double()
    mean, var = ref.mean(0).cpu(), ref.var(0, unbiased=False).cpu()
    for v in (res[0][0], res[1][0]):
        assert float(((v[0].double() - mean).abs() / var.sqrt()).max()) < 1e-6
        assert float((v[1].double() * var.sqrt() - 1).abs().max()) < 1e-5
