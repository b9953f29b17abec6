"""GPU tier: the full VQ-VAE-GAN train step (vq_vae_gan/main.py:75-111 semantics) on the HIP engine
against the golden captured from the reference model + torch.optim.Adam (two iterations, B=2, 64x64)."""
import pytest
import torch

from conftest import load_golden, as_t, close_l2, check_state
from vae_gan_amd.testing import fill_state_dict, digest

pytestmark = pytest.mark.gpu


def _setup(dtype):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import ops
    from vae_gan_amd.vq_vae_gan.model import VQVAEGAN
    from vae_gan_amd.vq_vae_gan.config import Config
    from vae_gan_amd.train import build_optimizers
    ops.set_compute_dtype(dtype)
    cfg = Config()
    model = VQVAEGAN(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.cuda().train()
    opt_g, opt_d, _, _ = build_optimizers(model, cfg)
    return model, opt_g, opt_d


def test_vq_step_f32_matches_reference():
    from vae_gan_amd.train import vq_train_step
    g = load_golden("vq_step_64")
    model, opt_g, opt_d = _setup(torch.float32)
    images = as_t(g["images"]).cuda()
    for it in range(2):
        dl, gl, idx2 = vq_train_step(model, images, opt_g, opt_d, elide_d_wgrad=(it == 1))
        torch.cuda.synchronize()
        mine = {"d_loss": dl["total"], "d_recon": dl["recon"], "g_loss": gl["total"], "recon": gl["recon"],
                "vq": gl["vq"], "adv": gl["adv"]}
        for k, v in mine.items():
            ref = float(g[f"it{it}_{k}"])
            tol = max(1e-4, 4 * float(g[f"noise:it{it}_{k}"]))
            assert abs(float(v.detach()) - ref) <= tol * max(abs(ref), 1e-3), (it, k, float(v.detach()), ref, tol)
        match = (idx2.cpu() == as_t(g[f"it{it}_idx2"])).float().mean().item()
        assert match >= (1.0 if it == 0 else float(g["idxeq64:it1_idx2"]) - 0.02), (it, match)
    # BatchNorm counters exactly (generator +2 / iteration, discriminator +3), every other entry by norm AND probe dot
    check_state(g, model.state_dict(), float(g["noise:it1_recon_img"]))
    from vae_gan_amd import ops
    ops.set_compute_dtype(torch.bfloat16)


def test_vq_step_bf16_close_to_reference():
    """bf16 mode: first-iteration losses within 2e-2 of the fp32 reference (north_star bf16 tolerance), except where the
    reference's own bf16-storage runs are further from it than that (bf16_bound: the discriminator's hinge loss, 4.2e-2
    here = 1.5 x the largest of the reference's eight samples, 2.8 %; the engine measures 1.9-2.5 %).  The tight bf16
    checks are the per-layer ones (tests/test_gpu_layer_grads.py), held against a bf16-storage run of the reference."""
    from vae_gan_amd.train import vq_train_step
    g = load_golden("vq_step_64")
    model, opt_g, opt_d = _setup(torch.bfloat16)
    images = as_t(g["images"]).cuda()
    dl, gl, idx2 = vq_train_step(model, images, opt_g, opt_d)
    torch.cuda.synchronize()
    for k, v in {"d_loss": dl["total"], "g_loss": gl["total"], "recon": gl["recon"], "vq": gl["vq"]}.items():
        ref = float(g[f"it0_{k}"])
        assert abs(float(v.detach()) - ref) <= bf16_bound(g, k) * max(abs(ref), 1e-3), (k, float(v.detach()), ref)
    assert all(torch.isfinite(p).all() for p in model.parameters())


def bf16_bound(g, k):
    """north_star's 2e-2, or 1.5 x the largest distance of the REFERENCE's own bf16-storage runs from its exact run where
    that is larger.  'noise_bf16:it0_*' in the goldens (make_golden.py:_bf16_floor) is the largest of EIGHT runs of the
    reference modules with bf16 storage, which differ only by a 2^-23 relative jitter before each rounding (= another f32
    accumulation order): the discriminator's hinge loss sits behind a stack of LeakyReLU masks (hierarchical: and a nearly
    one-hot softmax) and moves in steps with the rounding points - the reference's own eight samples span 0.3 % .. 2.8 %
    (vq_step_64) and 0.2 % .. 3.7 % (hier_step_128), every other scalar stays below 0.7 %.  DESIGN.md section 2 has the
    table of bounds, floors and measured values (tests/gpu_bf16_margins.py prints it)."""
    return max(2e-2, 1.5 * float(g[f"noise_bf16:it0_{k}"]))


def _profiled_kernels(fn):
    """Run fn() with a KernelProfile installed; the set of kernel variants the library dispatched."""
    from vae_gan_amd import ops
    prof = ops.KernelProfile()
    ops.PROFILE[0] = prof
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        ops.PROFILE[0] = None
    return out, {name for name, _, _, _ in prof.records}


def test_vq_step_256_f32_matches_reference():
    """one iteration of the reference step (vq_vae_gan/main.py:75-111) at the reference's own
    image_size = 256 (B = 2), where the layer shapes are those of the benchmarked configuration."""
    from vae_gan_amd.train import vq_train_step
    from vae_gan_amd import ops
    g = load_golden("vq_step_256")
    model, opt_g, opt_d = _setup(torch.float32)
    images = as_t(g["images"]).cuda()
    dl, gl, idx2 = vq_train_step(model, images, opt_g, opt_d)
    torch.cuda.synchronize()
    mine = {"d_loss": dl["total"], "d_recon": dl["recon"], "g_loss": gl["total"], "recon": gl["recon"], "vq": gl["vq"], "adv": gl["adv"]}
    for k, v in mine.items():
        ref = float(g[f"it0_{k}"])
        tol = max(1e-4, 4 * float(g[f"noise:it0_{k}"]))
        scale = max(abs(ref), 1e-3, abs(float(g["it0_d_loss"])) if k == "adv" else 0.0)     # adv: a mean of signed logits
        assert abs(float(v.detach()) - ref) <= tol * scale, (k, float(v.detach()), ref, tol)
    assert torch.equal(idx2.cpu(), as_t(g["it0_idx2"]))                    # 2 x 32 x 32 code indices, bit-exact
    check_state(g, model.state_dict(), float(g["noise:state"]))      # the reference's own fp32-vs-fp64 distance in the final state
    ops.set_compute_dtype(torch.bfloat16)


def test_vq_step_256_bf16_runs_the_benchmarked_kernels():
    """bf16 mode at 256 x 256: first-iteration losses within the bf16 bounds of the fp32 reference, and the step went
    through the kernels bench.py measures: the test names the kernel variants it covered (vg_last_conv_kernel /
    vg_last_wgrad_kernel per launch), so a change of the selection rules cannot silently move it off them."""
    from vae_gan_amd.train import vq_train_step
    g = load_golden("vq_step_256")
    model, opt_g, opt_d = _setup(torch.bfloat16)
    images = as_t(g["images"]).cuda()
    (dl, gl, idx2), kernels = _profiled_kernels(lambda: vq_train_step(model, images, opt_g, opt_d))
    for k, v in {"d_loss": dl["total"], "g_loss": gl["total"], "recon": gl["recon"], "vq": gl["vq"]}.items():
        ref = float(g[f"it0_{k}"])
        assert abs(float(v.detach()) - ref) <= bf16_bound(g, k) * max(abs(ref), 1e-3), (k, float(v.detach()), ref)
    match = (idx2.cpu() == as_t(g["it0_idx2"])).float().mean().item()
    assert match >= float(g["idxeq_bf16:it0_idx2"]) - 0.05, match           # the reference's own bf16-storage run keeps this share
    for want in ("igemm3<bf16,4,2,4,2,3x3>", "igemm3<bf16,4,2,4,2,2x2>", "igemm3s2<bf16,", "igemm4<bf16,3x3", "igemm4c<bf16,3x3",
                 "igemm4p<bf16,2x2", "thin_conv", "conv1", "wgrad3<4,2,1>", "wgrad3<2,2,2>", "wgrad3s2<"):
        assert any(n.startswith(want) for n in kernels), (want, sorted(kernels))
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_shared_encoder_equals_literal_schedule():
    """The default step runs the encoder once per iteration and books both BatchNorm buffer updates
    (train.vq_train_step); it must land where the literal two-forward schedule of vq_vae_gan/main.py:78,99 lands.
    Compared after ONE iteration: from the second on, Adam's lr*sign(g) on rounding-noise gradients makes even
    the reference's own fp32 run irreproducible below 2e-2 (noise:it1_d_loss in the golden)."""
    from vae_gan_amd.train import vq_train_step
    from vae_gan_amd import ops
    g = load_golden("vq_step_64")
    images = as_t(g["images"]).cuda()
    runs = []
    for share in (True, False):
        model, opt_g, opt_d = _setup(torch.float32)
        dl, gl, idx2 = vq_train_step(model, images, opt_g, opt_d, share_encoder=share)
        out = (float(dl["total"].detach()), float(gl["total"].detach()), float(gl["vq"]), float(gl["recon"]), idx2.cpu())
        runs.append((out, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    (a, sda), (b, sdb) = runs
    for x, y in zip(a[:4], b[:4]):
        assert abs(x - y) <= 1e-5 * abs(y), (a[:4], b[:4])
    assert torch.equal(a[4], b[4])
    for k in sda:
        if sda[k].dtype == torch.int64:
            assert int(sda[k]) == int(sdb[k]), k
        elif "running_" in k or k.startswith("vector_quantizer"):
            close_l2(sda[k], sdb[k], rtol=1e-5, what=k)
        else:
            close_l2(sda[k], sdb[k], rtol=2e-3, what=k)
    ops.set_compute_dtype(torch.bfloat16)
