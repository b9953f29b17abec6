"""GPU tier: the HIP VQ-VAE-GAN modules against the golden vectors (reference in float64 +
the reference's own fp32 noise floor) and against the oracle.  fp32 mode: rel-L2 <= max(1e-4,
4*noise); bf16 mode: 2e-2 on forward values / losses, looser documented bounds on deep gradients."""
import numpy as np
import pytest
import torch

from conftest import load_golden, as_t, close_l2, check_digests, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import vae_gan_amd.lib as L
    L.load()
    from vae_gan_amd import ops
    return ops


def _fill(module):
    """Same key-hash fill as the golden generator (keys relative to the module itself)."""
    from vae_gan_amd.testing import fill_state_dict
    module.load_state_dict(fill_state_dict(module.state_dict()))
    return module


def _run(module, g, n_out=1):
    x = as_t(g["x"]).cuda().requires_grad_(True)
    y = module(x)
    ys = list(y) if isinstance(y, (tuple, list)) else [y]
    loss = sum((t * as_t(g[f"cot{i}"]).cuda()).sum() for i, t in enumerate(ys))
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad for k, p in module.named_parameters()}
    return ys, x.grad, grads


@pytest.mark.parametrize("tag,K,D", [("a", 512, 64), ("b", 512, 64), ("c", 8192, 256)])
def test_vector_quantizer_golden(vg, tag, K, D):
    from vae_gan_amd.vq_vae_gan.model import VectorQuantizer
    from vae_gan_amd.testing import fill_state_dict
    from oracle import vaegan_oracle as O
    g = load_golden("vq_quantizer_" + tag)
    vq = VectorQuantizer(K, D, 0.25, 0.99)
    sd = fill_state_dict(vq.state_dict())
    if tag == "b":
        sd["embeddings.weight"] = as_t(g["codebook_before"]).clone()
        sd["ema_w"] = as_t(g["ema_w_before"]).clone()
        sd["ema_cluster_size"] = as_t(g["ema_cluster_before"]).clone()
    vq.load_state_dict(sd)
    vq.cuda()
    z = as_t(g["z"]).cuda().requires_grad_(True)
    zq, loss, idx = vq(z, training=True)
    ((zq * as_t(g["cot"]).cuda()).sum() + loss).backward()
    torch.cuda.synchronize()
    ref_idx = as_t(g["idx"])
    mism = (idx.cpu() != ref_idx).flatten().nonzero().flatten()
    # documented near-tie rule: a mismatch is admissible iff the reference distances of the
    # two candidates differ by <= 4 ulp(fp32) of the distance
    if len(mism):
        flat = as_t(g["z"]).permute(0, 2, 3, 1).reshape(-1, D)
        dist = O.vq_distances(flat, sd["embeddings.weight"])
        for i in mism.tolist():
            a, b = dist[i, idx.flatten()[i].item()], dist[i, ref_idx.flatten()[i]]
            ulp = np.spacing(np.float32(abs(b.item())))
            assert abs(a.item() - b.item()) <= 4 * ulp, f"row {i}: not a near-tie ({a.item()} vs {b.item()})"
    assert len(mism) <= max(1, idx.numel() // 500)
    if len(mism) == 0:
        close_l2(zq, g["zq"], what="zq", rtol=1e-6)
        close_l2(loss, g["vq_loss"], what="vq_loss", rtol=1e-5)
        close_l2(z.grad, g["dz"], what="dz", rtol=1e-5)
        close_l2(vq.ema_cluster_size, g["ema_cluster_size"], what="ema_cluster", rtol=1e-6)
        if "ema_w" in g:
            close_l2(vq.ema_w, g["ema_w"], what="ema_w", rtol=1e-5)
            ca, cb = vq.embeddings.weight.detach().cpu().double(), as_t(g["codebook_after"]).double()
            assert ((ca - cb).abs() <= 1e-4 * cb.abs() + 1e-6).all()   # incl. the x1e5 dead-code rows


def test_vector_quantizer_random_vs_oracle(vg):
    """Larger seeded case (B=4, 32x32 latent) against the oracle; indices bit-exact up to near-ties."""
    from vae_gan_amd.vq_vae_gan.model import VectorQuantizer
    from oracle import vaegan_oracle as O
    torch.manual_seed(3)
    K, D = 512, 64
    vq = VectorQuantizer(K, D, 0.25).cuda()
    vq.embeddings.weight.data.normal_(0, 0.5)
    vq.ema_w.copy_(vq.embeddings.weight.data)
    z = torch.randn(4, D, 32, 32) * 0.5
    sd = {"vq.embeddings.weight": vq.embeddings.weight.detach().cpu().clone(), "vq.ema_w": vq.ema_w.cpu().clone(),
          "vq.ema_cluster_size": vq.ema_cluster_size.cpu().clone()}
    zq_o, loss_o, idx_o = O.vector_quantizer(sd, z, 0.25, 0.99, True, p="vq")
    zq, loss, idx = vq(z.cuda(), training=True)
    torch.cuda.synchronize()
    frac = (idx.cpu() != idx_o).float().mean().item()
    assert frac <= 2e-3, f"index mismatch fraction {frac}"
    if frac == 0:
        close_l2(vq.embeddings.weight, sd["vq.embeddings.weight"], rtol=1e-5, what="codebook")
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * abs(loss_o.item())


CASES = {
    "resblock64": lambda: __import__("vae_gan_amd.nn", fromlist=["x"]).ResidualBlock(64, 64),
    "vq_encoder": lambda: __import__("vae_gan_amd.vq_vae_gan.model", fromlist=["x"]).Encoder(3, [64, 128, 256, 512], 64),
    "vq_decoder": lambda: __import__("vae_gan_amd.vq_vae_gan.model", fromlist=["x"]).Decoder(64, [512, 256, 128, 64], 3),
    "vq_discriminator": lambda: __import__("vae_gan_amd.vq_vae_gan.model", fromlist=["x"]).Discriminator(3, [64, 128, 256, 512]),
}


class _NHWCWrap(torch.nn.Module):
    """ResidualBlock works on internal NHWC tensors: wrap it with the boundary conversions."""

    def __init__(self, inner, c):
        super().__init__()
        self.inner, self.c = inner, c

    def forward(self, x):
        from vae_gan_amd import ops
        return ops.to_nchw(self.inner(ops.to_nhwc(x)), self.c)


@pytest.mark.parametrize("name", list(CASES))
def test_module_f32(vg, name):
    vg.set_compute_dtype(torch.float32)
    g = load_golden(name)
    inner = _fill(CASES[name]())
    m = (_NHWCWrap(inner, 64) if name == "resblock64" else inner).cuda().train()
    ys, dx, _ = _run(m, g)
    grads = {k: p.grad for k, p in inner.named_parameters()}
    for i, t in enumerate(ys):
        close_l2(t, g[f"y{i}"], g[f"noise:y{i}"], what=f"{name}.y{i}")
    close_l2(dx, g["dx"], g["noise:dx"], what=name + ".dx")
    check_digests(g, grads)
    for key, ref in g.items():
        if key.startswith("buf:"):
            t = dict(inner.named_buffers())[key[4:]]
            if t.dtype == torch.int64:
                assert int(t) == int(ref), key
            else:
                close_l2(t, ref, 0.0, rtol=1e-5, what=key)


@pytest.mark.parametrize("name", list(CASES))
def test_module_bf16(vg, name):
    """bf16 storage + bf16 MFMA: forward of the whole 13-27 layer stacks within 2e-2 (north_star bf16 tolerance).
    Gradients in bf16 mode are held layer by layer in tests/test_gpu_layer_grads.py (1e-2 against the reference with
    bf16 storage, every parameter gradient); through a whole stack the LeakyReLU masks of pre-activations within bf16
    rounding of zero flip in every layer (the bf16-storage reference itself is 3e-2..5e-2 from float64 after ONE
    layer, tests/golden/layer_units.npz 'noise_bf16:'), so the stack-level bound on dx is only a sanity check."""
    vg.set_compute_dtype(torch.bfloat16)
    g = load_golden(name)
    inner = _fill(CASES[name]())
    m = (_NHWCWrap(inner, 64) if name == "resblock64" else inner).cuda().train()
    ys, dx, _ = _run(m, g)
    for i, t in enumerate(ys):
        close_l2(t, g[f"y{i}"], 0.0, rtol=2e-2, what=f"{name}.y{i}")
    e = rel_l2(dx.cpu(), g["dx"])
    assert e <= (6e-2 if name == "resblock64" else 0.25), f"{name}.dx rel-L2 {e}"
    vg.set_compute_dtype(torch.bfloat16)
