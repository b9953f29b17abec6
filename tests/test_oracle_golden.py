"""CPU tier 1: the oracle (oracle/vaegan_oracle.py) against golden vectors captured from the
reference modules by tests/golden/make_golden.py.  fp32, rtol 1e-4;
VQ indices bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden, as_t, close_l2, check_step, check_state, check_digests as _check_digests
from oracle import vaegan_oracle as O
from vae_gan_amd.testing import fill_state_dict, cotangent, digest, hier_step_eps

RTOL = 1e-4


def close(a, b, rtol=RTOL, atol=1e-6, what=""):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item()
    assert err <= rtol * scale + atol, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def leaves(sd):
    return {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and
                not k.endswith(("running_mean", "running_var")) else v.clone()) for k, v in sd.items()}


def check_digests(g, sd, prefix_map=lambda k: k):
    return _check_digests(g, {k: v.grad for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad},
                          name_map=prefix_map)


def check_buffers(g, sd, prefix=""):
    for key, ref in g.items():
        if key.startswith("buf:"):
            close(sd[prefix + key[4:]].detach(), ref, what=key)


# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,K,D", [("a", 512, 64), ("b", 512, 64), ("c", 8192, 256)])
def test_vector_quantizer(tag, K, D):
    g = load_golden("vq_quantizer_" + tag)
    sd = fill_state_dict({"vector_quantizer.ema_cluster_size": torch.zeros(K),
                          "vector_quantizer.ema_w": torch.zeros(K, D),
                          "vector_quantizer.embeddings.weight": torch.zeros(K, D)})
    sd = {k.replace("vector_quantizer.", "vq."): v for k, v in sd.items()}
    if tag == "b":
        sd["vq.embeddings.weight"] = as_t(g["codebook_before"]).clone()
        sd["vq.ema_w"] = as_t(g["ema_w_before"]).clone()
        sd["vq.ema_cluster_size"] = as_t(g["ema_cluster_before"]).clone()
    else:
        # fill_state_dict is keyed by name: regenerate with the reference's (prefix-free) names
        sd = {"vq." + k: v for k, v in fill_state_dict({"ema_cluster_size": torch.zeros(K), "ema_w": torch.zeros(K, D),
                                                        "embeddings.weight": torch.zeros(K, D)}).items()}
    z = as_t(g["z"]).clone().requires_grad_(True)
    zq, loss, idx = O.vector_quantizer(sd, z, 0.25, 0.99, True, p="vq")
    assert torch.equal(idx, as_t(g["idx"])), "VQ indices must be bit-exact on CPU"
    ((zq * as_t(g["cot"])).sum() + loss).backward()
    close(zq.detach(), g["zq"], what="zq")
    close(loss.detach(), g["vq_loss"], what="vq_loss")
    close(z.grad, g["dz"], what="dz")
    close(sd["vq.ema_cluster_size"], g["ema_cluster_size"], what="ema_cluster")
    if "ema_w" in g:
        close(sd["vq.ema_w"], g["ema_w"], what="ema_w")
        # dead codes are scaled by ~1e5: compare relative to each row
        ca, cb = sd["vq.embeddings.weight"].detach().double(), as_t(g["codebook_after"]).double()
        assert ((ca - cb).abs() <= 1e-4 * cb.abs() + 1e-6).all()
    else:
        for nm, t in (("ema_w", sd["vq.ema_w"]), ("codebook_after", sd["vq.embeddings.weight"])):
            d = digest(nm, t)
            assert abs(d[0] - g["gd:" + nm][0]) <= 1e-4 * g["gd:" + nm][0]


def test_residual_block():
    tmpl = {}
    O._t_res(tmpl, "rb", 64)
    _run_module("resblock64", {k[3:]: v for k, v in tmpl.items()},
                lambda sd, x: O.residual_block({"rb." + k: v for k, v in sd.items()}, "rb", x))


def _run_module(gname, tmpl, fn, strip=""):
    """tmpl keys are the reference's own (module-relative) names → same fill as the golden."""
    g = load_golden(gname)
    sd = leaves(fill_state_dict(tmpl))
    x = as_t(g["x"]).clone().requires_grad_(True)
    ys = fn(sd, x)
    ys = list(ys) if isinstance(ys, (tuple, list)) else [ys]
    flat = []
    for t in ys:
        flat.extend(t if isinstance(t, (tuple, list)) else [t])
    sum((t * as_t(g[f"cot{i}"])).sum() for i, t in enumerate(flat)).backward()
    for i, t in enumerate(flat):
        close_l2(t, g[f"y{i}"], g[f"noise:y{i}"], what=f"{gname}.y{i}")
    close_l2(x.grad, g["dx"], g["noise:dx"], what=gname + ".dx")
    check_digests(g, sd, lambda k: k[len(strip):] if strip and k.startswith(strip) else k)
    check_buffers(g, {strip + k: v for k, v in sd.items()})
    return g, sd


def test_vq_encoder():
    _run_module("vq_encoder", O.vq_encoder_template(p="encoder"), lambda sd, x: O.vq_encoder(sd, x, p="encoder"))


def test_vq_decoder():
    _run_module("vq_decoder", O.vq_decoder_template(p="decoder"), lambda sd, x: O.vq_decoder(sd, x, p="decoder"))


def test_vq_discriminator():
    _run_module("vq_discriminator", O.patch_disc_template(p="model"),
                lambda sd, x: O.patch_discriminator(sd, x, p="model"))


def test_vq_full_step():
    g = load_golden("vq_step_64")
    sd = fill_state_dict(O.vq_template())
    images = as_t(g["images"])
    opt = {"g": {}, "d": {}}
    for it in range(2):
        r = O.vq_train_step(sd, images, O.VQ_DEFAULT_CFG, opt)
        for k in ("d_loss", "d_recon", "g_loss", "recon", "vq", "adv"):
            ref = float(g[f"it{it}_{k}"])
            tol = max(1e-4, 4 * float(g[f"noise:it{it}_{k}"]))   # reference's own fp32 noise floor
            assert abs(r[k] - ref) <= tol * max(abs(ref), 1e-3), (it, k, r[k], ref, tol)
        for nm in ("idx1", "idx2"):
            match = (r[nm] == as_t(g[f"it{it}_{nm}"])).float().mean().item()
            # iteration 0 is bit-exact; later iterations inherit Adam sign(g) noise (see make_golden)
            assert match >= min(1.0, float(g[f"idxeq64:it{it}_{nm}"])) - (0.0 if it == 0 else 0.02), (it, nm, match)
        close_l2(r["recon_img"], g[f"it{it}_recon_img"], g[f"noise:it{it}_recon_img"], what="recon")
    last_noise = max(float(g["noise:it1_recon_img"]), 1e-4)
    for key, ref in g.items():
        if key.startswith("sd:"):
            k = key[3:]
            if sd[k].dtype == torch.int64:
                assert int(sd[k]) == int(ref), k    # generator BN +2/iter, discriminator +3/iter
            elif k.endswith("ema_cluster_size"):
                close_l2(sd[k], ref, last_noise, what=k)
        elif key.startswith("sdd:"):
            k = key[4:]
            d = digest(k, sd[k])
            assert abs(d[0] - ref[0]) <= max(1e-4, 4 * last_noise) * max(ref[0], 1e-6), k


def test_vq_full_step_256():
    """one iteration at the reference's own image_size = 256 (B = 2): the oracle against the
    reference-made golden, so that the GPU test of the same golden (tests/test_gpu_step.py) has a pinned checker."""
    g = load_golden("vq_step_256")
    sd = fill_state_dict(O.vq_template())
    r = O.vq_train_step(sd, as_t(g["images"]), O.VQ_DEFAULT_CFG, {"g": {}, "d": {}})
    for k in ("d_loss", "d_recon", "g_loss", "recon", "vq", "adv"):
        ref = float(g[f"it0_{k}"])
        tol = max(1e-4, 4 * float(g[f"noise:it0_{k}"]))
        scale = max(abs(ref), 1e-3, abs(float(g["it0_d_loss"])) if k == "adv" else 0.0)
        assert abs(r[k] - ref) <= tol * scale, (k, r[k], ref, tol)
    for nm in ("idx1", "idx2"):
        assert torch.equal(r[nm], as_t(g[f"it0_{nm}"])), nm
    close_l2(r["recon_img"], g["it0_recon_img"], g["noise:it0_recon_img"], what="recon")
    for key, ref in g.items():
        if key.startswith("sd:") and sd[key[3:]].dtype == torch.int64:
            assert int(sd[key[3:]]) == int(ref), key


def test_losses():
    g = load_golden("losses")
    mu, lv, eps = as_t(g["mu"]), as_t(g["logvar"]), as_t(g["eps"])
    close(O.reparameterize(mu, lv, eps), g["z"], what="z")
    close(O.kl_loss(mu, lv), g["kl"], what="kl")
    close(O.hinge_d_loss(as_t(g["d_real"]), as_t(g["d_fake"])), g["hinge"], what="hinge")
    close(-as_t(g["d_fake"]).mean(), g["adv"], what="adv")


# ---- β modules (256×256 input is forced by the Linear layers; slower) ----------------------
def test_beta_encoder():
    t = {k[len("encoder."):]: v for k, v in O.beta_template().items() if k.startswith("encoder.")}
    _run_module("beta_encoder", t, lambda sd, x: O.beta_encoder({"encoder." + k: v for k, v in sd.items()}, x))


def test_beta_decoder():
    t = {k[len("decoder."):]: v for k, v in O.beta_template().items() if k.startswith("decoder.")}
    _run_module("beta_decoder", t, lambda sd, x: O.beta_decoder({"decoder." + k: v for k, v in sd.items()}, x))


def test_beta_discriminator_128():
    _run_module("beta_discriminator_128", O.patch_disc_template(p="model"),
                lambda sd, x: O.patch_discriminator(sd, x, p="model"))


# ---- hierarchical modules -------------------------------------------------------------------
def _attn_t(c):
    t = {}
    O._t_attn(t, "a", c)
    return {k[2:]: v for k, v in t.items()}


@pytest.mark.parametrize("c", [128, 256])
def test_hier_attention(c):
    _run_module(f"hier_attention{c}", _attn_t(c),
                lambda sd, x: O.self_attention({"a." + k: v for k, v in sd.items()}, "a", x))


def test_hier_encoder():
    t = {k[len("encoder."):]: v for k, v in O.hier_template().items() if k.startswith("encoder.")}
    _run_module("hier_encoder", t, lambda sd, x: O.hier_encoder({"encoder." + k: v for k, v in sd.items()}, x))


def test_hier_discriminator():
    t = {k[len("discriminator."):]: v for k, v in O.hier_template().items() if k.startswith("discriminator.")}
    _run_module("hier_discriminator", t,
                lambda sd, x: O.hier_discriminator({"discriminator." + k: v for k, v in sd.items()}, x))


@pytest.mark.parametrize("j,attn", [(0, True), (1, True), (2, False)])
def test_hier_decoder_block(j, attn):
    pre = f"decoder.decoders.{j}."
    t = {k[len("decoder."):]: v for k, v in O.hier_template().items() if k.startswith(pre)}
    # the golden was captured on the sub-module alone: its own key names are relative to it,
    # the digests were stored with the 'decoders.j.' prefix
    # the golden filled the whole HierarchicalDecoder (keys 'decoders.j....') and ran one sub-block
    g = load_golden(f"hier_dec_block{j}")
    sd = leaves({k[len(f"decoders.{j}."):]: v for k, v in fill_state_dict(t).items()})
    x = as_t(g["x"]).clone().requires_grad_(True)
    y = O.hier_decoder_block({pre + k: v for k, v in sd.items()}, j, x, attn)
    (y * as_t(g["cot0"])).sum().backward()
    close_l2(y, g["y0"], g["noise:y0"], what="y")
    close_l2(x.grad, g["dx"], g["noise:dx"], what="dx")
    check_digests(g, sd, lambda k: k[len(f"decoders.{j}."):])


def test_hier_output_conv():
    t = {k[len("decoder."):]: v for k, v in O.hier_template().items() if k.startswith("decoder.output_conv.")}
    g = load_golden("hier_output_conv")
    sd = leaves({k[len("output_conv."):]: v for k, v in fill_state_dict(t).items()})
    x = as_t(g["x"]).clone().requires_grad_(True)
    y = O.hier_output_conv({"decoder.output_conv." + k: v for k, v in sd.items()}, x)
    (y * as_t(g["cot0"])).sum().backward()
    close_l2(y, g["y0"], g["noise:y0"], what="y")
    close_l2(x.grad, g["dx"], g["noise:dx"], what="dx")
    check_digests(g, sd, lambda k: k[len("output_conv."):])


def test_hier_full_step():
    """hierarchical_vae_gan/main.py:75-111 with the documented shims D2 + D1 (make_golden.g_hier_step: the
    reference modules and loss_function do the arithmetic)."""
    g = load_golden("hier_step_128")
    sd = fill_state_dict(O.hier_template())
    images = as_t(g["images"])
    eps = hier_step_eps(2, 128)
    opt = {"g": {}, "d": {}}
    for it in range(2):
        r = O.hier_train_step(sd, images, eps[it], O.HIER_DEFAULT_CFG, opt)
        check_step(g, it, r, ("d_loss", "d_kl", "g_loss", "recon", "kl", "adv"))
    check_state(g, sd, max(float(g["noise:it1_recon_img"]), 1e-4))


def test_beta_full_step():
    """beta_vae_gans/main.py:75-108 with shim D1; 172 M parameters, two iterations at 2x3x256x256."""
    from vae_gan_amd.testing import synthetic_images
    g = load_golden("beta_step_256")
    sd = fill_state_dict(O.beta_template())
    images = synthetic_images(2, 256, seed=int(g["images_seed"]))
    opt = {"g": {}, "d": {}}
    for it in range(2):
        r = O.beta_train_step(sd, images, as_t(g[f"eps{it}"]), O.BETA_DEFAULT_CFG, opt)
        check_step(g, it, r, ("d_loss", "g_loss", "recon", "kl", "adv"))
    check_state(g, sd, max(float(g["noise:it1_recon_img"]), 1e-4))


def test_template_sizes():
    """state_dict contract, SURVEY.md §8(b): 199 / 200 / 267 tensors."""
    assert len(O.beta_template()) == 199
    assert len(O.vq_template()) == 200
    assert len(O.hier_template()) == 267
