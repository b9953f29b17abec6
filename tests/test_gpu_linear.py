"""GPU tier: the wide-Linear backward kernels (csrc/linear.hip: fc_mu / fc_logvar of the beta model) through the C ABI
against float64 matrix products: even / ragged batches, a batch longer than one register pass, accumulate."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,fin,fout", [(32, 8192, 128), (5, 4098, 48), (40, 6000, 128), (2, 131072, 128)])
def test_linear_backward_against_float64(b, fin, fout):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import lib as L
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(b + fin + fout)
    x = torch.randn(b, fin, device="cuda", generator=g)
    w = torch.randn(fout, fin, device="cuda", generator=g) / fin ** 0.5
    gy = torch.randn(b, fout, device="cuda", generator=g)
    dw = torch.full((fout, fin), float("nan"), device="cuda")
    L.check(lib.vg_linear_wgrad(x.data_ptr(), gy.data_ptr(), dw.data_ptr(), b, fin, fout, 0, L.stream()), "linear_wgrad")
    acc = dw.clone()
    L.check(lib.vg_linear_wgrad(x.data_ptr(), gy.data_ptr(), acc.data_ptr(), b, fin, fout, 1, L.stream()), "linear_wgrad")
    dx = torch.full((b, fin), float("nan"), device="cuda")
    L.check(lib.vg_linear_dgrad(gy.data_ptr(), w.data_ptr(), dx.data_ptr(), b, fin, fout, L.stream()), "linear_dgrad")
    torch.cuda.synchronize()
    dw_ref = gy.double().t() @ x.double()
    dx_ref = gy.double() @ w.double()
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())
    assert rel(dw, dw_ref) <= 2e-6, rel(dw, dw_ref)
    assert rel(dx, dx_ref) <= 2e-6, rel(dx, dx_ref)
    assert rel(acc, 2 * dw_ref) <= 2e-6
    # no atomics: a second run gives the same bits
    dw2 = torch.empty_like(dw)
    L.check(lib.vg_linear_wgrad(x.data_ptr(), gy.data_ptr(), dw2.data_ptr(), b, fin, fout, 0, L.stream()), "linear_wgrad")
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)


def test_linear_layer_uses_the_wide_kernels_and_matches_autograd():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(4, 8192, device="cuda", generator=g, requires_grad=True)
    w = (torch.randn(64, 8192, device="cuda", generator=g) / 90).requires_grad_()
    bias = torch.randn(64, device="cuda", generator=g).requires_grad_()
    y = ops.linear(x, w, bias)
    gy = torch.randn(4, 64, device="cuda", generator=g)
    y.backward(gy)
    xr, wr, br = (t.detach().double().requires_grad_() for t in (x, w, bias))
    (xr @ wr.t() + br).backward(gy.double())
    for mine, ref in ((y.detach(), (xr @ wr.t() + br).detach()), (x.grad, xr.grad), (w.grad, wr.grad), (bias.grad, br.grad)):
        assert float((mine.double() - ref).norm() / ref.norm()) <= 1e-5
