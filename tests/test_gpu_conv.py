"""GPU tier: the conv primitives through the raw C ABI against torch CPU float64 convolutions (forward, input
gradient, weight gradient, fused BN statistics, zero padding lanes), and the three kernel generations
(im2col-row, spatial-tile, ping-pong) against each other on the same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import gpu_ops_check as G  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-3)], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", G.CASES, ids=lambda c: "n%d_%dx%d_ci%d_co%d_k%ds%dp%d" % c)
def test_conv_primitives(gpu, case, dtype, tol):
    r = G.check_conv(*case, dtype, verbose=False)
    for k in ("fwd", "dgrad", "stat_sum", "stat_sq"):
        assert r[k] <= tol, (k, r[k])
    # f32 atomics over pixel splits; bf16 inputs are rounded before both sides, so the bound is the same order
    assert r["wgrad"] <= (tol if dtype == torch.float32 else 5e-3), r["wgrad"]
    assert r["pad0"] == 0.0


def test_kernel_generations_agree(gpu):
    import gpu_conv_xcheck as X
    assert X.main(exit=False) == 0


@pytest.mark.parametrize("case", [(8, 64, 64, 64, 64, 3, 1, 1), (4, 64, 64, 128, 256, 4, 2, 1), (2, 40, 36, 64, 64, 3, 1, 1)],
                         ids=lambda c: "n%d_%dx%d_ci%d_co%d_k%ds%dp%d" % c)
def test_spatial_tile_weight_gradient_is_bit_reproducible(gpu, case):
    """The spatial-tile weight-gradient kernels write one partial image per tile group and sum them in a fixed order
    (no float atomics): two launches on the same inputs give the same bits, and accumulate=1 adds exactly once."""
    import ctypes as C
    from vae_gan_amd import lib as L
    n, h, w, ci, co, k, stride, pad = case
    lib = L.load()
    d = G.make_desc(n, h, w, ci, co, k, stride, pad, torch.bfloat16)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(n, h, w, d.xcs, device="cuda", generator=g).bfloat16()
    gy = torch.randn(n, d.oh, d.ow, d.ycs, device="cuda", generator=g).bfloat16()
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(2):
        dw = torch.full((co, ci, k, k), float("nan"), dtype=torch.float32, device="cuda")
        ws.fill_(float("nan"))          # nothing may depend on what the workspace held
        L.check(lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, L.stream()), "wgrad")
        outs.append(dw)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    acc = outs[0].clone()
    L.check(lib.vg_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), acc.data_ptr(), ws.data_ptr(), 1, L.stream()), "wgrad")
    torch.cuda.synchronize()
    assert torch.equal(acc, outs[0] * 2)
