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
