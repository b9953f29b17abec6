"""GPU tier: vg_channel_sum_noalloc (the entry point that allocates nothing: static device scratch inside the
library) against vg_channel_sum (caller workspace), through the C ABI, on two calls in a row, the second with a
smaller input (stale scratch contents of the first call must not leak into it)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, L, t, c_valid, accumulate, seed):
    cs = t.shape[-1]
    rows = t.numel() // cs
    g = torch.Generator(device="cuda").manual_seed(seed)
    base = torch.randn(c_valid, device="cuda", generator=g) if accumulate else torch.full((c_valid,), float("nan"), device="cuda")
    want, got = base.clone(), base.clone()
    ws = torch.empty(L.STATS_REPLICAS * cs, dtype=torch.float64, device="cuda")
    L.check(lib.vg_channel_sum(t.data_ptr(), rows, cs, c_valid, L.vg_dtype(t.dtype), want.data_ptr(), ws.data_ptr(),
                               accumulate, L.stream()), "channel_sum")
    L.check(lib.vg_channel_sum_noalloc(t.data_ptr(), rows, cs, c_valid, L.vg_dtype(t.dtype), got.data_ptr(), accumulate,
                                       L.stream()), "channel_sum_noalloc")
    torch.cuda.synchronize()
    return want, got


def test_noalloc_channel_sum_matches_workspace_entry_point():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import lib as L
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    # first call: the largest supported stride (1024 channels, bf16), many rows; second call: smaller in every way
    big = torch.randn(3000, 1024, device="cuda", generator=g).to(torch.bfloat16)
    small = torch.randn(257, 48, device="cuda", generator=g)               # f32, stride 48, 41 valid channels
    thin = torch.randn(70000, 4, device="cuda", generator=g).to(torch.bfloat16)      # the bf16 image-side layout (8-byte pixels)
    for t, c_valid, acc in ((big, 1024, 0), (small, 41, 1), (big[:, :512].contiguous(), 500, 0), (small[:5].contiguous(), 48, 0),
                            (thin, 3, 0), (thin[:100].contiguous(), 3, 1)):
        want, got = _both(lib, L, t, c_valid, acc, seed=c_valid)
        ref = t.double().sum(0)[:c_valid]
        if acc:
            gg = torch.Generator(device="cuda").manual_seed(c_valid)
            ref = ref + torch.randn(c_valid, device="cuda", generator=gg).double()
        assert torch.isfinite(got).all()
        # same kernels, same launch geometry: only the order of the f64 atomics differs between the two calls
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6 * float(ref.abs().max())), float((got - want).abs().max())
        assert torch.allclose(got.double(), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))


def test_noalloc_rejects_what_the_static_scratch_cannot_hold():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import lib as L
    lib = L.load()
    t = torch.zeros(4, 2048, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(2048, device="cuda")
    rc = lib.vg_channel_sum_noalloc(t.data_ptr(), 4, 2048, 2048, L.VG_BF16, out.data_ptr(), 0, L.stream())
    assert rc != 0 and "exceeds the static scratch" in L.last_error()
    # the workspace entry point takes the same shape
    ws = torch.empty(L.STATS_REPLICAS * 2048, dtype=torch.float64, device="cuda")
    L.check(lib.vg_channel_sum(t.data_ptr(), 4, 2048, 2048, L.VG_BF16, out.data_ptr(), ws.data_ptr(), 0, L.stream()), "channel_sum")
    torch.cuda.synchronize()
