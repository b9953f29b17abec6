"""GPU tier: the shortlist codebook search (vg_vq_assign_shortlist: bf16-MFMA candidates + exact-f32 distances) gives
the indices of the exact search (vg_vq_assign) BIT FOR BIT - random and trained-like codebooks, ragged sizes, exact
ties (duplicate codes: smallest index), near-ties (codes one ulp apart), more candidates than the list holds (the
fallback), and the reference-made golden at K = 8192 / D = 256."""
import ctypes as C
import pytest
import torch

from conftest import load_golden, as_t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import vae_gan_amd.lib as L
    return L.load()


def _both(lib, z, cb):
    from vae_gan_amd import lib as L
    rows, d = z.shape
    k = cb.shape[0]
    ref = torch.full((rows,), -1, dtype=torch.int64, device="cuda")
    norms = torch.empty(k, dtype=torch.float32, device="cuda")
    L.check(lib.vg_vq_assign(z.data_ptr(), cb.data_ptr(), rows, k, d, ref.data_ptr(), norms.data_ptr(), L.stream()), "vq_assign")
    out = torch.full((rows,), -1, dtype=torch.int64, device="cuda")
    nbytes = lib.vg_vq_shortlist_ws_bytes(rows, k, d)
    assert nbytes > 0
    ws = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")       # nothing may depend on what the workspace held
    flag = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    L.check(lib.vg_vq_assign_shortlist(z.data_ptr(), cb.data_ptr(), rows, k, d, out.data_ptr(), ws.data_ptr(), flag.data_ptr(),
                                       L.stream()), "vq_assign_shortlist")
    torch.cuda.synchronize()
    return ref, out, int(flag.item())


@pytest.mark.parametrize("rows,k,d", [(65536, 8192, 256), (4096, 8192, 256), (1000, 1000, 64), (5000, 2050, 128), (140000, 512, 64),
                                      (33, 40, 256)])
@pytest.mark.parametrize("kind", ["init", "trained"])
def test_shortlist_matches_exact_search(lib, rows, k, d, kind):
    g = torch.Generator(device="cuda").manual_seed(rows + k + d)
    if kind == "init":          # the reference's initial state: codes uniform in +-1/K, latents of order one
        cb = (torch.rand(k, d, device="cuda", generator=g) * 2 - 1) / k
        z = torch.randn(rows, d, device="cuda", generator=g)
    else:                       # latents scattered around the codes
        cb = torch.randn(k, d, device="cuda", generator=g) * 0.5
        pick = torch.randint(0, k, (rows,), device="cuda", generator=g)
        z = cb[pick] + 0.3 * torch.randn(rows, d, device="cuda", generator=g)
    ref, out, flag = _both(lib, z.contiguous(), cb.contiguous())
    assert torch.equal(ref, out), f"{int((ref != out).sum())} of {rows} indices differ"
    assert flag == 0            # the bound was not hit: the bf16 + exact path produced them


def test_shortlist_mixed_code_norms(lib):
    """A codebook a few EMA updates after initialisation: some codes have moved to the latents' scale, most are still
    ~1/K.  The error bound is taken per code, so a small code next to a large one in the same tile of 32 is not a
    candidate because of its neighbour (a tile-wide bound overflowed the lists here and fell back)."""
    g = torch.Generator(device="cuda").manual_seed(9)
    k, d, rows = 8192, 256, 16384
    cb = (torch.rand(k, d, device="cuda", generator=g) * 2 - 1) / k
    moved = torch.randperm(k, device="cuda", generator=g)[:600]
    cb[moved] = torch.randn(600, d, device="cuda", generator=g) * 0.6
    z = torch.randn(rows, d, device="cuda", generator=g) * 0.6
    z[::2] = cb[moved[torch.randint(0, 600, (rows // 2,), device="cuda", generator=g)]] + 0.2 * torch.randn(rows // 2, d, device="cuda", generator=g)
    z[1::4] *= 0.02                                              # latents for which the small codes are nearest
    ref, out, flag = _both(lib, z.contiguous(), cb.contiguous())
    assert torch.equal(ref, out), f"{int((ref != out).sum())} of {rows} indices differ"
    assert flag == 0
    assert 0 < int(torch.isin(ref, moved).sum()) < rows          # both kinds of codes win somewhere


def test_shortlist_ties_and_near_ties(lib):
    """Duplicate codes tie exactly (the smallest index wins), codes one ulp apart in one coordinate are near-ties that
    only the exact-f32 arithmetic orders: both must come out as in the exact search."""
    g = torch.Generator(device="cuda").manual_seed(5)
    k, d, rows = 4096, 256, 8192
    cb = torch.randn(k, d, device="cuda", generator=g) * 0.5
    cb[1::4] = cb[0::4]                                          # exact duplicates
    near = cb[0::4].clone()
    near[:, 7] = torch.nextafter(near[:, 7], near[:, 7] + 1)     # one ulp away in one coordinate
    cb[2::4] = near
    near2 = cb[0::4].clone()
    near2[:, 100] = torch.nextafter(near2[:, 100], near2[:, 100] - 1)
    cb[3::4] = near2
    pick = torch.randint(0, k, (rows,), device="cuda", generator=g)
    z = cb[pick] + 0.05 * torch.randn(rows, d, device="cuda", generator=g)
    ref, out, flag = _both(lib, z.contiguous(), cb.contiguous())
    assert torch.equal(ref, out), f"{int((ref != out).sum())} of {rows} indices differ"
    assert flag == 0
    assert int((ref % 4 == 1).sum()) == 0                        # a duplicate never beats the code it copies


def test_shortlist_overflow_falls_back_to_exact_search(lib):
    """Forty copies of every code: every row has more candidates than the list holds, the call redoes the search
    with the exact kernel and says so."""
    g = torch.Generator(device="cuda").manual_seed(6)
    base = torch.randn(200, 64, device="cuda", generator=g)
    cb = base.repeat_interleave(40, dim=0).contiguous()          # 8000 codes
    z = base[torch.randint(0, 200, (3000,), device="cuda", generator=g)] + 0.01 * torch.randn(3000, 64, device="cuda", generator=g)
    ref, out, flag = _both(lib, z.contiguous(), cb)
    assert flag == 1
    assert torch.equal(ref, out)
    assert int((ref % 40).sum()) == 0                            # first of the forty


def test_shortlist_golden_large_codebook(lib):
    """The reference-made golden at K = 8192 / D = 256 (tests/golden/vq_quantizer_c.npz): indices equal to the
    reference's wherever the exact search's are (near-ties are documented in test_gpu_vq.py)."""
    from vae_gan_amd.testing import fill_state_dict
    from vae_gan_amd.vq_vae_gan.model import VectorQuantizer
    gold = load_golden("vq_quantizer_c")
    sd = fill_state_dict(VectorQuantizer(8192, 256, 0.25, 0.99).state_dict())
    cb = sd["embeddings.weight"].cuda().contiguous()
    z = as_t(gold["z"]).permute(0, 2, 3, 1).reshape(-1, 256).contiguous().cuda()
    ref, out, flag = _both(lib, z, cb)
    assert torch.equal(ref, out) and flag == 0
    assert int((out.cpu() != as_t(gold["idx"]).flatten()).sum()) <= max(1, out.numel() // 500)


def test_shortlist_rejects_bad_arguments(lib):
    from vae_gan_amd import lib as L
    assert lib.vg_vq_shortlist_ws_bytes(100, 100, 48) == 0
    t = torch.zeros(1024, device="cuda")
    i = torch.zeros(16, dtype=torch.int64, device="cuda")
    assert lib.vg_vq_assign_shortlist(t.data_ptr(), t.data_ptr(), 4, 4, 48, i.data_ptr(), t.data_ptr(), None, L.stream()) != 0


def test_shortlist_worst_case_rounding(lib):
    """Aligned worst-case bf16 rounding: every coordinate of z and of the winning code W sits just beside
    a bf16 midpoint, with the signs arranged so that ALL rounding errors make z.W look smaller (31 coordinates positive
    and rounding down, 31 with a negative product and rounding up in magnitude), while the exact z.W nearly cancels
    (one exactly representable coordinate tips it to ~ +5e-4 .. +9e-4); the runner-up X = -W gets the same error with the
    opposite sign.  The bf16 distances then order the pair wrongly by ~2^-6 |z| (|W| + |X|) = 1.94 against an exact gap of
    4 z.W ~ 3e-3 the other way: the per-code bound must be (2^-6 + 2^-15) |z||E_k| for pass 2 to keep W.  With round 2's
    constant (1.02 * 2^-7: the two bounds sum to 1.0) W is dropped and X comes out."""
    d, k, rows = 64, 64, 48
    u = 2.0 ** -23
    w = torch.zeros(d, dtype=torch.float64)
    a0, b0 = 1 + 2.0 ** -8 - 8 * u, 1 + 2.0 ** -8 + 8 * u      # just below / just above the midpoint of 1 and 1 + 2^-7
    w[:31], w[31:62], w[62] = a0, -b0, 2.0 ** -5
    zs = torch.zeros(rows, d, dtype=torch.float64)
    for r in range(rows):                                    # the same construction at 48 distances from the midpoints
        e = (8 + r) * u
        zs[r, :31], zs[r, 31:62], zs[r, 62] = 1 + 2.0 ** -8 - e, 1 + 2.0 ** -8 + e, 2.0 ** -5
    assert float((zs @ w).min()) > 4e-4                      # W (code 0) beats -W (code 1) for every row, exactly
    g = torch.Generator().manual_seed(3)
    cb = torch.randn(k, d, generator=g, dtype=torch.float64) * 100.0        # everybody else is far away
    cb[0], cb[1] = w, -w
    zf, cf = zs.float().cuda().contiguous(), cb.float().cuda().contiguous()
    assert torch.equal(zf[:, :63].double().cpu(), zs[:, :63]) and torch.equal(cf[:2].double().cpu(), cb[:2])     # exact in f32
    ref, out, flag = _both(lib, zf, cf)
    assert torch.equal(ref.cpu(), torch.zeros(rows, dtype=torch.int64)), "the exact search must pick W (code 0)"
    assert torch.equal(ref, out), f"{int((ref != out).sum())} of {rows} worst-case rows lost their argmin"
    assert flag == 0
