"""CPU tier: the entropy-coder oracle (oracle/vq_codec_oracle.py) against the bitstreams the REFERENCE's
ArithmeticEncoder.encode produced (tests/golden/vq_codec.npz, made by make_golden.py:g_vq_codec), and its decode as
the inverse of that format."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import vq_codec_oracle as C

TAGS = ["k512_skewed", "k512_uniform_probs", "k8192", "k16_tiny", "k16_empty", "k512_one_symbol"]


def case(g, tag):
    K = int(g[f"{tag}:K"])
    idx = torch.from_numpy(g[f"{tag}:idx"])
    probs = torch.from_numpy(g[f"{tag}:probs"]) if f"{tag}:probs" in g else None
    return K, idx, probs, g[f"{tag}:bytes"].tobytes()


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_reproduces_reference_bitstream(tag):
    K, idx, probs, want = case(load_golden("vq_codec"), tag)
    if probs is not None:
        assert torch.equal(C.codebook_usage(idx, K), probs)
    assert bytes(C.encode(idx, probs, K)) == want


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_decode_inverts_reference_bitstream(tag):
    K, idx, probs, data = case(load_golden("vq_codec"), tag)
    assert torch.equal(C.decode(data, idx.shape, probs, K), idx)


def test_reference_decode_is_not_an_inverse():
    """Recorded fact about the reference (model.py:491, :530-532), the reason decode is defined as encode's inverse."""
    g = load_golden("vq_codec")
    assert not bool(g["k512_skewed:ref_decode_roundtrips"]) and not bool(g["k16_tiny:ref_decode_roundtrips"])


def test_histc_binning_quirk_at_8192():
    p = C.codebook_usage(torch.arange(8192), 8192)
    assert float(p[8190]) == 0.0 and float(p[8191]) == 2.0 / 8192


def test_zero_probability_symbol_is_rejected():
    with pytest.raises(ValueError):
        C.encode(torch.tensor([8190]), C.codebook_usage(torch.tensor([8190]), 8192), 8192)


def test_roundtrip_random_streams():
    rng = np.random.default_rng(5)
    for K, n, z in [(2, 50, 0.0), (3, 77, 1.0), (512, 300, 1.5), (8192, 200, 1.1), (16, 1000, 3.0)]:
        w = 1.0 / np.arange(1, K + 1) ** z
        idx = torch.from_numpy(rng.choice(K, size=n, p=w / w.sum()))
        probs = torch.bincount(idx, minlength=K).float()
        probs = probs / probs.sum()
        data = C.encode(idx, probs, K)
        assert torch.equal(C.decode(data, (n,), probs, K), idx)
