"""CPU oracle of the VQ index entropy coder.  TEST INFRASTRUCTURE ONLY: imported by tests/,
and by bench.py's cpu_baseline leg of the codec bench; the product path (vae_gan_amd/vq_vae_gan/codec.py ->
libvg_hip.so) never touches it.

Restates, in plain Python / numpy, the reference's
  * calculate_codebook_usage        vq_vae_gan/model.py:540-554  (torch.histc over float32 indices)
  * ArithmeticEncoder.encode        vq_vae_gan/model.py:355-445  (32-bit binary arithmetic coder, float64 table)
and states the INVERSE of encode for decoding.  The reference's own ArithmeticEncoder.decode (model.py:447-537) is
not an inverse of its encode: it re-uses the symbol counter `i` as its bit pointer (:491, :530-532) and searches the
symbol through a second, differently rounded expression (:494-498); tests/golden/vq_codec.npz records that it fails
to round-trip every non-trivial golden stream.  The format is therefore defined by encode(), and decode() here is
the exact inverse of that format (same table, same integer interval arithmetic).

Pinned by tests/golden/vq_codec.npz (bitstreams produced by the reference's encode, see make_golden.py:g_vq_codec).
"""
from __future__ import annotations

import numpy as np
import torch

PRECISION = 32
MAX_VAL = (1 << PRECISION) - 1
HALF = 1 << (PRECISION - 1)
QUARTER = 1 << (PRECISION - 2)


def codebook_usage(indices: torch.Tensor, num_embeddings: int) -> torch.Tensor:
    """model.py:540-554.  torch.histc bins float32(index) with floor(x / (K-1) * K) in float32: at K = 8192 index 8190
    lands in bin 8191 (8190/8191 rounds to 1 - 2^-13), so that symbol gets probability 0 — reference behaviour."""
    flat = indices.flatten()
    hist = torch.histc(flat.float(), bins=num_embeddings, min=0, max=num_embeddings - 1)
    return hist / hist.sum()


def cum_table(probs, num_symbols: int) -> np.ndarray:
    """model.py:369-380: float32 normalisation (none for the uniform default), then a sequential float64 running sum."""
    if probs is None:
        p = torch.ones(num_symbols) / num_symbols
    else:
        p = probs / probs.sum()
    p = p.cpu().numpy()
    cum = np.zeros(num_symbols + 1)
    for i in range(num_symbols):
        cum[i + 1] = cum[i] + p[i]
    return cum


def encode_with_table(flat_indices, cum: np.ndarray) -> bytearray:
    """model.py:382-445 on a prepared table."""
    low, high = 0, MAX_VAL
    bits = []
    pending = 0
    for symbol in flat_indices:
        symbol = int(symbol)
        rng = high - low + 1
        high = low + int(rng * cum[symbol + 1]) - 1
        low = low + int(rng * cum[symbol])
        if high < low:
            raise ValueError(f"symbol {symbol} has probability 0 in the table: the reference's loop never terminates here")
        while True:
            if high < HALF:
                bits.append(0)
                bits.extend([1] * pending)
                pending = 0
            elif low >= HALF:
                bits.append(1)
                bits.extend([0] * pending)
                pending = 0
                low -= HALF
                high -= HALF
            elif low >= QUARTER and high < 3 * QUARTER:
                pending += 1
                low -= QUARTER
                high -= QUARTER
            else:
                break
            low = low * 2
            high = high * 2 + 1
    pending += 1
    if low < QUARTER:
        bits.append(0)
        bits.extend([1] * pending)
    else:
        bits.append(1)
        bits.extend([0] * pending)
    out = bytearray()
    for i in range(0, len(bits), 8):
        byte = 0
        for j in range(8):
            if i + j < len(bits) and bits[i + j]:
                byte |= 1 << (7 - j)
        out.append(byte)
    return out


def encode(indices: torch.Tensor, probs, num_symbols: int) -> bytearray:
    """ArithmeticEncoder(num_symbols).encode(indices, probs), model.py:355-445."""
    flat = indices.flatten().cpu().numpy().astype(int)
    return encode_with_table(flat, cum_table(probs, num_symbols))


def decode_with_table(data, n: int, cum: np.ndarray) -> np.ndarray:
    """Exact inverse of encode_with_table: bits past the end of the stream read as 0."""
    K = len(cum) - 1
    nbits = len(data) * 8

    def bit(p):
        return (data[p >> 3] >> (7 - (p & 7))) & 1 if p < nbits else 0

    value = 0
    for p in range(PRECISION):
        value = (value << 1) | bit(p)
    pos = PRECISION
    low, high = 0, MAX_VAL
    out = np.zeros(n, dtype=np.int64)
    for t in range(n):
        rng = high - low + 1
        # largest s with low + int(rng*cum[s]) <= value  (the encoder's own interval bounds; monotone in s)
        lo_s, hi_s = 0, K
        while hi_s - lo_s > 1:
            mid = (lo_s + hi_s) // 2
            if low + int(rng * cum[mid]) <= value:
                lo_s = mid
            else:
                hi_s = mid
        s = lo_s
        out[t] = s
        high = low + int(rng * cum[s + 1]) - 1
        low = low + int(rng * cum[s])
        while True:
            if high < HALF:
                pass
            elif low >= HALF:
                low -= HALF
                high -= HALF
                value -= HALF
            elif low >= QUARTER and high < 3 * QUARTER:
                low -= QUARTER
                high -= QUARTER
                value -= QUARTER
            else:
                break
            low = low * 2
            high = high * 2 + 1
            value = value * 2 + bit(pos)
            pos += 1
    return out


def decode(data, shape, probs, num_symbols: int) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(decode_with_table(bytes(data), n, cum_table(probs, num_symbols)).reshape(shape))
