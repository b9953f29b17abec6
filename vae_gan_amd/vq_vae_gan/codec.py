"""VQ index entropy coding on the MI355X — drop-in for vq_vae_gan/model.py:336-673.

Same names and argument meaning as the reference: ``ArithmeticEncoder(num_symbols, precision=32)`` with
``encode(indices, probs=None) -> bytearray`` and ``decode(byte_array, shape, probs=None) -> LongTensor``,
``calculate_codebook_usage``, ``compress_vq_indices``, ``decompress_vq_indices``, ``compress_image``,
``decompress_image``.  The bytes are the reference's, bit for bit (tests/golden/vq_codec.npz).  Histogram, table
and coder run in libvg_hip.so (csrc/codec.hip); indices and probabilities stay on the device, only the finished
byte string crosses to the host.  There is no CPU fallback.

Differences from the reference, all documented in DESIGN.md §7:
  * the reference never imports numpy, so its encode raises NameError (model.py:378); this one runs;
  * ``decode`` is the inverse of ``encode``; the reference's decode is not (model.py:491, :530-532);
  * a symbol whose table probability is 0 raises ValueError (the reference's encode loop never terminates on it);
  * ``encode_streams`` / ``decode_streams`` code many independent streams (e.g. one per image) in one launch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import lib as L
from ..ops import _require_cuda

_STATUS = {1: "a symbol with probability 0 in the table (the reference's encode loop never terminates on it; "
              "with calculate_codebook_usage at num_embeddings=8192 index 8190 is binned as 8191 by torch.histc — "
              "pass exact_bins=True)",
           2: "stream capacity exceeded", 3: "index outside [0, num_symbols)", 4: "coder state outside the supported range"}


def calculate_codebook_usage(indices: torch.Tensor, num_embeddings: int, exact_bins: bool = False) -> torch.Tensor:
    """model.py:540-554; float32 probabilities on the device.  exact_bins=False keeps torch.histc's binning."""
    _require_cuda(indices, "calculate_codebook_usage")
    flat = indices.flatten().to(torch.int64).contiguous()
    probs = torch.empty(num_embeddings, dtype=torch.float32, device=flat.device)
    ws = torch.empty(num_embeddings + 1, dtype=torch.int32, device=flat.device)
    L.check(L.load().vg_codebook_usage(flat.data_ptr(), flat.numel(), num_embeddings, int(exact_bins), ws.data_ptr(),
                                       probs.data_ptr(), L.stream()), "codebook_usage")
    return probs


class ArithmeticEncoder:
    """model.py:337-537."""

    def __init__(self, num_symbols, precision=32):
        if precision != 32:
            raise ValueError("the HIP coder implements the reference's default precision=32 only")
        self.num_symbols = num_symbols
        self.precision = precision
        self.max_val = (1 << precision) - 1
        self.half = 1 << (precision - 1)
        self.quarter = 1 << (precision - 2)

    # -- table ----------------------------------------------------------------------------------------------
    def table(self, probs, device) -> torch.Tensor:
        """Cumulative float64 table [num_symbols + 1] on the device (model.py:369-380)."""
        cum = torch.empty(self.num_symbols + 1, dtype=torch.float64, device=device)
        p = None
        if probs is not None:
            _require_cuda(probs, "ArithmeticEncoder: probs")
            p = probs.to(torch.float32).contiguous()
            if p.numel() != self.num_symbols:
                raise ValueError(f"probs has {p.numel()} entries for {self.num_symbols} symbols")
        L.check(L.load().vg_ac_table(L.ptr(p), self.num_symbols, cum.data_ptr(), L.stream()), "ac_table")
        return cum

    # -- many streams per launch ------------------------------------------------------------------------------
    def encode_streams(self, indices: torch.Tensor, probs=None):
        """indices [S, ...]: S independent streams sharing one table.  Returns (bytes uint8 [S, cap], nbytes int32 [S]),
        both on the device."""
        _require_cuda(indices, "ArithmeticEncoder.encode")
        lib = L.load()
        s = indices.shape[0]
        flat = indices.reshape(s, -1).to(torch.int64).contiguous()
        n = flat.shape[1]
        cum = self.table(probs, flat.device)
        cap = int(lib.vg_ac_stream_capacity(n))
        out = torch.empty(s, cap, dtype=torch.uint8, device=flat.device)
        nbytes = torch.empty(s, dtype=torch.int32, device=flat.device)
        status = torch.empty(s, dtype=torch.int32, device=flat.device)
        L.check(lib.vg_ac_encode(flat.data_ptr() if n else None, s, n, cum.data_ptr(), self.num_symbols, out.data_ptr(), cap,
                                 nbytes.data_ptr(), status.data_ptr(), L.stream()), "ac_encode")
        bad = status.nonzero()
        if bad.numel():
            code = int(status[bad[0, 0]])
            raise ValueError(f"arithmetic encode, stream {int(bad[0, 0])}: {_STATUS.get(code, code)}")
        return out, nbytes

    def decode_streams(self, data: torch.Tensor, nbytes: torch.Tensor, symbols_per_stream: int, probs=None) -> torch.Tensor:
        """Inverse of encode_streams: int64 [S, symbols_per_stream] on the device."""
        _require_cuda(data, "ArithmeticEncoder.decode")
        s, cap = data.shape
        cum = self.table(probs, data.device)
        idx = torch.empty(s, symbols_per_stream, dtype=torch.int64, device=data.device)
        status = torch.empty(s, dtype=torch.int32, device=data.device)
        L.check(L.load().vg_ac_decode(data.data_ptr(), nbytes.data_ptr(), s, cap, symbols_per_stream, cum.data_ptr(),
                                      self.num_symbols, idx.data_ptr() if symbols_per_stream else None, status.data_ptr(),
                                      L.stream()), "ac_decode")
        bad = status.nonzero()
        if bad.numel():
            code = int(status[bad[0, 0]])
            raise ValueError(f"arithmetic decode, stream {int(bad[0, 0])}: {_STATUS.get(code, code)}")
        return idx

    # -- the reference's surface ---------------------------------------------------------------------------------
    def encode(self, indices, probs=None) -> bytearray:
        """model.py:355-445: the whole tensor is one stream."""
        out, nbytes = self.encode_streams(indices.reshape(1, -1), probs)
        return bytearray(out[0, :int(nbytes[0])].cpu().numpy().tobytes())

    def decode(self, byte_array, shape, probs=None) -> torch.Tensor:
        """model.py:447-537 (see the module docstring): int64 tensor of `shape`, on the device of `probs` or cuda:0."""
        device = probs.device if probs is not None else torch.device("cuda", torch.cuda.current_device())
        raw = np.frombuffer(bytes(byte_array), dtype=np.uint8)
        data = torch.zeros(1, max(len(raw), 1), dtype=torch.uint8, device=device)
        if len(raw):
            data[0, :len(raw)] = torch.from_numpy(raw.copy()).to(device)
        nbytes = torch.tensor([len(raw)], dtype=torch.int32, device=device)
        n = int(np.prod(shape))
        return self.decode_streams(data, nbytes, n, probs).reshape(tuple(shape))


def compress_vq_indices(model, indices):
    """model.py:556-576."""
    probs = calculate_codebook_usage(indices, model.num_embeddings)
    return ArithmeticEncoder(model.num_embeddings).encode(indices, probs), probs


def decompress_vq_indices(model, compressed, shape, probs=None):
    """model.py:578-597."""
    return ArithmeticEncoder(model.num_embeddings).decode(compressed, shape, probs)


def compress_image(model, image, device='cuda'):
    """model.py:599-635."""
    model.eval()
    image = image.to(device)
    with torch.no_grad():
        _, _, _, indices = model.encode(image, training=False)
    compressed, probs = compress_vq_indices(model, indices)
    original_size = image.numel() * 8
    compressed_size = len(compressed) * 8
    metadata = {'shape': indices.shape, 'probs': probs, 'compression_ratio': original_size / compressed_size,
                'bpp': compressed_size / (image.size(2) * image.size(3) * image.size(0))}
    return compressed, metadata


def decompress_image(model, compressed, metadata, device='cuda'):
    """model.py:637-673."""
    model.eval()
    shape, probs = metadata['shape'], metadata['probs']
    indices = decompress_vq_indices(model, compressed, shape, probs).to(device)
    with torch.no_grad():
        embeddings = model.vector_quantizer.embeddings(indices.view(-1))
        batch_size, height, width = shape
        z_q = embeddings.view(batch_size, height, width, model.embedding_dim).permute(0, 3, 1, 2).contiguous()
        return model.decode(z_q)
