"""torch.autograd.Function wrappers around the C ABI of libvg_hip.so.

PyTorch is the tensor allocator and autograd tape only: every arithmetic op of the hot path is
a hand-written HIP kernel reached through ``vae_gan_amd.lib``.  Internal activations are NHWC
tensors ``[N, H, W, Cs]`` (``Cs`` = channel count padded to a 64-byte granule) of the compute
dtype (bf16 or f32); module boundaries convert from/to the reference's NCHW f32 tensors.
"""
from __future__ import annotations

import ctypes as C
import os as _os
import weakref
from typing import Optional

import torch

from . import lib as L

# bumped by the fused optimizers whenever parameters change outside torch's version counter
PARAM_EPOCH = [0]

_COMPUTE_DTYPE = [torch.bfloat16]


def set_compute_dtype(dt: torch.dtype) -> None:
    """bf16 (MFMA bf16, the benchmarked mode) or f32 (exact-f32 MFMA, the parity mode).

    The f32 mode is also the reproducible one: it turns ``vg_set_deterministic`` on (statistics through fixed-point
    accumulators, no f32 atomics between blocks), so two runs of the same step give the same bits.  ``VG_DETERMINISTIC=1``
    asks for that in bf16 too (a few per cent slower), ``VG_DETERMINISTIC=0`` keeps the f32 mode on the faster atomics."""
    L.vg_dtype(dt)
    _COMPUTE_DTYPE[0] = dt
    env = _os.environ.get("VG_DETERMINISTIC")
    set_deterministic(dt == torch.float32 if env is None else env not in ("0", ""))


def set_deterministic(on: bool) -> None:
    L.load().vg_set_deterministic(1 if on else 0)


def deterministic() -> bool:
    return bool(L.load().vg_get_deterministic())


def compute_dtype() -> torch.dtype:
    return _COMPUTE_DTYPE[0]


class KernelProfile:
    """Optional per-launch HIP-event timing of the conv kernels (bench.py's live roofline numbers).
    Events are recorded on torch's current stream, which is the stream the kernels are launched on."""

    def __init__(self, shapes: bool = False, only_tags=None):
        self.records = []     # (kernel name, algorithmic flops, start event, end event)
        self.shapes = shapes  # key the summary by layer shape as well (diagnostics)
        self.only_tags = only_tags   # set of layer-shape tags to time (None = every conv launch)
        self.tag_kernel = {}  # layer-shape tag -> kernel variant the library dispatched

    def launch(self, name, flops, fn, tag=None):
        if self.only_tags is not None and tag not in self.only_tags:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        if name is None:      # gather-GEMM launch: ask the library which kernel variant it dispatched
            name = L.load().vg_last_conv_kernel().decode()
        elif name == "wgrad":   # weight gradient: likewise (the launch also runs the ordered sum / unpack kernel)
            name = L.load().vg_last_wgrad_kernel().decode() + "+unpack"
        if tag is not None:
            self.tag_kernel[tag] = name
        if self.shapes and tag is not None:
            name += " " + tag
        self.records.append((name, flops, e0, e1))
        return rc

    def summary(self):
        """name -> dict(launches, flops, ms); call after torch.cuda.synchronize()."""
        out = {}
        for name, flops, e0, e1 in self.records:
            d = out.setdefault(name, {"launches": 0, "flops": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += e0.elapsed_time(e1)
        return out


PROFILE = [None]   # set PROFILE[0] = KernelProfile() to time conv launches


def _shape_tag(kind, d) -> str:
    return f"{kind} n{d.n} {d.h}x{d.w} {d.ci}->{d.co} k{d.r}s{d.stride}"


def _conv_flops(d) -> float:
    return 2.0 * d.n * d.oh * d.ow * d.co * d.ci * d.r * d.s


def _timed(name, flops, fn, tag=None):
    prof = PROFILE[0]
    return fn() if prof is None else prof.launch(name, flops, fn, tag)


def _npad(n: int) -> int:
    return 32 if n <= 32 else (64 if n <= 64 else (n + 127) // 128 * 128)


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the HIP path needs a tensor on an MI355X device (got {t.device}); "
                           "there is no CPU fallback")


# ----------------------------------------------------------------------------------------------
# layout conversion
# ----------------------------------------------------------------------------------------------
def _nchw_to_nhwc(x: torch.Tensor, cs: int, dt: torch.dtype) -> torch.Tensor:
    n, c, h, w = x.shape
    out = torch.empty(n, h, w, cs, dtype=dt, device=x.device)
    L.check(L.load().vg_nchw_to_nhwc(x.data_ptr(), out.data_ptr(), n, c, h, w, cs, L.vg_dtype(dt), L.stream()), "nchw_to_nhwc")
    return out


def _nhwc_to_nchw(x: torch.Tensor, c: int) -> torch.Tensor:
    n, h, w, cs = x.shape
    out = torch.empty(n, c, h, w, dtype=torch.float32, device=x.device)
    L.check(L.load().vg_nhwc_to_nchw(x.data_ptr(), out.data_ptr(), n, c, h, w, cs, L.vg_dtype(x.dtype), L.stream()), "nhwc_to_nchw")
    return out


class ToNHWC(torch.autograd.Function):
    """NCHW f32 -> NHWC compute dtype, channels zero-padded to ``cs``."""

    @staticmethod
    def forward(ctx, x, cs, dt):
        _require_cuda(x, "to_nhwc")
        ctx.c = x.shape[1]
        return _nchw_to_nhwc(x.contiguous().float(), cs, dt)

    @staticmethod
    def backward(ctx, g):
        return _nhwc_to_nchw(g.contiguous(), ctx.c), None, None


class ToNCHW(torch.autograd.Function):
    """NHWC compute dtype -> NCHW f32 (first ``c`` channels)."""

    @staticmethod
    def forward(ctx, x, c):
        ctx.cs, ctx.dt = x.shape[3], x.dtype
        return _nhwc_to_nchw(x.contiguous(), c)

    @staticmethod
    def backward(ctx, g):
        return _nchw_to_nhwc(g.contiguous().float(), ctx.cs, ctx.dt), None


THIN_CS = 4     # channel stride of the 3-channel image-side tensors in bf16 mode (csrc/thin.hip): 8 bytes per pixel


def _thin_enabled() -> bool:
    return not _os.environ.get("VG_NO_THIN")


def to_nhwc(x: torch.Tensor, dt: Optional[torch.dtype] = None) -> torch.Tensor:
    dt = dt or compute_dtype()
    c = x.shape[1]
    cs = THIN_CS if (dt == torch.bfloat16 and c <= 3 and _thin_enabled()) else L.pad_channels(c, dt)
    return ToNHWC.apply(x, cs, dt)


def _relayout(t: torch.Tensor, c: int, cs: int) -> torch.Tensor:
    """NHWC tensor with another channel stride (a thin tensor meeting a layer the thin kernels do not cover)."""
    return _nchw_to_nhwc(_nhwc_to_nchw(t.contiguous(), c), cs, t.dtype)


def to_nchw(x: torch.Tensor, c: int) -> torch.Tensor:
    return ToNCHW.apply(x, c)


# ----------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------
class RepackTable:
    """Every packed weight image whose master weight belongs to ONE optimizer (it hangs on the epoch cell the fused
    optimizer shares with its parameters).  ``FlatAdam.step`` calls ``run()`` once: all images are re-packed by a
    single launch (``vg_conv_pack_batch``) instead of one ~7 us launch per layer on its next forward / backward.
    Entries hold weak references: a layer or parameter that is gone drops out (its buffers may have been freed)."""

    def __init__(self):
        self.entries = []          # [ref(PackedWeights), key, ref(weight), item bytes, dtype, weight data_ptr]
        self.tables = {}           # dtype -> (device table, n)
        self.dirty = True

    def add(self, owner, key, weight, desc, transposed, buf, dt):
        lib = L.load()
        item = (C.c_ubyte * lib.vg_conv_pack_item_bytes())()
        L.check(lib.vg_conv_pack_item(C.byref(desc), transposed, weight.data_ptr(), buf.data_ptr(), item), "pack_item")
        self.entries = [e for e in self.entries if not (e[0]() is owner and e[1] == key)]      # its old image is gone
        self.entries.append((weakref.ref(owner), key, weakref.ref(weight), bytes(item), dt, weight.data_ptr()))
        self.dirty = True

    def run(self):
        live = []
        for e in self.entries:
            owner, weight = e[0](), e[2]()
            if owner is None or weight is None or weight.data_ptr() != e[5] or e[1] not in owner.cache:
                self.dirty = True          # dropped: the pointers in its table entry may dangle
                continue
            live.append(e)
        self.entries = live
        if not live:
            return
        lib = L.load()
        if self.dirty:
            self.tables = {}
            dev = live[0][2]().device
            for dt in {e[4] for e in live}:
                raw = b"".join(e[3] for e in live if e[4] == dt)
                self.tables[dt] = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev), sum(1 for e in live if e[4] == dt))
            self.dirty = False
        for dt, (table, n) in self.tables.items():
            L.check(lib.vg_conv_pack_batch(table.data_ptr(), n, L.vg_dtype(dt), L.stream()), "pack_batch")
        for e in live:                     # the cached images are current again
            owner, weight = e[0](), e[2]()
            owner.cache[e[1]] = (PackedWeights.tag(weight), owner.cache[e[1]][1])


class EpochCell(list):
    """``[n]``: bumped by a fused optimizer whenever its parameters change; carries the optimizer's RepackTable."""

    def __init__(self, n):
        super().__init__([n])
        self.table = RepackTable()


def repack_all(epoch_cell) -> None:
    epoch_cell.table.run()


class PackedWeights:
    """Per-layer cache of the packed (and transposed-packed) weight images in the compute dtype."""

    def __init__(self):
        self.cache = {}

    @staticmethod
    def tag(weight):
        # a fused optimizer bumps the epoch cell it shares with ITS parameters (the D step does not invalidate the
        # generator's packed weights and vice versa); everything else shows up in torch's version counter
        return (weight._version, getattr(weight, "_vg_epoch", PARAM_EPOCH)[0], weight.data_ptr())

    def get(self, weight: torch.Tensor, desc: L.ConvDesc, transposed: int, dt: torch.dtype) -> torch.Tensor:
        key = (transposed, dt, desc.stride, desc.pad)
        tag = self.tag(weight)
        hit = self.cache.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1]
        lib = L.load()
        nbytes = lib.vg_conv_packed_bytes(C.byref(desc), transposed)
        if nbytes == 0:
            raise RuntimeError("conv: unsupported geometry for weight packing: " + L.last_error())
        fresh = hit is None or hit[1].numel() * hit[1].element_size() != nbytes or hit[0][2] != tag[2]
        buf = torch.empty(nbytes // (2 if dt == torch.bfloat16 else 4), dtype=dt, device=weight.device) if fresh else hit[1]
        L.check(lib.vg_conv_pack_weight(C.byref(desc), transposed, weight.data_ptr(), buf.data_ptr(), L.stream()), "pack_weight")
        self.cache[key] = (tag, buf)
        cell = getattr(weight, "_vg_epoch", None)
        if fresh and isinstance(cell, EpochCell):   # from now on the optimizer's step re-packs it with all the others
            cell.table.add(self, key, weight, desc, transposed, buf, dt)
        return buf


def _conv_desc(n, h, w, ci, co, k, stride, pad, xcs, ycs, dt) -> L.ConvDesc:
    oh = (h + 2 * pad - k) // stride + 1
    ow = (w + 2 * pad - k) // stride + 1
    return L.ConvDesc(n=n, h=h, w=w, oh=oh, ow=ow, ci=ci, co=co, xcs=xcs, ycs=ycs, r=k, s=k, stride=stride, pad=pad,
                      dtype=L.vg_dtype(dt))


class _ZeroArena:
    """Zeroed f64 scratch for one training step.  Every BatchNorm forward/backward and bias gradient needs a small
    zeroed f64 accumulator; as separate torch.zeros calls those are ~130 fill kernels per step.  Inside
    ``step_scope()`` they are slices of one buffer that is cleared by ONE fill at the start of the step."""

    def __init__(self):
        self.buf, self.off, self.want = None, 0, 1 << 20

    def begin(self, device):
        if self.buf is None or self.buf.numel() < self.want or self.buf.device != device:
            self.buf = torch.empty(self.want, dtype=torch.float64, device=device)
            self.buf.zero_()
        elif self.off:
            self.buf[:self.off].zero_()
        self.off = 0

    def take(self, n: int):
        n_al = (n + 31) // 32 * 32
        if self.buf is None or self.off + n_al > self.buf.numel():
            self.want = max(self.want, 2 * (self.off + n_al))     # grow at the next step
            return None
        out = self.buf[self.off:self.off + n]
        self.off += n_al
        return out


_ARENA = _ZeroArena()
_STEP = [False]     # inside step_scope(): arena slices instead of torch.zeros, gradients written straight into .grad


class step_scope:
    """Context of one optimizer phase of a training step (forward + backward).  Enables (a) the zero arena and
    (b) direct gradient accumulation: weight / bias / BatchNorm gradients are added into the parameter's existing
    ``.grad`` (FlatAdam's flat buffer, zeroed by ``zero_grad``) by the producing kernel instead of being returned
    to autograd, which would add them with one elementwise kernel per parameter."""

    def __init__(self, device):
        self.device = torch.device(device)

    def __enter__(self):
        _ARENA.begin(self.device)
        _STEP[0] = True
        return self

    def __exit__(self, *exc):
        _STEP[0] = False
        return False


def _zeros_f64(n: int, device) -> torch.Tensor:
    if _STEP[0]:
        t = _ARENA.take(n)
        if t is not None:
            return t
    return torch.zeros(n, dtype=torch.float64, device=device)


def _grad_slot(p) -> Optional[torch.Tensor]:
    """The tensor a kernel may accumulate ``p``'s gradient into, or None (plain autograd path)."""
    if not _STEP[0] or p is None:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


def _stats_buf(npad: int, device) -> torch.Tensor:
    return _zeros_f64(L.STATS_REPLICAS * 2 * npad, device)


def _act_bwd(g: torch.Tensor, y: torch.Tensor, act: int) -> torch.Tensor:
    out = torch.empty_like(y)
    L.check(L.load().vg_act_bwd(g.data_ptr(), y.data_ptr(), out.data_ptr(), y.numel(), act, L.vg_dtype(y.dtype), L.stream()), "act_bwd")
    return out


def _channel_sum(t: torch.Tensor, c: int, param=None) -> Optional[torch.Tensor]:
    """Per-channel sum (bias gradient).  Returns None when it was accumulated straight into ``param.grad``."""
    cs = t.shape[-1]
    rows = t.numel() // cs
    slot = _grad_slot(param)
    out = slot if slot is not None else torch.empty(c, dtype=torch.float32, device=t.device)
    ws = torch.empty(L.STATS_REPLICAS * cs, dtype=torch.float64, device=t.device)
    L.check(L.load().vg_channel_sum(t.data_ptr(), rows, cs, c, L.vg_dtype(t.dtype), out.data_ptr(), ws.data_ptr(),
                                    int(slot is not None), L.stream()), "channel_sum")
    return None if slot is not None else out


def _wgrad(desc: L.ConvDesc, x: torch.Tensor, dy: torch.Tensor, wshape, param=None) -> Optional[torch.Tensor]:
    """Weight gradient (OIHW f32).  Returns None when it was accumulated straight into ``param.grad``."""
    lib = L.load()
    ws = torch.empty(lib.vg_conv_wgrad_ws_bytes(C.byref(desc)) // 4, dtype=torch.float32, device=x.device)
    slot = _grad_slot(param)
    dw = slot if slot is not None else torch.empty(wshape, dtype=torch.float32, device=x.device)
    L.check(_timed("wgrad", _conv_flops(desc),
                   lambda: lib.vg_conv_wgrad(C.byref(desc), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                             int(slot is not None), L.stream()), _shape_tag("", desc)), "conv_wgrad")
    return None if slot is not None else dw


def _conv1_enabled() -> bool:
    """VG_NO_CONV1=1 sends the PatchGAN tail back to the gather-GEMM kernels (A/B)."""
    return not _os.environ.get("VG_NO_CONV1")


def _wgrad1(desc: L.ConvDesc, x: torch.Tensor, dy: torch.Tensor, wshape, param=None) -> Optional[torch.Tensor]:
    """Weight gradient of the one-output-channel 4x4 conv (csrc/tail.hip); same contract as ``_wgrad``."""
    lib = L.load()
    ws = torch.empty(lib.vg_conv1_wgrad_ws_bytes(C.byref(desc)) // 4, dtype=torch.float32, device=x.device)
    slot = _grad_slot(param)
    dw = slot if slot is not None else torch.empty(wshape, dtype=torch.float32, device=x.device)
    L.check(_timed("conv1", _conv_flops(desc),
                   lambda: lib.vg_conv1_wgrad(C.byref(desc), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                              int(slot is not None), L.stream()), _shape_tag("wgrad", desc)), "conv1_wgrad")
    return None if slot is not None else dw


class BiasLink:
    """Ties a conv to the BatchNorm that consumes its raw output.  The conv's bias gradient is the per-channel sum
    of the BatchNorm input gradient; ``vg_bn_bwd_apply`` takes that sum while it writes dx, straight into the bias
    ``.grad`` (inside ``step_scope``), and marks the link so that the conv's backward skips its own pass over dx."""
    __slots__ = ("bias", "done")

    def __init__(self, bias):
        self.bias, self.done = bias, False


class ResLink:
    """Ties the first conv of a ResidualBlock to the block's last BatchNorm: both send a gradient to the block
    input (conv path and identity skip).  The BatchNorm backward parks its skip gradient here instead of returning
    it, and the conv's input-gradient kernel accumulates onto it in its epilogue: no separate add pass."""
    __slots__ = ("dres",)

    def __init__(self):
        self.dres = None


class BwdStatLink:
    """Ties a training-mode BatchNorm to the ONE layer that consumes its output.  That layer's input-gradient launch
    writes the BatchNorm's upstream gradient, so it can take the BatchNorm backward sums (sum dz, sum dz*xhat) in
    its epilogue (``vg_conv_dgrad_bnstats`` / ``vg_conv_fwd_bnstats``); the BatchNorm backward then skips its own
    two-tensor reduction pass.  The link rides on the BatchNorm's output tensor (``_vg_bwdstat``) and is only used
    when a container that knows the topology armed it (FusedSequential: element k feeds element k+1 and nothing
    else; ResidualBlock: bn1 feeds conv2) and exactly one convolution claimed it."""
    __slots__ = ("x", "y", "vec", "c", "cpad", "act", "armed", "claims", "sums", "fused", "masked")

    def __init__(self):
        self.x = self.y = self.vec = self.sums = None
        self.c = self.cpad = self.act = 0
        self.armed, self.claims, self.fused, self.masked = False, 0, False, False


def arm_bwdstat(t) -> None:
    lk = getattr(t, "_vg_bwdstat", None)
    if lk is not None:
        lk.armed = True


def claim_bwdstat(t):
    """Called by a convolution on its input tensor: the producer BatchNorm's link, if it is armed."""
    lk = getattr(t, "_vg_bwdstat", None)
    if lk is None or not lk.armed:
        return None
    lk.claims += 1
    return lk


# Which BatchNorm-backward sums ride in the consumer's input-gradient epilogue.  bit 0: residual BatchNorms (the fused
# launch also stores the masked gradient: it replaces a pass that reads three tensors and writes one), bit 1: plain ones
# (replaces a two-tensor read).  Round 3, same box, 20-step pairs: none 38.24 ms, residual only 38.09, plain only 38.47,
# both 38.34: since the 16x16x32 main loops the epilogue's wait for the BatchNorm input costs more than the two-tensor pass.
_BNSTAT_FUSION = int(_os.environ.get("VG_BNSTAT_FUSION", "1"))


def _dgrad_with_bnstats(prod, final: bool, dtype) -> bool:
    if not (prod is not None and final and prod.claims == 1 and prod.x is not None and dtype == torch.bfloat16
            and prod.act in (L.ACT_NONE, L.ACT_LRELU) and not _os.environ.get("VG_NO_BNSTAT_FUSION")):
        return False
    return bool(_BNSTAT_FUSION & (1 if prod.y is not None else 2))


def _masked_bn_bwd() -> bool:
    """Residual BatchNorm backward on the masked gradient dz (written by the pass that takes the sums): the apply pass
    reads two tensors and writes one instead of three and two.  VG_NO_MASKED_BN_BWD=1 restores the old passes (A/B)."""
    return not _os.environ.get("VG_NO_MASKED_BN_BWD")


def _bn_bwd_stats_struct(prod, sums):
    vec = prod.vec
    # a residual BatchNorm (mask read from y): the fused launch stores the masked gradient, which is then also the skip
    # gradient of the block (BNActFn.backward)
    mask_out = int(prod.y is not None and prod.act == L.ACT_LRELU and _masked_bn_bwd())
    return L.BnBwdStats(x=prod.x.data_ptr(), y=L.ptr(prod.y), mean=vec[0].data_ptr(), invstd=vec[1].data_ptr(),
                        scale=vec[2].data_ptr(), shift=vec[3].data_ptr(), sums=sums.data_ptr(), cpad=prod.cpad, act=prod.act,
                        mask_out=mask_out)


def _bias_done(link) -> bool:
    if link is None or not link.done:
        return False
    link.done = False
    return True


class Conv2dFn(torch.autograd.Function):
    """y = act(conv2d(x, W) + b) on NHWC tensors; optionally emits BatchNorm statistics."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, want_stats, packs, link=None, reslink=None, prod=None):
        _require_cuda(x, "conv2d")
        ctx.set_materialize_grads(False)    # else autograd zero-fills a gradient for the (non-differentiable) statistics
        ctx.link, ctx.reslink, ctx.prod = link, reslink, prod
        x = x.contiguous()
        n, h, w, xcs = x.shape
        co, ci, k, _ = weight.shape
        dt = x.dtype
        ycs = L.pad_channels(co, dt)
        lib = L.load()
        d = _conv_desc(n, h, w, ci, co, k, stride, pad, xcs, ycs, dt)
        # thin layers (3-channel side on a 4-channel NHWC tensor, csrc/thin.hip)
        kind, ctx.thin_in_cs = 0, 0
        if dt == torch.bfloat16:
            if xcs == THIN_CS and ci <= 3:
                kind = lib.vg_thin_conv_supported(C.byref(d))
                if kind != 1:          # a geometry the thin kernels do not cover: back to the 32-padded layout
                    kind, ctx.thin_in_cs = 0, xcs
                    x = _relayout(x, ci, L.pad_channels(ci, dt))
                    xcs = x.shape[-1]
                    d = _conv_desc(n, h, w, ci, co, k, stride, pad, xcs, ycs, dt)
            elif co <= 3 and _thin_enabled():
                d4 = _conv_desc(n, h, w, ci, co, k, stride, pad, xcs, THIN_CS, dt)
                if lib.vg_thin_conv_supported(C.byref(d4)) == 2:
                    kind, d, ycs = 2, d4, THIN_CS
        y = torch.empty(n, d.oh, d.ow, ycs, dtype=dt, device=x.device)
        stats = _stats_buf(_npad(co), x.device) if want_stats else None
        if kind:
            w32 = weight.detach().contiguous()
            tws = torch.empty(lib.vg_thin_conv_ws_bytes(), dtype=torch.uint8, device=x.device)
            L.check(_timed("thin_conv", _conv_flops(d),
                           lambda: lib.vg_thin_conv_fwd(C.byref(d), x.data_ptr(), w32.data_ptr(), L.ptr(bias), y.data_ptr(),
                                                        L.ptr(stats), act, tws.data_ptr(), L.stream()), _shape_tag("fwd", d)),
                    "thin_conv_fwd")
        elif (co == 1 and not want_stats and dt == torch.bfloat16 and _conv1_enabled()
              and lib.vg_conv1_supported(C.byref(d))):
            # the PatchGAN tail (512 -> 1, 4x4 on 16x16): a dot product per pixel on the vector ALUs (csrc/tail.hip)
            kind = 3
            w32 = weight.detach().contiguous()
            L.check(_timed("conv1", _conv_flops(d),
                           lambda: lib.vg_conv1_fwd(C.byref(d), x.data_ptr(), w32.data_ptr(), L.ptr(bias), y.data_ptr(), act,
                                                    L.stream()), _shape_tag("fwd", d)), "conv1_fwd")
        elif (co <= 32 and not want_stats and dt == torch.bfloat16 and not deterministic()
              and n * d.oh * d.ow <= 32768 and k * k * xcs >= 2048):
            # few output tiles, a long reduction (the PatchGAN tail, 512 -> 1 4x4 on 16x16: 57 tiles of 256 K steps):
            # the K dimension split over blocks, partial sums meet in an f32 workspace (91 -> ~20 us)
            wp = packs.get(weight, d, 0, dt)
            ws = torch.empty(lib.vg_conv_splitk_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device=x.device)
            L.check(_timed(None, _conv_flops(d),
                           lambda: lib.vg_conv_fwd_splitk(C.byref(d), x.data_ptr(), wp.data_ptr(), L.ptr(bias), y.data_ptr(),
                                                          act, 0, ws.data_ptr(), L.stream()), _shape_tag("fwd", d)), "conv_fwd_splitk")
        else:
            wp = packs.get(weight, d, 0, dt)
            L.check(_timed(None, _conv_flops(d),
                           lambda: lib.vg_conv_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), L.ptr(bias), y.data_ptr(),
                                                   L.ptr(stats), act, 0, L.stream()), _shape_tag("fwd", d)), "conv_fwd")
        ctx.kind = kind
        ctx.d, ctx.act, ctx.packs, ctx.has_bias = d, act, packs, bias is not None
        ctx.wparam, ctx.bparam = weight, bias
        ctx.save_for_backward(x, weight, y if act != L.ACT_NONE else None)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, gy, _gstats):
        if gy is None:
            return (None,) * 11
        x, weight, y = ctx.saved_tensors
        d, lib = ctx.d, L.load()
        gy = gy.contiguous()
        if ctx.act != L.ACT_NONE:
            gy = _act_bwd(gy, y, ctx.act)
        dx = dw = db = None
        if ctx.kind in (1, 2):  # thin layer: both gradients straight from the master weights
            w32 = weight.detach().contiguous()
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                tws = torch.empty(lib.vg_thin_conv_ws_bytes(), dtype=torch.uint8, device=x.device)
                L.check(_timed("thin_conv", _conv_flops(d),
                               lambda: lib.vg_thin_conv_dgrad(C.byref(d), gy.data_ptr(), w32.data_ptr(), dx.data_ptr(),
                                                              tws.data_ptr(), L.stream()), _shape_tag("dgrad", d)), "thin_conv_dgrad")
            if ctx.needs_input_grad[1]:
                ws = torch.empty(lib.vg_thin_conv_wgrad_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device=x.device)
                slot = _grad_slot(ctx.wparam)
                dw = slot if slot is not None else torch.empty(weight.shape, dtype=torch.float32, device=x.device)
                L.check(_timed("thin_conv", _conv_flops(d),
                               lambda: lib.vg_thin_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                                              int(slot is not None), L.stream()), _shape_tag("wgrad", d)),
                        "thin_conv_wgrad")
                if slot is not None:
                    dw = None
            if ctx.has_bias and ctx.needs_input_grad[2] and not _bias_done(ctx.link):
                db = _channel_sum(gy, d.co, ctx.bparam)
            return dx, dw, db, None, None, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            wpt = ctx.packs.get(weight, d, 1, x.dtype)
            parked = ctx.reslink.dres if ctx.reslink is not None else None
            if parked is not None:          # skip-path gradient of the enclosing ResidualBlock: accumulate onto it
                ctx.reslink.dres = None
                dx = parked
            else:
                dx = torch.empty_like(x)
            prod = ctx.prod
            # dx is the complete gradient of x unless a ResidualBlock skip gradient is still on its way through autograd
            if _dgrad_with_bnstats(prod, ctx.reslink is None or parked is not None, x.dtype):
                sums = _zeros_f64(L.STATS_REPLICAS * 2 * prod.cpad + 2 * prod.cpad, x.device)
                bs, fused = _bn_bwd_stats_struct(prod, sums), C.c_int(0)
                L.check(_timed(None, _conv_flops(d),
                               lambda: lib.vg_conv_dgrad_bnstats(C.byref(d), gy.data_ptr(), wpt.data_ptr(), dx.data_ptr(),
                                                                 int(parked is not None), C.byref(bs), C.byref(fused), L.stream()),
                               _shape_tag("dgrad", d)), "conv_dgrad_bnstats")
                if fused.value:
                    prod.sums, prod.fused, prod.masked = sums, True, bool(bs.mask_out)
            else:
                L.check(_timed(None, _conv_flops(d),
                               lambda: lib.vg_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), None, dx.data_ptr(), None,
                                                         L.ACT_NONE, int(parked is not None), L.stream()),
                               _shape_tag("dgrad", d)), "conv_dgrad")
        if ctx.needs_input_grad[1]:
            dw = (_wgrad1 if ctx.kind == 3 else _wgrad)(d, x, gy, weight.shape, ctx.wparam)
        if ctx.has_bias and ctx.needs_input_grad[2] and not _bias_done(ctx.link):
            db = _channel_sum(gy, d.co, ctx.bparam)
        if dx is not None and ctx.thin_in_cs:      # the input arrived in the thin layout: hand its gradient back the same way
            dx = _relayout(dx, d.ci, ctx.thin_in_cs)
        return dx, dw, db, None, None, None, None, None, None, None, None


class ConvTranspose2dFn(torch.autograd.Function):
    """y = act(conv_transpose2d(x, W) + b): the dgrad kernel of the conv object W[O=in][I=out]."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, want_stats, packs, link=None, prod=None):
        _require_cuda(x, "conv_transpose2d")
        ctx.set_materialize_grads(False)
        ctx.link, ctx.prod = link, prod
        x = x.contiguous()
        n, h, w, xcs = x.shape
        cin_t, cout_t, k, _ = weight.shape
        dt = x.dtype
        oh, ow = (h - 1) * stride - 2 * pad + k, (w - 1) * stride - 2 * pad + k
        ycs = L.pad_channels(cout_t, dt)
        d = _conv_desc(n, oh, ow, cout_t, cin_t, k, stride, pad, ycs, xcs, dt)
        if d.oh != h or d.ow != w:
            raise RuntimeError("conv_transpose2d: unsupported geometry (output_padding != 0)")
        wpt = packs.get(weight, d, 1, dt)
        y = torch.empty(n, oh, ow, ycs, dtype=dt, device=x.device)
        stats = _stats_buf(_npad(cout_t), x.device) if want_stats else None
        lib = L.load()
        L.check(_timed(None, _conv_flops(d),
                       lambda: lib.vg_conv_dgrad(C.byref(d), x.data_ptr(), wpt.data_ptr(), L.ptr(bias), y.data_ptr(),
                                                 L.ptr(stats), act, 0, L.stream()), _shape_tag("convT", d)), "convT_fwd")
        ctx.d, ctx.act, ctx.packs, ctx.has_bias = d, act, packs, bias is not None
        ctx.wparam, ctx.bparam = weight, bias
        ctx.save_for_backward(x, weight, y if act != L.ACT_NONE else None)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, gy, _gstats):
        if gy is None:
            return (None,) * 10
        x, weight, y = ctx.saved_tensors
        d, lib = ctx.d, L.load()
        gy = gy.contiguous()
        if ctx.act != L.ACT_NONE:
            gy = _act_bwd(gy, y, ctx.act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wp = ctx.packs.get(weight, d, 0, x.dtype)
            dx = torch.empty_like(x)
            prod = ctx.prod
            if _dgrad_with_bnstats(prod, True, x.dtype):
                sums = _zeros_f64(L.STATS_REPLICAS * 2 * prod.cpad + 2 * prod.cpad, x.device)
                bs, fused = _bn_bwd_stats_struct(prod, sums), C.c_int(0)
                L.check(_timed(None, _conv_flops(d),
                               lambda: lib.vg_conv_fwd_bnstats(C.byref(d), gy.data_ptr(), wp.data_ptr(), dx.data_ptr(), 0,
                                                               C.byref(bs), C.byref(fused), L.stream()),
                               _shape_tag("convT_dgrad", d)), "convT_dgrad_bnstats")
                if fused.value:
                    prod.sums, prod.fused, prod.masked = sums, True, bool(bs.mask_out)
            else:
                L.check(_timed(None, _conv_flops(d),
                               lambda: lib.vg_conv_fwd(C.byref(d), gy.data_ptr(), wp.data_ptr(), None, dx.data_ptr(), None,
                                                       L.ACT_NONE, 0, L.stream()), _shape_tag("convT_dgrad", d)), "convT_dgrad")
        if ctx.needs_input_grad[1]:
            dw = _wgrad(d, gy, x, weight.shape, ctx.wparam)
        if ctx.has_bias and ctx.needs_input_grad[2] and not _bias_done(ctx.link):
            db = _channel_sum(gy, d.ci, ctx.bparam)
        return dx, dw, db, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# BatchNorm (+ residual) + activation
# ----------------------------------------------------------------------------------------------
def channel_stats(x: torch.Tensor, c: int) -> torch.Tensor:
    cs = x.shape[-1]
    rows = x.numel() // cs
    npad = _npad(c)
    stats = _stats_buf(npad, x.device)
    L.check(L.load().vg_channel_stats(x.data_ptr(), stats.data_ptr(), rows, cs, c, npad, L.vg_dtype(x.dtype), L.stream()), "channel_stats")
    return stats


# Data parallel with global-batch statistics (train.DataParallel(bn_sync='global')): an object with ``world`` and
# ``sum_(tensor)`` (in-place sum over the ranks).  The reference is single-device, so its BatchNorm batch statistics
# and its VQ-EMA counts run over the WHOLE batch; with this set, every rank normalises with the statistics of the
# global batch (forward sums, backward sums and EMA statistics are summed over the ranks before they are used).
STAT_SYNC = [None]

BN_UPDATES = [1]    # running-statistics updates one training-mode BatchNorm forward stands for (see bn_updates)


class bn_updates:
    """Inside this context one BatchNorm forward applies its running-stat update ``n`` times.  The VQ and
    hierarchical iterations run the encoder twice on the same images with the same weights
    (vq_vae_gan/main.py:78 and :99; only the discriminator changed in between): train-mode BatchNorm makes the
    two passes bit-identical, so the step executes one and books the buffer updates of both."""

    def __init__(self, n: int):
        self.n = int(n)

    def __enter__(self):
        self.prev, BN_UPDATES[0] = BN_UPDATES[0], self.n
        return self

    def __exit__(self, *exc):
        BN_UPDATES[0] = self.prev
        return False


class BNActFn(torch.autograd.Function):
    """y = act(batchnorm_train(x) (+ res)); statistics come from the producing conv's epilogue."""

    @staticmethod
    def forward(ctx, x, stats, gamma, beta, res, running_mean, running_var, nbt, eps, momentum, act, c, link=None, reslink=None,
                bslink=None):
        lib = L.load()
        x = x.contiguous()
        cs = x.shape[-1]
        rows = x.numel() // cs
        cpad = stats.numel() // (L.STATS_REPLICAS * 2)
        vec = torch.empty(4, c, dtype=torch.float32, device=x.device)   # mean, invstd, scale, shift
        sync = STAT_SYNC[0]
        count = rows
        if sync is not None:
            sync.sum_stats(stats, 2 * cpad)
            count = rows * sync.world
        ctx.count, ctx.sync = count, sync
        L.check(lib.vg_bn_finalize(stats.data_ptr(), c, cpad, count, gamma.data_ptr(), beta.data_ptr(), eps, momentum,
                                   L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt), BN_UPDATES[0], vec[0].data_ptr(),
                                   vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), L.stream()), "bn_finalize")
        y = torch.empty_like(x)
        if res is not None:
            res = res.contiguous()
        L.check(lib.vg_bn_act_fwd(x.data_ptr(), L.ptr(res), y.data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), rows, cs, c,
                                  act, L.vg_dtype(x.dtype), L.stream()), "bn_act_fwd")
        ctx.c, ctx.cpad, ctx.act, ctx.has_res = c, cpad, act, res is not None
        ctx.gparam, ctx.bparam, ctx.link, ctx.reslink = gamma, beta, link, reslink
        ctx.bslink = bslink
        if bslink is not None:      # what the consumer's input-gradient launch needs to take this layer's backward sums
            bslink.x, bslink.vec, bslink.c, bslink.cpad, bslink.act = x, vec, c, cpad, act
            bslink.y = y if (res is not None or act not in (L.ACT_NONE, L.ACT_LRELU)) else None
        ctx.save_for_backward(x, y, vec, gamma)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, vec, gamma = ctx.saved_tensors
        lib = L.load()
        gy = gy.contiguous()
        cs = x.shape[-1]
        rows = x.numel() // cs
        c, cpad = ctx.c, ctx.cpad
        dtc = L.vg_dtype(x.dtype)
        # without a residual the LeakyReLU mask is recomputed from x (one tensor read less per pass)
        yp = None if (not ctx.has_res and ctx.act in (L.ACT_NONE, L.ACT_LRELU)) else y.data_ptr()
        bsl = ctx.bslink
        act, dres, dres_out = ctx.act, None, None
        if bsl is not None and bsl.fused:       # the consumer's input-gradient launch already took the sums (BwdStatLink)
            sums = bsl.sums
            if bsl.masked:                      # ... and stored the masked gradient: gy is dz, and the skip gradient
                act, yp = L.ACT_NONE, None
                dres = gy if ctx.has_res else None
        else:
            sums = _zeros_f64(L.STATS_REPLICAS * 2 * cpad + 2 * cpad, x.device)
            if ctx.has_res and ctx.act == L.ACT_LRELU and _masked_bn_bwd():
                # the reduction writes dz = gy * mask (the skip gradient); the apply pass then reads (dz, x) only
                dres = torch.empty_like(x)
                L.check(lib.vg_bn_bwd_reduce_dz(gy.data_ptr(), yp, x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(),
                                                vec[2].data_ptr(), vec[3].data_ptr(), sums.data_ptr(), dres.data_ptr(), rows,
                                                cs, c, cpad, ctx.act, dtc, L.stream()), "bn_bwd_reduce_dz")
                gy, act, yp = dres, L.ACT_NONE, None
            else:
                L.check(lib.vg_bn_bwd_reduce(gy.data_ptr(), yp, x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(),
                                             vec[2].data_ptr(), vec[3].data_ptr(), sums.data_ptr(), rows, cs, c, cpad, ctx.act, dtc,
                                             L.stream()), "bn_bwd_reduce")
        if bsl is not None:
            bsl.x = bsl.y = bsl.vec = bsl.sums = None
            bsl.fused = bsl.masked = False
        sync = ctx.sync
        if sync is not None:
            sync.sum_stats(sums[:L.STATS_REPLICAS * 2 * cpad], 2 * cpad)
        dx = torch.empty_like(x)
        if ctx.has_res and dres is None:
            dres = dres_out = torch.empty_like(x)
        gslot, bslot = _grad_slot(ctx.gparam), _grad_slot(ctx.bparam)
        direct = gslot is not None and bslot is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]
        dgb = (gslot, bslot) if direct else torch.empty(2, c, dtype=torch.float32, device=x.device)
        # bias gradient of the producing conv = per-channel sum of dx: taken by the same kernel (see BiasLink)
        link = ctx.link
        cbslot = _grad_slot(link.bias) if (link is not None and link.bias is not None and link.bias.requires_grad) else None
        L.check(lib.vg_bn_bwd_apply(gy.data_ptr(), yp, x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(),
                                    vec[2].data_ptr(), vec[3].data_ptr(), gamma.data_ptr(), sums.data_ptr(), dx.data_ptr(),
                                    L.ptr(dres_out), dgb[0].data_ptr(),
                                    dgb[1].data_ptr(), int(direct), L.ptr(cbslot), rows, ctx.count,
                                    1.0 if sync is None else 1.0 / sync.world, cs, c, cpad, act, dtc,
                                    L.stream()), "bn_bwd_apply")
        if cbslot is not None:
            link.done = True
        if dres is not None and ctx.reslink is not None and _STEP[0] and ctx.needs_input_grad[4]:
            ctx.reslink.dres, dres = dres, None      # picked up by the block's first conv (see ResLink)
        if direct:
            return dx, None, None, None, dres, None, None, None, None, None, None, None, None, None, None
        return dx, None, dgb[0], dgb[1], dres, None, None, None, None, None, None, None, None, None, None


def bn_eval_act(x: torch.Tensor, gamma, beta, running_mean, running_var, eps: float, act: int, c: int,
                res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inference-mode BatchNorm (+res) + activation: per-channel affine from the running statistics.
    (The [C]-sized scale/shift vectors are formed with torch ops; the tensor pass is the HIP kernel.)"""
    scale = (gamma * torch.rsqrt(running_var + eps)).float().contiguous()
    shift = (beta - running_mean * scale).float().contiguous()
    y = torch.empty_like(x)
    cs = x.shape[-1]
    L.check(L.load().vg_bn_act_fwd(x.data_ptr(), L.ptr(res), y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                   x.numel() // cs, cs, c, act, L.vg_dtype(x.dtype), L.stream()), "bn_act_fwd")
    return y


class GradBoundaryFn(torch.autograd.Function):
    """Identity whose backward calls ``fn()`` first: placed on the tensor that enters a group of layers, it fires
    when the backward pass has enqueued every gradient kernel of that group (data-parallel bucket boundary)."""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.fn()
        return g, None


def grad_boundary(x: torch.Tensor, fn) -> torch.Tensor:
    return GradBoundaryFn.apply(x, fn) if (torch.is_grad_enabled() and x.requires_grad) else x


class FanOutFn(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers; the backward adds the two gradients with the HIP add kernel
    (left to autograd, the sum would be one of torch's elementwise kernels)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        out = ga.contiguous().clone()
        L.check(L.load().vg_add_inplace(out.data_ptr(), gb.contiguous().data_ptr(), out.numel(), L.vg_dtype(out.dtype), L.stream()), "add")
        return out


def fan_out(x: torch.Tensor):
    return FanOutFn.apply(x) if (torch.is_grad_enabled() and x.requires_grad) else (x, x)


class AddFn(torch.autograd.Function):
    """a + b on NHWC activations (hand-written kernel; used where a block output fans in)."""

    @staticmethod
    def forward(ctx, a, b):
        out = a.contiguous().clone()
        L.check(L.load().vg_add_inplace(out.data_ptr(), b.contiguous().data_ptr(), out.numel(), L.vg_dtype(out.dtype), L.stream()), "add")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


# ----------------------------------------------------------------------------------------------
# vector quantiser
# ----------------------------------------------------------------------------------------------
VQ_SHORTLIST_MIN_K = 2048     # below this the exact-f32 search is tens of microseconds


class VectorQuantizeFn(torch.autograd.Function):
    """vq_vae_gan/model.py:68-134 on NCHW f32 input: indices, straight-through output, VQ loss and the
    in-place EMA codebook update (when ``training``)."""

    @staticmethod
    def forward(ctx, z, codebook, ema_cluster, ema_w, commitment, decay, training):
        _require_cuda(z, "vector_quantizer")
        lib = L.load()
        z = z.contiguous().float()
        b, d, h, w = z.shape
        k = codebook.shape[0]
        rows = b * h * w
        dev = z.device
        flat = torch.empty(rows, d, dtype=torch.float32, device=dev)
        L.check(lib.vg_nchw_to_nhwc(z.data_ptr(), flat.data_ptr(), b, d, h, w, d, L.VG_F32, L.stream()), "vq.flatten")
        idx = torch.empty(rows, dtype=torch.int64, device=dev)
        if k >= VQ_SHORTLIST_MIN_K and d in (64, 128, 256) and not _os.environ.get("VG_NO_VQ_SHORTLIST"):
            # large codebooks: candidates from the bf16 MFMA, exact-f32 distances for those only (same indices)
            ws = torch.empty(lib.vg_vq_shortlist_ws_bytes(rows, k, d), dtype=torch.uint8, device=dev)
            L.check(lib.vg_vq_assign_shortlist(flat.data_ptr(), codebook.data_ptr(), rows, k, d, idx.data_ptr(), ws.data_ptr(),
                                               None, L.stream()), "vq_assign_shortlist")
        else:
            norms = torch.empty(k, dtype=torch.float32, device=dev)
            L.check(lib.vg_vq_assign(flat.data_ptr(), codebook.data_ptr(), rows, k, d, idx.data_ptr(), norms.data_ptr(),
                                     L.stream()), "vq_assign")
        zq_flat = torch.empty_like(flat)
        scal = _zeros_f64(2, dev)       # [sse, unused]
        counts = esum = None
        if training:
            buf = torch.zeros(k * (d + 1) + 1, dtype=torch.float32, device=dev)
            counts, esum, nsum = buf[:k], buf[k:k + k * d], buf[k + k * d:]
        L.check(lib.vg_vq_gather(flat.data_ptr(), codebook.data_ptr(), idx.data_ptr(), rows, k, d, zq_flat.data_ptr(),
                                 scal.data_ptr(), L.ptr(counts), L.ptr(esum), L.stream()), "vq_gather")
        if training:
            if STAT_SYNC[0] is not None:
                STAT_SYNC[0].sum_(buf[:k * (d + 1)])        # counts and embedded sums of the global batch
            L.check(lib.vg_vq_ema_update(ema_cluster.data_ptr(), ema_w.data_ptr(), codebook.data_ptr(), counts.data_ptr(),
                                         esum.data_ptr(), k, d, decay, nsum.data_ptr(), L.stream()), "vq_ema_update")
        zq = torch.empty_like(z)
        L.check(lib.vg_nhwc_to_nchw(zq_flat.data_ptr(), zq.data_ptr(), b, d, h, w, d, L.VG_F32, L.stream()), "vq.unflatten")
        loss = torch.empty((), dtype=torch.float32, device=dev)
        L.check(lib.vg_scalar_finish(scal.data_ptr(), (1.0 + commitment) / z.numel(), loss.data_ptr(), L.stream()), "vq.loss")
        ctx.coef = 2.0 * commitment / z.numel()
        ctx.save_for_backward(z, zq)
        idx = idx.view(b, h, w)
        ctx.mark_non_differentiable(idx)
        return zq, loss, idx

    @staticmethod
    def backward(ctx, g_zq, g_loss, _gidx):
        z, zq = ctx.saved_tensors
        dz = torch.empty_like(z)
        g_zq = g_zq.contiguous() if g_zq is not None else None
        g_loss = g_loss.contiguous().float() if g_loss is not None else None
        L.check(L.load().vg_vq_bwd(L.ptr(g_zq), L.ptr(g_loss), z.data_ptr(), zq.data_ptr(), z.numel(), ctx.coef,
                                   dz.data_ptr(), L.stream()), "vq_bwd")
        return dz, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# loss reductions (0-dim f32 device tensors; no host sync)
# ----------------------------------------------------------------------------------------------
def _finish(acc: torch.Tensor, scale: float) -> torch.Tensor:
    out = torch.empty((), dtype=torch.float32, device=acc.device)
    L.check(L.load().vg_scalar_finish(acc.data_ptr(), scale, out.data_ptr(), L.stream()), "scalar_finish")
    return out


class MSEFn(torch.autograd.Function):
    """mean((a-b)^2) over two same-shaped f32 tensors (F.mse_loss, <arch>/model.py loss_function)."""

    @staticmethod
    def forward(ctx, a, b):
        _require_cuda(a, "mse")
        if a.shape != b.shape:
            # same failure mode as the reference (beta_vae_gans/model.py:255 raises on 128 vs 256)
            raise RuntimeError(f"The size of tensor a {tuple(a.shape)} must match the size of tensor b {tuple(b.shape)}")
        a, b = a.contiguous().float(), b.contiguous().float()
        n = a.numel()
        if n % 4:
            raise RuntimeError("mse: element count must be a multiple of 4")
        acc = _zeros_f64(1, a.device)
        L.check(L.load().vg_sqdiff_sum(a.data_ptr(), b.data_ptr(), n // 4, 4, 4, L.VG_F32, acc.data_ptr(), L.stream()), "sqdiff_sum")
        ctx.save_for_backward(a, b)
        return _finish(acc, 1.0 / n)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        n = a.numel()
        g = g.contiguous().float()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            L.check(L.load().vg_sqdiff_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), 2.0 / n, n // 4, 4, 4, L.VG_F32,
                                           da.data_ptr(), L.stream()), "sqdiff_bwd")
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(b)
            L.check(L.load().vg_sqdiff_bwd(b.data_ptr(), a.data_ptr(), g.data_ptr(), 2.0 / n, n // 4, 4, 4, L.VG_F32,
                                           db.data_ptr(), L.stream()), "sqdiff_bwd")
        return da, db


class MeanFn(torch.autograd.Function):
    """scale * mean(f(x)) with f = identity (mode 0), relu(1-x) (1) or relu(1+x) (2)."""

    @staticmethod
    def forward(ctx, x, mode, scale):
        _require_cuda(x, "mean")
        x = x.contiguous().float()
        acc = _zeros_f64(1, x.device)
        L.check(L.load().vg_reduce_f32(x.data_ptr(), x.numel(), mode, acc.data_ptr(), L.stream()), "reduce_f32")
        ctx.mode, ctx.scale = mode, scale
        ctx.save_for_backward(x)
        return _finish(acc, scale / x.numel())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        g = g.contiguous().float()
        L.check(L.load().vg_reduce_f32_bwd(x.data_ptr(), g.data_ptr(), ctx.scale / x.numel(), x.numel(), ctx.mode,
                                           dx.data_ptr(), L.stream()), "reduce_f32_bwd")
        return dx, None, None


def mse(a, b):
    return MSEFn.apply(a, b)


def hinge_d(disc_real, disc_fake):
    """mean(relu(1 - real)) + mean(relu(1 + fake))   (<arch>/model.py loss_function, D branch)."""
    return MeanFn.apply(disc_real, 1, 1.0) + MeanFn.apply(disc_fake, 2, 1.0)


def neg_mean(x):
    return MeanFn.apply(x, 0, -1.0)


class ReparamFn(torch.autograd.Function):
    """z = mu + eps * exp(0.5 * logvar) (beta_vae_gans/model.py:211-216)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        _require_cuda(mu, "reparameterize")
        mu, logvar, eps = mu.contiguous().float(), logvar.contiguous().float(), eps.contiguous().float()
        z = torch.empty_like(mu)
        L.check(L.load().vg_reparam_kl_fwd(mu.data_ptr(), logvar.data_ptr(), eps.data_ptr(), mu.numel(), z.data_ptr(), None,
                                           L.stream()), "reparam")
        ctx.save_for_backward(mu, logvar, eps)
        return z

    @staticmethod
    def backward(ctx, gz):
        mu, logvar, eps = ctx.saved_tensors
        dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
        L.check(L.load().vg_reparam_kl_bwd(mu.data_ptr(), logvar.data_ptr(), eps.data_ptr(), gz.contiguous().data_ptr(), None,
                                           0.0, mu.numel(), dmu.data_ptr(), dlv.data_ptr(), L.stream()), "reparam_bwd")
        return dmu, dlv, None


class KLFn(torch.autograd.Function):
    """-0.5 * mean(1 + logvar - mu^2 - exp(logvar)) (beta_vae_gans/model.py:258)."""

    @staticmethod
    def forward(ctx, mu, logvar):
        _require_cuda(mu, "kl")
        mu, logvar = mu.contiguous().float(), logvar.contiguous().float()
        acc = _zeros_f64(1, mu.device)
        L.check(L.load().vg_reparam_kl_fwd(mu.data_ptr(), logvar.data_ptr(), None, mu.numel(), None, acc.data_ptr(),
                                           L.stream()), "kl")
        ctx.save_for_backward(mu, logvar)
        return _finish(acc, -0.5 / mu.numel())

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
        g = g.contiguous().float()
        L.check(L.load().vg_reparam_kl_bwd(mu.data_ptr(), logvar.data_ptr(), None, None, g.data_ptr(), -0.5 / mu.numel(),
                                           mu.numel(), dmu.data_ptr(), dlv.data_ptr(), L.stream()), "kl_bwd")
        return dmu, dlv


def reparameterize(mu, logvar, eps):
    return ReparamFn.apply(mu, logvar, eps)


def kl_divergence(mu, logvar):
    return KLFn.apply(mu, logvar)


# ----------------------------------------------------------------------------------------------
# Linear (beta_vae_gans/model.py:74-75,107): a 1x1 conv over an NHWC [B,1,1,F] f32 tensor
# ----------------------------------------------------------------------------------------------
def _lin_desc(rows, cin, cout, xcs=None, ycs=None):
    dt = torch.float32
    return L.ConvDesc(n=rows, h=1, w=1, oh=1, ow=1, ci=cin, co=cout, xcs=xcs or L.pad_channels(cin, dt),
                      ycs=ycs or L.pad_channels(cout, dt), r=1, s=1, stride=1, pad=0, dtype=L.VG_F32)


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b in f32 on the exact-f32 MFMA.  The row-major [out][in] master weight is its own packed
    image, so nothing is repacked: forward = split-K igemm (weight streamed once), input gradient and weight
    gradient = the pixel-reduction (wgrad) kernel with the roles of the operands chosen so that both are read
    in their natural row-major layout."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_cuda(x, "linear")
        lib = L.load()
        x = x.contiguous().float()
        b, fin = x.shape
        fout = weight.shape[0]
        if fin % 16 or fout % 16:
            raise RuntimeError("linear: feature counts must be multiples of 16")
        d = _lin_desc(b, fin, fout)
        y = torch.empty(b, fout, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.vg_conv_splitk_ws_bytes(C.byref(d)) // 4, dtype=torch.float32, device=x.device)
        L.check(lib.vg_conv_fwd_splitk(C.byref(d), x.data_ptr(), weight.data_ptr(), L.ptr(bias), y.data_ptr(), L.ACT_NONE, 0,
                                       ws.data_ptr(), L.stream()), "linear_fwd")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.wparam = weight
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        lib = L.load()
        gy = gy.contiguous().float()
        b, fin = x.shape
        fout = weight.shape[0]
        bp = L.pad_channels(b, torch.float32)
        dx = dw = db = None
        # wide layers (fc_mu / fc_logvar of the beta model: 524288 -> 128): one pass over the [fout][fin] matrix each
        wide = fin >= 4096 and fin % 2 == 0 and fout * 128 <= 160 * 1024
        if wide:
            if ctx.needs_input_grad[0]:
                dx = torch.empty(b, fin, dtype=torch.float32, device=x.device)
                L.check(lib.vg_linear_dgrad(gy.data_ptr(), weight.data_ptr(), dx.data_ptr(), b, fin, fout, L.stream()), "linear_dgrad")
            if ctx.needs_input_grad[1]:
                slot = _grad_slot(ctx.wparam)
                dwt = slot if slot is not None else torch.empty(fout, fin, dtype=torch.float32, device=x.device)
                L.check(_timed("linear_wgrad", 2.0 * b * fin * fout,
                               lambda: lib.vg_linear_wgrad(x.data_ptr(), gy.data_ptr(), dwt.data_ptr(), b, fin, fout,
                                                           int(slot is not None), L.stream()), ""), "linear_wgrad")
                dw = None if slot is not None else dwt
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = torch.empty(fout, dtype=torch.float32, device=x.device)
                L.check(lib.vg_colsum_f32(gy.data_ptr(), b, fout, db.data_ptr(), L.stream()), "linear.dbias")
            return dx, dw, db
        if ctx.needs_input_grad[0]:
            # dx[b][k] = sum_o gy[b][o] W[o][k]: "pixels" = o; dy := gy^T [o][bp], x := W [o][fin]
            gyt = torch.empty(fout, bp, dtype=torch.float32, device=x.device)
            L.check(lib.vg_nchw_to_nhwc(gy.data_ptr(), gyt.data_ptr(), 1, b, 1, fout, bp, L.VG_F32, L.stream()), "linear.gyT")
            d2 = L.ConvDesc(n=1, h=1, w=fout, oh=1, ow=fout, ci=fin, co=b, xcs=fin, ycs=bp, r=1, s=1, stride=1, pad=0,
                            dtype=L.VG_F32)
            dx = _wgrad(d2, weight, gyt, (b, fin))
        if ctx.needs_input_grad[1]:
            # dW[o][k] = sum_b gy[b][o] x[b][k]: "pixels" = b
            d = _lin_desc(b, fin, fout)
            dw = _wgrad(d, x, gy, (fout, fin), ctx.wparam)     # inside step_scope: straight into the flat .grad buffer
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(fout, dtype=torch.float32, device=x.device)
            L.check(lib.vg_colsum_f32(gy.data_ptr(), b, fout, db.data_ptr(), L.stream()), "linear.dbias")
        return dx, dw, db


def linear(x, weight, bias):
    return LinearFn.apply(x, weight, bias)


class ClampFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo, hi):
        _require_cuda(x, "clamp")
        x = x.contiguous().float()
        y = torch.empty_like(x)
        L.check(L.load().vg_clamp(x.data_ptr(), None, lo, hi, y.data_ptr(), x.numel(), L.stream()), "clamp")
        ctx.save_for_backward(x)
        ctx.lo, ctx.hi = lo, hi
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        L.check(L.load().vg_clamp(x.data_ptr(), g.contiguous().float().data_ptr(), ctx.lo, ctx.hi, dx.data_ptr(), x.numel(),
                                  L.stream()), "clamp_bwd")
        return dx, None, None


def clamp(x, lo, hi):
    return ClampFn.apply(x, float(lo), float(hi))


def avg_pool2(x: torch.Tensor) -> torch.Tensor:
    """2x2 area downsample of an NCHW f32 tensor (no gradient: used on the target image only)."""
    _require_cuda(x, "avg_pool2")
    x = x.detach().contiguous().float()
    n, c, h, w = x.shape
    out = torch.empty(n, c, h // 2, w // 2, dtype=torch.float32, device=x.device)
    L.check(L.load().vg_avgpool2_nchw(x.data_ptr(), out.data_ptr(), n * c, h, w, L.stream()), "avgpool2")
    return out


# ----------------------------------------------------------------------------------------------
# self attention (hierarchical_vae_gan/model.py:24-47)
# ----------------------------------------------------------------------------------------------
def _to_f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype == torch.float32:
        return t.contiguous()
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    L.check(L.load().vg_cast(t.data_ptr(), out.data_ptr(), t.numel(), L.VG_BF16, L.VG_F32, L.stream()), "cast")
    return out


def _from_f32(t: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    if dt == torch.float32:
        return t
    out = torch.empty(t.shape, dtype=dt, device=t.device)
    L.check(L.load().vg_cast(t.contiguous().data_ptr(), out.data_ptr(), t.numel(), L.VG_F32, L.VG_BF16, L.stream()), "cast")
    return out


class AttentionFn(torch.autograd.Function):
    """out = gamma * softmax(q k^T) v + x on NHWC tensors [B,H,W,*] (q,k: [..,dqk_s], v,x: [..,cs]).
    bf16 mode: the products run on the bf16 MFMA straight from the bf16 tensors (vg_attn_*_bf16) where the shape is
    covered; otherwise, and always in f32 mode (or with VG_ATTN_F32=1 in the environment), on the exact-f32 MFMA."""

    @staticmethod
    def forward(ctx, q, k, v, x, gamma):
        _require_cuda(q, "attention")
        lib = L.load()
        dt = x.dtype
        b, hh, ww, cs = v.shape
        hw, d = hh * ww, q.shape[-1]
        fast = dt == torch.bfloat16 and bool(lib.vg_attn_supported_bf16(hw, d, cs)) and not _os.environ.get("VG_ATTN_F32")
        xf = _to_f32(x)
        o = torch.empty(b, hh, ww, cs, dtype=torch.float32, device=q.device)
        lse = torch.empty(b * hw, dtype=torch.float32, device=q.device)
        if fast:
            qf, kf, vf = q.contiguous(), k.contiguous(), v.contiguous()
            L.check(lib.vg_attn_fwd_bf16(qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), o.data_ptr(), lse.data_ptr(), b, hw, d, cs,
                                         L.stream()), "attn_fwd_bf16")
        else:
            qf, kf, vf = _to_f32(q), _to_f32(k), _to_f32(v)
            L.check(lib.vg_attn_fwd(qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), o.data_ptr(), lse.data_ptr(), b, hw, d, cs,
                                    L.stream()), "attn_fwd")
        y = torch.empty_like(o)
        gam = gamma.detach().float().contiguous()
        L.check(lib.vg_gamma_residual(o.data_ptr(), xf.data_ptr(), gam.data_ptr(), y.data_ptr(), y.numel(), L.stream()), "gamma_residual")
        ctx.save_for_backward(qf, kf, vf, o, lse, gam)
        ctx.dt, ctx.dims, ctx.fast = dt, (b, hw, d, cs), fast
        ctx.in_dts = (q.dtype, k.dtype, v.dtype)
        return _from_f32(y, dt)

    @staticmethod
    def backward(ctx, gy):
        qf, kf, vf, o, lse, gam = ctx.saved_tensors
        lib = L.load()
        b, hw, d, cs = ctx.dims
        gyf = _to_f32(gy)
        d_o = torch.empty_like(o)
        dgam = torch.zeros(1, dtype=torch.float64, device=o.device)
        L.check(lib.vg_gamma_residual_bwd(gyf.data_ptr(), o.data_ptr(), gam.data_ptr(), d_o.data_ptr(), dgam.data_ptr(),
                                          o.numel(), L.stream()), "gamma_residual_bwd")
        dq = torch.empty(qf.shape, dtype=torch.float32, device=o.device)
        dk = torch.empty(kf.shape, dtype=torch.float32, device=o.device)
        dv = torch.empty(vf.shape, dtype=torch.float32, device=o.device)
        ws = torch.empty(b * hw, dtype=torch.float32, device=o.device)
        fn = lib.vg_attn_bwd_bf16 if ctx.fast else lib.vg_attn_bwd
        L.check(fn(qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(),
                   dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), b, hw, d, cs, L.stream()), "attn_bwd")
        dgamma = torch.empty(1, dtype=torch.float32, device=o.device)
        L.check(lib.vg_scalar_finish(dgam.data_ptr(), 1.0, dgamma.data_ptr(), L.stream()), "dgamma")
        dt = ctx.dt
        qd, kd, vd = ctx.in_dts
        return _from_f32(dq, qd), _from_f32(dk, kd), _from_f32(dv, vd), _from_f32(gyf, dt) if dt != torch.float32 else gyf, dgamma


def attention(q, k, v, x, gamma):
    return AttentionFn.apply(q, k, v, x, gamma)
