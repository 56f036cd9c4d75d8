"""Thin Python wrappers over the C ABI (include/siu3r_hip.h): torch tensors in, raw device
pointers + the current HIP stream out.  PyTorch provides memory and streams only; every
arithmetic operation below runs in libsiu3r_hip.so.  No CPU fallback: non-GPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import BF16, F32, AttnParams, GemmParams, check

ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise RuntimeError(f"unsupported dtype {t.dtype} (float32 or bfloat16 expected)")


def torch_dtype(code: int):
    return torch.float32 if code == F32 else torch.bfloat16


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("siu3r_amd ops need GPU tensors: the HIP path has no CPU fallback")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# weight packing
# ------------------------------------------------------------------------------------------------
@dataclass
class PackedWeight:
    hi: torch.Tensor                 # bf16 [N, Kpad]
    lo: Optional[torch.Tensor]       # bf16 [N, Kpad] (bf16x3 mode) or None
    bias: Optional[torch.Tensor] = None  # fp32
    x3: Optional[torch.Tensor] = None    # bf16 [N, Kpad/32, 2, 32]: hi | lo interleaved per 32-deep K tile (bf16x3 on the LDS-DMA path)
    n: int = 0
    k: int = 0
    kpad: int = 0
    meta: dict = None


def split_bf16(x2d: torch.Tensor, want_lo: bool, kpad: Optional[int] = None, want_x3: bool = False):
    """fp32 [rows, k] (last dim contiguous) -> bf16 hi [rows, kpad] (+ lo) (+ x3 [rows, kpad/32, 2, 32]: the two planes interleaved per
    32-deep K tile, the W layout of the bf16x3 LDS-DMA GEMM; returned as a fourth value)."""
    _gpu(x2d)
    assert x2d.dtype == torch.float32 and x2d.dim() == 2 and x2d.stride(1) == 1
    rows, k = x2d.shape
    kpad = kpad or ((k + 63) // 64) * 64
    hi = torch.empty((rows, kpad), dtype=torch.bfloat16, device=x2d.device)
    lo = torch.empty_like(hi) if want_lo else None
    x3 = torch.empty((rows, kpad // 32, 2, 32), dtype=torch.bfloat16, device=x2d.device) if want_x3 else None
    check(_lib.lib().siu3r_split_bf16(_p(x2d), _p(hi), _p(lo), _p(x3), rows, k, kpad, x2d.stride(0), _stream()))
    return (hi, lo, kpad, x3) if want_x3 else (hi, lo, kpad)


def pack_matrix(w2d: torch.Tensor, bias: Optional[torch.Tensor], split: bool, **meta) -> PackedWeight:
    w2d = w2d.contiguous().float()
    hi, lo, kpad, x3 = split_bf16(w2d, split, want_x3=True) if split else (*split_bf16(w2d, False), None)  # x3: one 128-byte row segment per K tile = [hi 32 | lo 32]
    b = None if bias is None else bias.detach().float().contiguous()
    return PackedWeight(hi=hi, lo=lo, bias=b, x3=x3, n=w2d.shape[0], k=w2d.shape[1], kpad=kpad, meta=meta)


def pack_linear(weight, bias, split):
    return pack_matrix(weight, bias, split)


def pack_linear_ln(weight, bias, gamma, beta, split) -> PackedWeight:
    """A Linear behind a LayerNorm, folded:  LN(x) W^T + b = rstd (x W'^T - mean c1) + c2  with W' = W diag(gamma),
    c1[n] = sum_k W'[n,k] (of the ROUNDED operand the MFMA multiplies) and c2 = W beta + b.  The GEMM then reads the un-normalised
    rows and applies the row statistics in its epilogue (siu3r_gemm_params.ln_*)."""
    w = weight.float()
    pw = pack_matrix(w * gamma.float()[None, :], None, split)
    rounded = pw.hi.float() if pw.lo is None else pw.hi.float() + pw.lo.float()
    c2 = w @ beta.float()
    if bias is not None:
        c2 = c2 + bias.float()
    pw.meta = dict(pw.meta or {}, ln_c1=rounded.sum(1).contiguous(), ln_c2=c2.contiguous())
    return pw


def cat_packed(pws: Sequence[PackedWeight]) -> PackedWeight:
    """Packed weights with the same K concatenated along the OUTPUT axis (several Linears that read the same rows as one GEMM); folded
    LayerNorms may differ per part (their c1 / c2 are per output column)."""
    assert all(w.k == pws[0].k and w.kpad == pws[0].kpad and (w.bias is None) == (pws[0].bias is None) for w in pws)
    ct = lambda k: None if getattr(pws[0], k) is None else torch.cat([getattr(w, k) for w in pws], 0).contiguous()
    meta = dict(pws[0].meta or {})
    for k in ("ln_c1", "ln_c2"):
        if k in meta:
            meta[k] = torch.cat([w.meta[k] for w in pws]).contiguous()
    return PackedWeight(hi=ct("hi"), lo=ct("lo"), bias=ct("bias"), x3=ct("x3"), n=sum(w.n for w in pws), k=pws[0].k, kpad=pws[0].kpad, meta=meta)


def stack_packed(pws: Sequence[PackedWeight]) -> PackedWeight:
    """Weight sets of equal shape as ONE packed tensor with a leading group axis (the two decoder sides in one launch)."""
    st = lambda k: None if getattr(pws[0], k) is None else torch.stack([getattr(w, k) for w in pws]).contiguous()
    meta = dict(pws[0].meta or {})
    for k in ("ln_c1", "ln_c2"):
        if k in meta:
            meta[k] = torch.stack([w.meta[k] for w in pws]).contiguous()
    meta["groups"] = len(pws)
    return PackedWeight(hi=st("hi"), lo=st("lo"), bias=st("bias"), x3=st("x3"), n=pws[0].n, k=pws[0].k, kpad=pws[0].kpad, meta=meta)


class RowStats:
    """Per-row LayerNorm statistics of an activation tensor [..., C] (C <= 1024): fp32 [rows, ceil(C/64), 2] = (mean, centred sum
    of squares) of every 64-column group, written by the epilogue of the GEMM that produced the tensor (stats_out) and merged by
    the epilogue of the GEMM that consumes its LayerNorm (ln).  `base` is the full tensor the row numbering refers to."""

    def __init__(self, base: torch.Tensor):
        assert base.is_contiguous()
        self.base, self.C = base, base.shape[-1]
        self.tiles = (self.C + 63) // 64
        assert self.tiles <= 16, "folded LayerNorm supports up to 1024 channels"
        self.buf = torch.empty((base.numel() // self.C, self.tiles, 2), dtype=torch.float32, device=base.device)

    def row_of(self, view: torch.Tensor) -> int:
        off = view.storage_offset() - self.base.storage_offset()
        assert off % self.C == 0 and view.shape[-1] == self.C
        return off // self.C

    def ptr(self, row: int) -> int:
        return self.buf.data_ptr() + row * self.tiles * 8

    def div(self, stride: int) -> int:
        assert stride % self.C == 0, (stride, self.C)
        return stride // self.C


def pack_conv(weight: torch.Tensor, bias, split, cin_pad: Optional[int] = None) -> PackedWeight:
    """[Cout, Cin, KH, KW] -> [Cout, KH*KW*Cin'] in (ky, kx, c) order for the NHWC gather."""
    co, ci, kh, kw = weight.shape
    w = weight.permute(0, 2, 3, 1)
    if cin_pad and cin_pad != ci:
        w = torch.nn.functional.pad(w, (0, cin_pad - ci))  # layout plumbing (zero channels)
        ci = cin_pad
    return pack_matrix(w.reshape(co, kh * kw * ci), bias, split, kh=kh, kw=kw, cin=ci)


def pack_conv_transpose(weight: torch.Tensor, bias, split) -> PackedWeight:
    """ConvTranspose2d weight [Cin, Cout, k, k] with kernel == stride -> [(ky,kx,co), Cin]."""
    ci, co, k, _ = weight.shape
    w = weight.permute(2, 3, 1, 0).reshape(k * k * co, ci)
    return pack_matrix(w, bias, split, up=k, cout=co)


# ------------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------------
class KernelTimer:
    """Optional per-launch HIP-event timing of the MFMA GEMM kernel (bench.py's roofline leg).
    Events are recorded on the stream the kernel is launched on (torch's current stream)."""

    def __init__(self):
        self.records = []  # (variant, flops, ev0, ev1, shape)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for variant, fl, e0, e1, _ in self.records:
            d = out.setdefault(variant, dict(launches=0, flops=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += fl
            d["ms"] += e0.elapsed_time(e1)
        return out

    def by_shape(self):
        torch.cuda.synchronize()
        out = {}
        for variant, fl, e0, e1, shape in self.records:
            d = out.setdefault((variant,) + shape, dict(launches=0, flops=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += fl
            d["ms"] += e0.elapsed_time(e1)
        return out


_timer: Optional[KernelTimer] = None


def set_kernel_timer(t: Optional[KernelTimer]):
    global _timer
    _timer = t


def kernel_timer_active() -> bool:
    return _timer is not None


# ---- split-K workspace.  The library decides tile and K split (siu3r_gemm_plan); the caller lends it a workspace: fp32 slabs plus int32
# tickets that are zero at allocation and that every split launch leaves zero again (the tile's last arriver resets its ticket), so
# launches that are ordered behind each other reuse them back to back without a fill kernel.  Eager launches use one workspace per
# STREAM (launches on different streams may overlap).  A HIP-graph capture is its own scope (splitk_scope): torch captures every graph
# on the same internal stream, and the chain graphs of a forward replay concurrently -- the owner of the graph (model.py) measures
# what the chain's GEMMs ask for in an eager pass (splitk_meter), allocates exactly that, keeps it next to the graph (it dies with it)
# and lends it for the capture; a chain without a split GEMM gets none.
SPLITK_WS_FLOATS = 16 * 1024 * 1024   # 64 MiB of slabs per eager stream
SPLITK_COUNTERS = 8192
_sk_pool: dict = {}
_sk_scope = None    # None: per-stream pool; ("scope", (slabs, tickets) | None): what the enclosing splitk_scope lends
_sk_meter = None


class splitk_meter:
    """with splitk_meter() as m: ...  -> m.floats / m.counters = the largest split-K workspace (siu3r_gemm_plan's ws_floats / counters)
    any GEMM launched inside asked for; both 0 when none of them splits."""

    def __enter__(self):
        global _sk_meter
        self.prev, _sk_meter = _sk_meter, self
        self.floats = self.counters = 0
        return self

    def __exit__(self, *exc):
        global _sk_meter
        _sk_meter = self.prev

    def workspace(self, dev):
        """a workspace of exactly the measured size (None if nothing splits); the tickets start at zero"""
        if self.floats == 0:
            return None
        return (torch.empty((self.floats,), dtype=torch.float32, device=dev), torch.zeros((max(1, self.counters),), dtype=torch.int32, device=dev))


class splitk_scope:
    """with splitk_scope(ws): GEMMs launched inside split K through `ws` = (slabs, tickets) -- allocated and zeroed by the caller, i.e.
    BEFORE a graph capture that the caller opens inside the block -- instead of the current stream's workspace; ws None = no split."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        global _sk_scope
        self.prev, _sk_scope = _sk_scope, ("scope", self.ws)
        return self

    def __exit__(self, *exc):
        global _sk_scope
        _sk_scope = self.prev


def _splitk_workspace(dev):
    if _sk_scope is not None:
        return _sk_scope[1]
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    ws = _sk_pool.get(key)
    if ws is None:
        ws = (torch.empty((SPLITK_WS_FLOATS,), dtype=torch.float32, device=dev), torch.zeros((SPLITK_COUNTERS,), dtype=torch.int32, device=dev))
        _sk_pool[key] = ws
    return ws


def gemm_plan(p: GemmParams) -> "_lib.GemmPlan":
    pl = _lib.GemmPlan()
    check(_lib.lib().siu3r_gemm_plan(C.byref(p), C.byref(pl)))
    return pl


_plan_log: Optional[list] = None  # tests / tools: when a list, every GEMM launch appends its siu3r_gemm_plan_t


def set_plan_log(log: Optional[list]):
    global _plan_log
    _plan_log = log


def gemm_tune(key: int, value: int):
    """siu3r_gemm_tune: 0 = default tile_cfg (0 auto, -1 128x64 family, 1..3 ping-pong 256x256 / 256x128 / 128x128), 1 = no skinny rows, 2 = no split-K, 3 = ignore the measured-choice table, 4 = skinny rows whenever applicable"""
    check(_lib.lib().siu3r_gemm_tune(key, value))


class Planes:
    """Pre-split bf16x3 copy of an fp32 activation [..., C] (siu3r_gemm_params.c_x3 / a_x3): per row and 32 columns the upper 16 bits of
    the 32 values, then bf16(value - upper): the two operands a bf16x3 product multiplies, which the ping-pong GEMM otherwise derives
    inside its K loop (12-18 % of a launch).  Same size and strides as the fp32 tensor (`t`: opaque storage).  A producer writes them
    when its plan can (`valid`), a consumer reads them when ITS plan can; otherwise both use the fp32 tensor."""

    def __init__(self, like: torch.Tensor, storage: Optional[torch.Tensor] = None):
        assert like.dtype == torch.float32
        self.t = storage if storage is not None else torch.empty_like(like)
        assert self.t.shape == like.shape and self.t.stride() == like.stride() and self.t.dtype == torch.float32
        self.valid = False
        self.only = False  # True: the fp32 tensor was NOT written (planes-only output).  A MIXED buffer (planes_from_col: storage is the
        # fp32 output itself, planes behind that column) keeps `only` False: its reader must know the column split, as the q | k | v users do


_NO_PRESPLIT = bool(os.environ.get("SIU3R_NO_PRESPLIT"))  # A/B switch: never emit / consume pre-split planes


def _attach_workspace(p: GemmParams, dev=None):
    """the split-K workspace of the current scope (the plan depends on its capacity)"""
    if p.splitk != 1 and not p.sk_ws:
        wsp = _splitk_workspace(dev if dev is not None else torch.device("cuda", torch.cuda.current_device()))
        if wsp is not None:
            ws, cnt = wsp
            p.sk_ws, p.sk_cnt, p.sk_ws_floats, p.sk_cnt_n = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()


def _gemm_launch(p: GemmParams, dev=None):
    if p.splitk != 1:
        _attach_workspace(p, dev)
        if _sk_meter is not None:
            pl = gemm_plan(p)
            if pl.splitk > 1:
                _sk_meter.floats, _sk_meter.counters = max(_sk_meter.floats, pl.ws_floats), max(_sk_meter.counters, pl.counters)
    if _plan_log is not None:
        _plan_log.append(gemm_plan(p))
    if _timer is None:
        check(_lib.lib().siu3r_gemm(C.byref(p), _stream()))
        return
    # label with the kernel the launcher picks, so that the event averages line up with rocprofv3's per-kernel rows
    pl = gemm_plan(p)
    variant = pl.kernel.decode()
    flops = 2.0 * p.m * p.n * p.k * max(1, p.batch)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(_lib.lib().siu3r_gemm(C.byref(p), _stream()))
    e1.record()
    _timer.records.append((variant, flops, e0, e1, (p.m, p.n, p.k, max(1, p.batch), p.a_mode, p.out_mode, p.kh, pl.tile_cfg, pl.splitk, pl.skinny_rows,
                                                    int(bool(p.ln_stats)), int(bool(p.rope_cos)), int(bool(p.residual)))))


_trace_buf: Optional[torch.Tensor] = None  # tools/gemm_trace.py: int64 [workgroups, 8] stamp buffer (tuning only)


def set_gemm_trace(buf: Optional[torch.Tensor]):
    global _trace_buf
    _trace_buf = buf


def _fill_common(p: GemmParams, a, pw: PackedWeight, out, act, residual, relu_in):
    p.a, p.w_hi, p.w_lo, p.c = _p(a), _p(pw.hi), _p(pw.lo), _p(out)
    p.w_x3 = _p(pw.x3)
    p.trace = _p(_trace_buf)
    p.bias = _p(pw.bias)
    p.residual = _p(residual)
    p.n, p.k, p.kpad = pw.n, pw.k, pw.kpad
    p.a_dtype, p.c_dtype = _dt(a), _dt(out)
    p.r_dtype = _dt(residual) if residual is not None else F32
    p.act, p.relu_in = act, int(relu_in)
    p.batch = 1
    if pw.lo is not None and a.dtype != torch.float32:
        raise RuntimeError("bf16x3 weights need fp32 activations")


def _rows_layout(t: torch.Tensor):
    """Describe a [..., C] tensor (last dim contiguous) as Z batches of M rows: (Z, M, row_stride, batch_stride)."""
    assert t.stride(-1) == 1
    if t.dim() == 1:
        return 1, 1, t.shape[0], 0
    if t.dim() == 2:
        return 1, t.shape[0], t.stride(0), 0
    # collapse leading dims when they form a single row stride
    shape, stride = t.shape[:-1], t.stride()[:-1]
    ok = all(stride[i] == stride[i + 1] * shape[i + 1] for i in range(len(shape) - 1))
    if ok:
        m = 1
        for s_ in shape:
            m *= s_
        return 1, m, stride[-1], 0
    if t.dim() == 3:
        return t.shape[0], t.shape[1], t.stride(1), t.stride(0)
    # [Z, ..., C] with collapsible inner dims
    inner_ok = all(stride[i] == stride[i + 1] * shape[i + 1] for i in range(1, len(shape) - 1))
    if inner_ok:
        m = 1
        for s_ in shape[1:]:
            m *= s_
        return shape[0], m, stride[-1], stride[0]
    raise RuntimeError(f"unsupported strided layout {tuple(t.shape)} / {tuple(t.stride())}")


def _apply_ln_stats_aux(p: GemmParams, pw: PackedWeight, x, out, lda, ldc, strides, ln, stats_out, aux_out, x_first=None):
    """strides = (sa, sa_i, sc, sc_i): batch strides of A and C in elements (0 when absent); x_first: the view whose storage offset
    locates A's first row (differs from x for a flipped group order)."""
    sa, sa_i, sc, sc_i = strides
    if ln is not None:
        assert isinstance(ln, RowStats) and "ln_c1" in (pw.meta or {}), "ln= needs a weight packed with pack_linear_ln"
        assert pw.bias is None and ln.C == pw.k
        p.ln_stats = ln.ptr(ln.row_of(x_first if x_first is not None else x))
        p.ln_c1, p.ln_c2 = _p(pw.meta["ln_c1"]), _p(pw.meta["ln_c2"])
        p.ln_tiles, p.ln_eps = ln.tiles, float(pw.meta.get("ln_eps", 1e-6))
        p.ln_ldm, p.ln_sz, p.ln_sz_i = ln.div(lda), (ln.div(abs(sa)) if sa else 0), (ln.div(abs(sa_i)) * (1 if sa_i >= 0 else -1) if sa_i else 0)
    if stats_out is not None:
        assert isinstance(stats_out, RowStats) and stats_out.C == pw.n and out.dtype == torch.float32
        p.stats_out = stats_out.ptr(stats_out.row_of(out))
        p.st_ldm, p.st_sz, p.st_sz_i = stats_out.div(ldc), (stats_out.div(sc) if sc else 0), (stats_out.div(sc_i) if sc_i else 0)
    if aux_out is not None:
        assert aux_out.dtype == torch.bfloat16 and aux_out.shape == out.shape and aux_out.stride() == out.stride()
        p.c_aux = _p(aux_out)


def linear(x: torch.Tensor, pw: PackedWeight, *, out_dtype=torch.float32, act=ACT_NONE,
           residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, relu_in=False,
           rope=None, ln: Optional[RowStats] = None, stats_out: Optional[RowStats] = None, aux_out: Optional[torch.Tensor] = None,
           splitk: int = 0, a_planes: Optional[Planes] = None, planes_out: Optional[Planes] = None, planes_only: bool = False,
           dry_run: bool = False, planes_from_col: int = 0):
    """y[..., N] = act(x[..., K] @ W^T + b) (+ residual).  x / out / residual may be strided views whose last
    dim is contiguous and which decompose into Z batches of M rows (e.g. tokens[:, :-1]).
    rope = (cos, sin, positions int64 [rows, 2] contiguous, ncols): fused RoPE2D on output columns [0, ncols).
    ln: x holds UN-normalised rows and pw was packed by pack_linear_ln: the LayerNorm is applied in the epilogue from these row
    statistics.  stats_out: write the row statistics of the (fp32) output for a later ln=.  aux_out: a bf16 copy of the output.
    splitk: siu3r_gemm_params.splitk (0 = the library decides, 1 = never, > 1 = exactly that many K slices).
    a_planes: pre-split planes of x (Planes): read instead of x when they are valid and this launch's plan can (bit-identical result).
    planes_out: Planes of the output, written when the plan can (planes_out.valid says so afterwards); planes_only: then do NOT write the
    fp32 output (`out` stays untouched; planes_out.only = True).  dry_run: launch nothing, return the siu3r_gemm_plan_t of the launch as it
    would be issued with fp32 A (its a_x3_ok / c_x3_ok tell whether planes could be consumed / emitted).  planes_from_col (with
    planes_out whose storage IS `out`): a mixed output -- fp32 in columns [0, planes_from_col), planes behind them."""
    _gpu(x, residual, out)
    assert x.shape[-1] == pw.k, (x.shape, pw.k)
    if out is None:
        out = torch.empty((*x.shape[:-1], pw.n), dtype=out_dtype, device=x.device)
    zx, mx, lda, sa = _rows_layout(x)
    zo, mo, ldc, sc = _rows_layout(out)
    layouts = [(zx, mx), (zo, mo)]
    zr = mr = ldr = sr = 0
    if residual is not None:
        zr, mr, ldr, sr = _rows_layout(residual)
        layouts.append((zr, mr))
    Z = max(z for z, _ in layouts)
    total = zx * mx
    assert all(z * m == total for z, m in layouts), layouts
    p = GemmParams()
    _fill_common(p, x, pw, out, act, residual, relu_in)
    bsa = bsc = 0
    if Z == 1:
        p.m, p.lda, p.ldc, p.ldr = total, lda, ldc, ldr
    else:
        M = total // Z
        def bs(z, m, ld, s):  # a collapsed operand is re-split into Z batches of M rows
            return s if z == Z else M * ld
        assert all(z in (1, Z) for z, _ in layouts)
        p.m, p.lda, p.ldc, p.ldr = M, lda, ldc, ldr
        bsa, bsc = bs(zx, mx, lda, sa), bs(zo, mo, ldc, sc)
        p.batch, p.sa, p.sw, p.sc = Z, bsa, 0, bsc
        p.sr = bs(zr, mr, ldr, sr) if residual is not None else 0
    if rope is not None:
        cos, sin, pos, ncols = rope
        assert pos.dtype == torch.int64 and pos.is_contiguous() and pos.numel() == 2 * total and cos.shape[1] == 16
        p.rope_cos, p.rope_sin, p.rope_pos, p.rope_ncols = _p(cos), _p(sin), _p(pos), ncols
    _apply_ln_stats_aux(p, pw, x, out, lda, ldc, (bsa, 0, bsc, 0), ln, stats_out, aux_out)
    p.splitk = splitk
    if dry_run or a_planes is not None or planes_out is not None:
        p.c_x3_col0 = planes_from_col
        pl = _apply_planes(p, x, out, a_planes, planes_out, planes_only and planes_from_col == 0, dry_run)
        if dry_run:
            return pl
    _gemm_launch(p)
    return out


def _apply_planes(p: GemmParams, x, out, a_planes, planes_out, planes_only, dry_run):
    """pre-split operands of a prepared launch (see linear()): query the plan with fp32 A, switch A / add the plane output where the plan
    can and the planes exist.  Returns the plan."""
    _attach_workspace(p)
    if planes_out is not None:
        p.c_x3 = _p(planes_out.t)  # (alignment of the destination is part of c_x3_ok)
    pl = gemm_plan(p)
    p.c_x3 = None
    if dry_run:
        return pl
    if a_planes is not None and a_planes.valid and not _NO_PRESPLIT:
        assert a_planes.t.shape == x.shape and a_planes.t.stride() == x.stride()
        if pl.a_x3_ok:
            p.a, p.a_x3 = _p(a_planes.t), 1
        else:
            assert not a_planes.only, f"the fp32 operand of this GEMM was never written (planes only) and {pl.kernel.decode()} cannot read planes"
    if planes_out is not None:
        assert planes_out.t.shape == out.shape and planes_out.t.stride() == out.stride()
        planes_out.valid = bool(pl.c_x3_ok) and not _NO_PRESPLIT
        planes_out.only = planes_out.valid and planes_only
        if planes_out.valid:
            p.c_x3 = _p(planes_out.t)
            if planes_only:
                p.c = None
    return pl


def linear_grouped(x: torch.Tensor, pw: PackedWeight, *, out_dtype=torch.float32, act=ACT_NONE, residual: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None, rope=None, ln: Optional[RowStats] = None, stats_out: Optional[RowStats] = None,
                   aux_out: Optional[torch.Tensor] = None, flip: bool = False, planes_out: Optional[Planes] = None, planes_from_col: int = 0):
    """G weight sets in ONE launch: x [B, G, M, K] (strided, last dim contiguous) times pw = stack_packed([...G sets]) ->
    out [B, G, M, N]; group g of every batch item uses weight set g (blockIdx.z = b * G + g).  flip: group g reads the rows of
    group G-1-g of x (and their statistics) -- the cross-attention memory of a decoder side is the other view's tokens.
    planes_out / planes_from_col: pre-split output columns, as in linear()."""
    _gpu(x, residual, out)
    G = pw.meta["groups"]
    assert x.dim() == 4 and x.shape[1] == G and x.shape[-1] == pw.k and x.stride(3) == 1, (x.shape, G, pw.k)
    B, _, M, _ = x.shape
    if out is None:
        out = torch.empty((B, G, M, pw.n), dtype=out_dtype, device=x.device)
    assert out.shape == (B, G, M, pw.n) and out.stride(3) == 1
    p = GemmParams()
    xf = x[:, G - 1] if flip else x
    _fill_common(p, xf, pw, out, act, residual, False)
    p.m, p.lda, p.ldc = M, x.stride(2), out.stride(2)
    p.batch, p.bmod = B * G, G
    sa_i = -x.stride(1) if flip else x.stride(1)
    p.sa, p.sa_i, p.sc, p.sc_i = x.stride(0), sa_i, out.stride(0), out.stride(1)
    p.sw = pw.hi.stride(0)
    p.sbias = pw.n
    if residual is not None:
        assert residual.shape == out.shape and residual.stride(3) == 1
        p.ldr, p.sr, p.sr_i = residual.stride(2), residual.stride(0), residual.stride(1)
    if rope is not None:
        cos, sin, pos, ncols = rope
        assert pos.dtype == torch.int64 and pos.is_contiguous() and pos.numel() == 2 * B * G * M and cos.shape[1] == 16
        p.rope_cos, p.rope_sin, p.rope_pos, p.rope_ncols = _p(cos), _p(sin), _p(pos), ncols
    _apply_ln_stats_aux(p, pw, x, out, p.lda, p.ldc, (p.sa, sa_i, p.sc, p.sc_i), ln, stats_out, aux_out, x_first=xf)
    if planes_out is not None:
        p.c_x3_col0 = planes_from_col
        _apply_planes(p, x, out, None, planes_out, False, False)
    _gemm_launch(p)
    return out


def bmm_nt(a: torch.Tensor, b_hi: torch.Tensor, b_lo: Optional[torch.Tensor], n: int, k: int, *, out_dtype=torch.float32,
           b_x3: Optional[torch.Tensor] = None):
    """out[z, M, n] = a[z, M, k] @ b[z, n, k]^T with pre-split bf16 b planes [Z, n, kpad] (b_x3: the interleaved planes of
    split_bf16(want_x3=True), which put a bf16x3 product on the LDS-DMA kernel)."""
    _gpu(a, b_hi)
    Z, M, K = a.shape
    assert K == k and a.stride(2) == 1
    out = torch.empty((Z, M, n), dtype=out_dtype, device=a.device)
    p = GemmParams()
    p.a, p.w_hi, p.w_lo, p.c = _p(a), _p(b_hi), _p(b_lo), _p(out)
    p.w_x3 = _p(b_x3)
    p.m, p.n, p.k, p.kpad = M, n, k, b_hi.shape[-1]
    p.lda, p.ldc = a.stride(1), n
    p.a_dtype, p.c_dtype, p.r_dtype = _dt(a), _dt(out), F32
    p.batch, p.sa, p.sw, p.sc = Z, a.stride(0), b_hi.stride(0), out.stride(0)
    _gemm_launch(p)
    return out


def conv2d(x: torch.Tensor, pw: PackedWeight, *, stride=1, pad=0, out_dtype=torch.float32, act=ACT_NONE,
           residual=None, relu_in=False, up_src: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           a_planes: Optional[Planes] = None, planes_out: Optional[Planes] = None, planes_only: bool = False, dry_run: bool = False):
    """NHWC implicit-GEMM convolution: x [B, IH, IW, Cin] -> [B, OH, OW, Cout] (out: optional contiguous destination).
    a_planes / planes_out / planes_only / dry_run: pre-split bf16x3 operands, as in linear()."""
    _gpu(x, residual, up_src, out)
    assert x.is_contiguous()
    B, IH, IW, Cin = x.shape
    kh, kw = pw.meta["kh"], pw.meta["kw"]
    assert Cin == pw.meta["cin"], (Cin, pw.meta)
    OH = (IH + 2 * pad - kh) // stride + 1
    OW = (IW + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((B, OH, OW, pw.n), dtype=out_dtype, device=x.device)
    assert out.shape == (B, OH, OW, pw.n) and out.is_contiguous()
    p = GemmParams()
    _fill_common(p, x, pw, out, act, residual, relu_in)
    p.m, p.ldc, p.ldr = B * OH * OW, pw.n, pw.n
    p.a_mode = 1
    p.ih, p.iw, p.cin, p.kh, p.kw, p.stride, p.pad, p.oh, p.ow = IH, IW, Cin, kh, kw, stride, pad, OH, OW
    if up_src is not None:
        assert up_src.is_contiguous() and up_src.shape == (B, OH // 2, OW // 2, pw.n)
        p.up_src, p.up_dtype = _p(up_src), _dt(up_src)
    if dry_run or a_planes is not None or planes_out is not None:
        pl = _apply_planes(p, x, out, a_planes, planes_out, planes_only, dry_run)
        if dry_run:
            return pl
    _gemm_launch(p)
    return out


def conv_transpose2d(x: torch.Tensor, pw: PackedWeight, *, out_dtype=torch.float32, residual=None):
    """ConvTranspose2d with kernel == stride (pixel shuffle epilogue): [B,IH,IW,Cin] -> [B,IH*up,IW*up,Cout].  x may be a view whose
    batch items are dense but lie further apart: each item is then one batch of the launch (blockIdx.z)."""
    _gpu(x, residual)
    B, IH, IW, Cin = x.shape
    x_bs = _batch_strided(x, IH * IW * Cin)
    up, cout = pw.meta["up"], pw.meta["cout"]
    out = torch.empty((B, IH * up, IW * up, cout), dtype=out_dtype, device=x.device)
    p = GemmParams()
    _fill_common(p, x, pw, out, ACT_NONE, residual, False)
    p.out_mode, p.up, p.cout, p.ih, p.iw = 1, up, cout, IH, IW
    if B == 1 or x_bs == IH * IW * Cin:
        p.m, p.lda = B * IH * IW, Cin
    else:
        assert residual is None or residual.is_contiguous()
        p.m, p.lda = IH * IW, Cin
        p.batch, p.sa, p.sw, p.sc = B, x_bs, 0, out[0].numel()
        p.sr = out[0].numel() if residual is not None else 0
    _gemm_launch(p)
    return out


def _grouped_common(p: GemmParams, x, pw: PackedWeight, out, residual):
    """two-level batching of a grouped launch over images [B, G, ...]: blockIdx.z = b * G + g, weight set g"""
    G = pw.meta["groups"]
    B = x.shape[0]
    assert x.shape[1] == G and x.is_contiguous() and out.is_contiguous()
    p.batch, p.bmod = B * G, G
    p.sa, p.sa_i, p.sc, p.sc_i = x.stride(0), x.stride(1), out.stride(0), out.stride(1)
    # (bias elements per weight set: n, or cout for the conv-transpose whose bias is per output channel of the pixel shuffle)
    p.sw, p.sbias = pw.hi.stride(0), (pw.bias.stride(0) if pw.bias is not None else pw.n)
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous()
        p.sr, p.sr_i = residual.stride(0), residual.stride(1)


def conv2d_grouped(x: torch.Tensor, pw: PackedWeight, *, stride=1, pad=0, out_dtype=torch.float32, act=ACT_NONE, residual=None, relu_in=False,
                   out: Optional[torch.Tensor] = None, a_planes: Optional[Planes] = None, planes_out: Optional[Planes] = None, planes_only: bool = False,
                   dry_run: bool = False):
    """conv2d for G networks of the same shape in ONE launch: x [B, G, IH, IW, Cin] -> [B, G, OH, OW, Cout], image (b, g) convolved with
    weight set g of pw = stack_packed([...G sets]) (the two DPT heads of a kind: head1 on view 0, head2 on view 1 -- reference
    model.py:352-375 runs them one after the other)."""
    _gpu(x, residual)
    B, G, IH, IW, Cin = x.shape
    kh, kw = pw.meta["kh"], pw.meta["kw"]
    assert Cin == pw.meta["cin"], (Cin, pw.meta)
    OH = (IH + 2 * pad - kh) // stride + 1
    OW = (IW + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((B, G, OH, OW, pw.n), dtype=out_dtype, device=x.device)
    assert out.shape == (B, G, OH, OW, pw.n) and out.is_contiguous()
    p = GemmParams()
    _fill_common(p, x, pw, out, act, residual, relu_in)
    p.m, p.ldc, p.ldr = OH * OW, pw.n, pw.n
    p.a_mode = 1
    p.ih, p.iw, p.cin, p.kh, p.kw, p.stride, p.pad, p.oh, p.ow = IH, IW, Cin, kh, kw, stride, pad, OH, OW
    _grouped_common(p, x, pw, out, residual)
    if dry_run or a_planes is not None or planes_out is not None:  # pre-split bf16x3 operands, as in linear()
        pl = _apply_planes(p, x, out, a_planes, planes_out, planes_only, dry_run)
        if dry_run:
            return pl
    _gemm_launch(p)
    return out


def conv_transpose2d_grouped(x: torch.Tensor, pw: PackedWeight, *, out_dtype=torch.float32):
    """conv_transpose2d (kernel == stride) for G weight sets in one launch: x [B, G, IH, IW, Cin] -> [B, G, IH*up, IW*up, Cout]"""
    _gpu(x)
    B, G, IH, IW, Cin = x.shape
    up, cout = pw.meta["up"], pw.meta["cout"]
    out = torch.empty((B, G, IH * up, IW * up, cout), dtype=out_dtype, device=x.device)
    p = GemmParams()
    _fill_common(p, x, pw, out, ACT_NONE, None, False)
    p.out_mode, p.up, p.cout, p.ih, p.iw = 1, up, cout, IH, IW
    p.m, p.lda = IH * IW, Cin
    _grouped_common(p, x, pw, out, None)
    _gemm_launch(p)
    return out


def patch_embed(img: torch.Tensor, pw: PackedWeight, out: torch.Tensor, stats_out: Optional[RowStats] = None, aux_out: Optional[torch.Tensor] = None):
    """Conv2d(3->C, k16, s16) on an NCHW fp32 image, written token-major into out[b, :h*w, :]
    (out is [B, Ntok(+extra), C] fp32; reference croco/patch_embed.py:19-29)."""
    _gpu(img, out)
    assert img.dtype == torch.float32 and img.is_contiguous()
    B, _, H, W = img.shape
    assert H % 16 == 0, f"Input image height ({H}) is not a multiple of patch size (16)."
    assert W % 16 == 0, f"Input image width ({W}) is not a multiple of patch size (16)."
    h, w = H // 16, W // 16
    p = GemmParams()
    _fill_common(p, img, pw, out, ACT_NONE, None, False)
    p.a_mode, p.ih, p.iw, p.oh, p.ow = 2, H, W, h, w
    if out.stride(0) == h * w * out.stride(1):
        p.m, p.ldc = B * h * w, out.stride(1)
    else:  # extra tokens per batch item: one launch per batch via blockIdx.z
        p.m, p.ldc = h * w, out.stride(1)
        p.batch, p.sa, p.sw, p.sc = B, 3 * H * W, 0, out.stride(0)
    if stats_out is not None:
        p.stats_out = stats_out.ptr(stats_out.row_of(out))
        p.st_ldm, p.st_sz = stats_out.div(p.ldc), (stats_out.div(p.sc) if p.batch > 1 else 0)
    if aux_out is not None:
        assert aux_out.dtype == torch.bfloat16 and aux_out.shape == out.shape and aux_out.stride() == out.stride()
        p.c_aux = _p(aux_out)
    _gemm_launch(p)
    return out


# ------------------------------------------------------------------------------------------------
# attention / norms / element-wise
# ------------------------------------------------------------------------------------------------
_ATTN_SPLITKV = __import__("os").environ.get("SIU3R_NO_ATTN_SPLITKV", "0") != "1"


def attention(q, k, v, *, heads: int, head_dim: int, scale: float, rope=None, qpos=None, kpos=None,
              mask: Optional[torch.Tensor] = None, split3=False, kv_bxor: int = 0, kv_planes: bool = False, dry_run: bool = False):
    """q [B,Nq,H,D] / k,v [B,Nk,H,D] strided views (D contiguous) -> out [B,Nq,H*D].  kv_bxor: batch item b attends to the keys / values
    of batch item b ^ kv_bxor (siu3r_attn_params.kv_bxor).  kv_planes: k and v are views of PRE-SPLIT planes (a projection written with
    planes_from_col); dry_run: launch nothing, return whether this launch could read such planes (siu3r_attention_kv_x3_ok)."""
    _gpu(q, k, v, mask)
    B, Nq = q.shape[0], q.shape[1]
    Nk = k.shape[1]
    out = q if dry_run else torch.empty((B, Nq, heads * head_dim), dtype=q.dtype, device=q.device)  # (a dry run only needs a non-null pointer)
    p = AttnParams()
    p.q, p.k, p.v, p.out = _p(q), _p(k), _p(v), _p(out)
    p.dtype = _dt(q)
    p.B, p.H, p.Nq, p.Nk, p.D = B, heads, Nq, Nk, head_dim
    p.q_sb, p.q_sn, p.q_sh = q.stride(0), q.stride(1), q.stride(2)
    p.k_sb, p.k_sn, p.k_sh = k.stride(0), k.stride(1), k.stride(2)
    p.v_sb, p.v_sn, p.v_sh = v.stride(0), v.stride(1), v.stride(2)
    p.scale = scale
    if rope is not None:
        cos, sin = rope
        p.rope_cos, p.rope_sin, p.rope_max_pos = _p(cos), _p(sin), cos.shape[0]
        assert qpos.dtype == torch.int64 and kpos.dtype == torch.int64 and qpos.is_contiguous() and kpos.is_contiguous()
        p.qpos, p.kpos = _p(qpos), _p(kpos)
    if mask is not None:  # [B, Nq, ld] uint8, ld >= Nk and a multiple of 64 (padding ignored)
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.shape[:2] == (B, Nq) and mask.shape[2] >= Nk
        p.mask, p.mask_ld = _p(mask), mask.shape[2]
    p.split3 = int(split3)
    p.kv_bxor = kv_bxor
    p.kv_x3 = int(kv_planes)
    ws = None
    fast = rope is None and ((q.dtype == torch.bfloat16 and not split3) or (q.dtype == torch.float32 and split3))
    if fast and _ATTN_SPLITKV:
        nkt = (Nk + 63) // 64
        qt = (Nq + 127) // 128
        splits = 1
        if Nq <= 128 and Nk >= 1024:
            # few queries against many keys (Mask2Former: 100 queries x 512..8192 keys): split the keys over workgroups
            splits = min(16, nkt // 4)
        # (splitting the ViT attention of one pair -- 288 workgroups on 512 slots -- into 2-3 key ranges measured no gain: 36 -> 39 us
        # bf16, 78 -> 76 us bf16x3, the combine pass eats what the extra workgroups win)
        if splits > 1:
            p.splits = splits
            ws = torch.empty((B, heads, splits, qt * 128, head_dim + 4), dtype=torch.float32, device=q.device)
            p.ws = _p(ws)
    if dry_run:  # (asked of exactly the parameter block the launch would carry, key split included)
        return bool(_lib.lib().siu3r_attention_kv_x3_ok(C.byref(p))) and not _NO_PRESPLIT
    check(_lib.lib().siu3r_attention(C.byref(p), _stream()))
    return out


def layernorm2(x: torch.Tensor, gamma, beta, eps: float):
    """LayerNorm with two outputs: (fp32, bf16 copy).  x contiguous fp32."""
    _gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    out2 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().siu3r_layernorm2(_p(x), _p(out), F32, _p(out2), _p(gamma), _p(beta), rows, Cc, Cc, Cc, Cc, eps, _stream()))
    return out, out2


def layernorm(x: torch.Tensor, gamma, beta, eps: float, out_dtype=torch.float32):
    _gpu(x)
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    x2 = x.reshape(rows, Cc) if x.is_contiguous() else x
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    if x.is_contiguous():
        ldx = Cc
    else:  # [B, N, C] view with a batch stride: handle per batch
        assert x.dim() == 3
        for b in range(x.shape[0]):
            check(_lib.lib().siu3r_layernorm(_p(x[b]), _p(out[b]), _dt(out), _p(gamma), _p(beta), x.shape[1], Cc,
                                             x.stride(1), Cc, eps, _stream()))
        return out
    check(_lib.lib().siu3r_layernorm(_p(x2), _p(out), _dt(out), _p(gamma), _p(beta), rows, Cc, ldx, Cc, eps, _stream()))
    return out


def add(a: torch.Tensor, b: torch.Tensor):
    """fp32 a[rows, C] + b[b_rows, C] broadcast over rows (row r uses b[r % b_rows])."""
    _gpu(a, b)
    assert a.is_contiguous() and b.is_contiguous() and a.dtype == torch.float32 and b.dtype == torch.float32
    Cc = a.shape[-1]
    y = torch.empty_like(a)
    check(_lib.lib().siu3r_add(_p(a), _p(b), _p(y), a.numel() // Cc, b.numel() // Cc, Cc, _stream()))
    return y


def pack_image_nhwc(img: torch.Tensor, out_dtype, cpad: int = 8):
    """[N,3,H,W] fp32 -> [N,H,W,cpad] channel-last, zero-padded channels (cpad 8, or 4 for the fp32 image of the bf16x3 mode)."""
    _gpu(img)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.shape[1] == 3
    N, _, H, W = img.shape
    out = torch.empty((N, H, W, cpad), dtype=out_dtype, device=img.device)
    check(_lib.lib().siu3r_pack_image_nhwc(_p(img), _p(out), _dt(out), N, H, W, cpad, _stream()))
    return out


def pack_image_nhwc8(img: torch.Tensor, out_dtype):
    return pack_image_nhwc(img, out_dtype, 8)


def pack_stem7(weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]]):
    """MFMA B fragments of G stem convolutions [256, 3, 7, 7] for siu3r_stem7x7_x3 -> (bf16 [G, 8, 14, 2, 64, 8], fp32 bias [G, 256] or None).
    K = 224 ordered (ky, kx in 0..7, c in 0..3) with zero weights at kx = 7 and c = 3; fragment of (wave w, K step s): lane l holds channel
    32 w + l % 32 and k = 16 s + 8 (l // 32) + 0..7; planes hi = bf16(w) (round to nearest even), lo = bf16(w - hi): the split of every packed weight."""
    frs = []
    for w in weights:
        _gpu(w)
        assert tuple(w.shape) == (256, 3, 7, 7), w.shape
        wk = torch.zeros((256, 7, 8, 4), dtype=torch.float32, device=w.device)
        wk[:, :, :7, :3] = w.detach().float().permute(0, 2, 3, 1)
        wk = wk.reshape(256, 14, 2, 8)                                   # [n, s, chunk, j]: k = 16 s + 8 chunk + j
        fr = wk.reshape(8, 32, 14, 2, 8).permute(0, 2, 3, 1, 4)          # [wave, s, chunk, n % 32, j]
        fr = fr.reshape(8, 14, 64, 8)                                    # lane = 32 chunk + n % 32
        hi = fr.to(torch.bfloat16)
        lo = (fr - hi.float()).to(torch.bfloat16)
        frs.append(torch.stack([hi, lo], dim=2))                         # [8, 14, 2, 64, 8]
    b = None
    if any(x is not None for x in biases):
        b = torch.stack([torch.zeros(256, device=weights[0].device) if x is None else x.detach().float() for x in biases]).contiguous()
    return torch.stack(frs).contiguous(), b


def stem7x7_x3(img: torch.Tensor, wfrag: torch.Tensor, bias: Optional[torch.Tensor], up_src: Optional[torch.Tensor], out: torch.Tensor, planes: bool = False):
    """out = ReLU(conv7x7(img) + bias) + x2 upsample of up_src, G heads in one launch (siu3r_stem7x7_x3): img [B, G, H, W, 4] fp32,
    up_src [B, G, H/2, W/2, 256], out [B, G, H, W, 256] fp32 -- written as values or, planes=True, as the pre-split planes of Planes."""
    _gpu(img, wfrag, bias, up_src, out)
    B, G, H, W, c = img.shape
    assert c == 4 and img.dtype == torch.float32 and img.is_contiguous() and tuple(wfrag.shape) == (G, 8, 14, 2, 64, 8) and wfrag.dtype == torch.bfloat16
    assert out.shape == (B, G, H, W, 256) and out.dtype == torch.float32 and out.is_contiguous()
    assert bias is None or (tuple(bias.shape) == (G, 256) and bias.is_contiguous())
    assert up_src is None or (up_src.shape == (B, G, H // 2, W // 2, 256) and up_src.dtype == torch.float32 and up_src.is_contiguous())
    check(_lib.lib().siu3r_stem7x7_x3(_p(img), _p(wfrag), _p(bias), _p(up_src), _p(out), B, G, H, W, 1 if planes else 0, _stream()))
    return out


def pack_proj(weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]]):
    """MFMA B fragments of G projection matrices [N, K] (a 1 x 1 convolution's [N, K, 1, 1] is reshaped) for siu3r_proj_rows_x3 ->
    (bf16 [G, NB, K/16, 2, 64, 8], fp32 bias [G, 32 NB] or None, N).  Fragment of (column block nb, K step s): lane l holds column
    32 nb + l % 32 (zero beyond N) and k = 16 s + 8 (l // 32) + 0..7; planes hi = bf16(w), lo = bf16(w - hi)."""
    frs, N = [], None
    for w in weights:
        _gpu(w)
        w = w.detach().float().reshape(w.shape[0], -1)
        N, K = w.shape
        NB = (N + 31) // 32
        assert K % 16 == 0
        wp = torch.zeros((NB * 32, K), dtype=torch.float32, device=w.device)
        wp[:N] = w
        fr = wp.reshape(NB, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(NB, K // 16, 64, 8)  # [nb, s, lane = 32 chunk + n % 32, j]
        hi = fr.to(torch.bfloat16)
        lo = (fr - hi.float()).to(torch.bfloat16)
        frs.append(torch.stack([hi, lo], dim=2))
    b = None
    if any(x is not None for x in biases):
        NB = frs[0].shape[0]
        b = torch.zeros((len(weights), NB * 32), dtype=torch.float32, device=frs[0].device)
        for i, x in enumerate(biases):
            if x is not None:
                b[i, :N] = x.detach().float()
    return torch.stack(frs).contiguous(), b, N


def proj_rows_x3(x: torch.Tensor, wfrag: torch.Tensor, bias: Optional[torch.Tensor], n: int, out: torch.Tensor):
    """out[b, g, m, :n] = x[b, g, m, :] W_g^T + bias_g (siu3r_proj_rows_x3): x [B, G, M, K] fp32 contiguous, out [B, G, M, n] fp32, dense rows,
    one stride between the (b, g) items (a [B, G] block of the model's raw-Gaussian buffer, or with G == 1 one view of every batch item)."""
    _gpu(x, wfrag, bias, out)
    B, G, M, K = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and wfrag.shape[0] == G and wfrag.dtype == torch.bfloat16
    assert out.shape == (B, G, M, n) and out.dtype == torch.float32 and out.stride(3) == 1 and out.stride(2) >= n
    zs = out.stride(1) if G > 1 else out.stride(0)
    assert zs >= M * out.stride(2) and (B == 1 or G == 1 or out.stride(0) == G * zs), (out.shape, out.stride())
    check(_lib.lib().siu3r_proj_rows_x3(_p(x), _p(wfrag), _p(bias), _p(out), B * G, G, M, K, n, out.stride(2), zs, _stream()))
    return out


def proj_rows_ok(K: int, n: int) -> bool:
    return (K == 256 and 65 <= n <= 96) or (K == 128 and n <= 32)


def image_channels(split: bool) -> int:
    """channels per pixel of the packed input image: the bf16x3 convolutions gather 16-byte chunks = 4 fp32, so RGB + one zero
    channel halves the K extent of the 7x7 / 3x3 stems against the 8 channels the bf16 gather (8 bf16 per chunk) needs"""
    return 4 if split else 8


def _batch_strided(t: torch.Tensor, inner: int) -> int:
    """element stride between the batch items of a [N, ..., C] tensor whose items are dense (`inner` elements each)"""
    dense = list(torch.empty(t.shape[1:], device="meta").stride())
    assert list(t.stride()[1:]) == dense and (t.shape[0] == 1 or t.stride(0) >= inner), (t.shape, t.stride())
    return t.stride(0) if t.shape[0] > 1 else inner


def resize_bilinear(x: torch.Tensor, size, align_corners: bool, *, addend=None, ch_scale=None, ch_shift=None,
                    out_dtype=None):
    """x [N,IH,IW,C] / addend [N,OH,OW,C]: dense, or views whose batch items are dense but lie further apart (token maps cut out of a
    longer per-item sequence are read in place)."""
    _gpu(x, addend)
    N, IH, IW, Cc = x.shape
    OH, OW = size
    out = torch.empty((N, OH, OW, Cc), dtype=out_dtype or x.dtype, device=x.device)
    x_bs = _batch_strided(x, IH * IW * Cc)
    a_bs = OH * OW * Cc
    if addend is not None:
        assert addend.shape == out.shape
        a_bs = _batch_strided(addend, OH * OW * Cc)
    check(_lib.lib().siu3r_resize_bilinear_strided(_p(x), _dt(x), _p(out), _dt(out), _p(addend), _dt(addend) if addend is not None else F32,
                                                   _p(ch_scale), _p(ch_shift), N, IH, IW, OH, OW, Cc, int(align_corners), x_bs, a_bs, _stream()))
    return out


def affine_add(x: torch.Tensor, addend, ch_scale, ch_shift, out_dtype=None):
    """y = x * scale[c] + shift[c] (+ addend); x / addend [N, ..., C] dense or batch-strided views (see resize_bilinear)."""
    _gpu(x, addend)
    Cc = x.shape[-1]
    rows_pb = x[0].numel() // Cc
    x_bs = _batch_strided(x, rows_pb * Cc)
    a_bs = rows_pb * Cc
    if addend is not None:
        assert addend.numel() == x.numel()
        a_bs = _batch_strided(addend.view(x.shape) if addend.is_contiguous() else addend, rows_pb * Cc)
    out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    check(_lib.lib().siu3r_affine_add_strided(_p(x), _dt(x), _p(addend), _dt(addend) if addend is not None else F32, _p(out), _dt(out), _p(ch_scale),
                                              _p(ch_shift), x.shape[0] * rows_pb, Cc, rows_pb, x_bs, a_bs, _stream()))
    return out


def maxpool3x3s2(x: torch.Tensor):
    _gpu(x)
    assert x.is_contiguous()
    N, IH, IW, Cc = x.shape
    out = torch.empty((N, (IH - 1) // 2 + 1, (IW - 1) // 2 + 1, Cc), dtype=x.dtype, device=x.device)
    check(_lib.lib().siu3r_maxpool3x3s2(_p(x), _p(out), _dt(x), N, IH, IW, Cc, _stream()))
    return out


def maxpool2x2s2(x: torch.Tensor):
    """NHWC 2 x 2 stride-2 max pool (floor): the pooling of the LPIPS VGG16 stack"""
    _gpu(x)
    assert x.is_contiguous()
    N, IH, IW, Cc = x.shape
    out = torch.empty((N, IH // 2, IW // 2, Cc), dtype=x.dtype, device=x.device)
    check(_lib.lib().siu3r_maxpool2x2s2(_p(x), _p(out), _dt(x), N, IH, IW, Cc, _stream()))
    return out


def lpips_layer(f0: torch.Tensor, f1: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """f0, f1 [..., C] fp32 channel-last feature maps of the two images, w [C] -> per-pixel weighted squared distance of the unit-normalised
    feature vectors, shape f0.shape[:-1] (siu3r_lpips_layer)."""
    _gpu(f0, f1, w)
    assert f0.shape == f1.shape and f0.is_contiguous() and f1.is_contiguous() and w.is_contiguous()
    assert f0.dtype == f1.dtype == w.dtype == torch.float32 and w.numel() == f0.shape[-1]
    dist = torch.empty(f0.shape[:-1], dtype=torch.float32, device=f0.device)
    check(_lib.lib().siu3r_lpips_layer(_p(f0), _p(f1), _p(w), _p(dist), dist.numel(), f0.shape[-1], float(eps), _stream()))
    return dist


def dwconv3x3_gelu(x: torch.Tensor, w9c: torch.Tensor, bias: torch.Tensor, H: int, W: int):
    _gpu(x)
    assert x.is_contiguous()
    B, N, Cc = x.shape
    assert N == 21 * (H * W // 4)
    out = torch.empty_like(x)
    check(_lib.lib().siu3r_dwconv3x3_gelu(_p(x), _p(out), _dt(x), _p(w9c), _p(bias), B, H, W, Cc, _stream()))
    return out


def msdeform_sample(value: torch.Tensor, offs_aw: torch.Tensor, ref: torch.Tensor, shapes: Sequence[Sequence[int]],
                    heads: int, points: int, out_dtype):
    """value [B,S,heads*d]; offs_aw fp32 [B,Q,heads*L*P*3]; ref fp32 [Q,L,2] -> [B,Q,heads*d]."""
    _gpu(value, offs_aw, ref)
    assert value.is_contiguous() and offs_aw.is_contiguous() and ref.is_contiguous() and offs_aw.dtype == torch.float32
    B, S, Cc = value.shape
    Q = offs_aw.shape[1]
    L = len(shapes)
    d = Cc // heads
    assert offs_aw.shape[2] == heads * L * points * 3 and ref.shape == (Q, L, 2)
    out = torch.empty((B, Q, Cc), dtype=out_dtype, device=value.device)
    arr = (C.c_int32 * (2 * L))(*[int(v) for s in shapes for v in s])
    check(_lib.lib().siu3r_msdeform_sample(_p(value), _dt(value), _p(offs_aw), _p(ref), arr, _p(out), _dt(out), B, S, Q,
                                           heads, d, L, points, _stream()))
    return out


def groupnorm(x: torch.Tensor, gamma, beta, groups=32, eps=1e-5, *, relu=False, addend=None, out_dtype=None):
    _gpu(x, addend)
    assert x.is_contiguous()
    N, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * Cc)
    out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    ws = torch.empty((N, groups, 2), dtype=torch.float32, device=x.device)
    check(_lib.lib().siu3r_groupnorm(_p(x), _dt(x), _p(out), _dt(out), _p(gamma), _p(beta), _p(ws), _p(addend),
                                     _dt(addend) if addend is not None else F32, int(relu), N, HW, Cc, groups, eps,
                                     _stream()))
    return out


def pts3d_exp_(xyz: torch.Tensor):
    _gpu(xyz)
    assert xyz.dtype == torch.float32 and xyz.is_contiguous() and xyz.shape[-1] == 3
    check(_lib.lib().siu3r_pts3d_exp(_p(xyz), xyz.numel() // 3, _stream()))
    return xyz


def gaussian_adapter(raw: torch.Tensor):
    """raw [..., 83] -> dict of fp32 tensors (reference gaussian_adapter.py:81-110)."""
    _gpu(raw)
    assert raw.is_contiguous() and raw.shape[-1] == 83
    lead = raw.shape[:-1]
    n = raw.numel() // 83
    dev = raw.device
    f = lambda *s: torch.empty((*lead, *s), dtype=torch.float32, device=dev)
    op, sc, rot, sh, cov = f(), f(3), f(4), f(3, 25), f(3, 3)
    check(_lib.lib().siu3r_gaussian_adapter(_p(raw), _dt(raw), _p(op), _p(sc), _p(rot), _p(sh), _p(cov), n, _stream()))
    return dict(opacities=op, scales=sc, rotations=rot, harmonics=sh, covariances=cov)


def m2f_attn_mask(mask_logits: torch.Tensor, size):
    """mask logits [B,T,IH,IW,Q] fp32 -> uint8 [B,Q,ld] with ld = T*OH*OW rounded up to 64 (1 = blocked, fully blocked
    rows cleared; the padding bytes are never read as keys)."""
    _gpu(mask_logits)
    assert mask_logits.is_contiguous() and mask_logits.dtype == torch.float32
    B, T, IH, IW, Q = mask_logits.shape
    OH, OW = size
    ld = ((T * OH * OW + 63) // 64) * 64
    out = torch.empty((B, Q, ld), dtype=torch.uint8, device=mask_logits.device)
    ws = torch.empty((B * Q,), dtype=torch.int32, device=mask_logits.device)
    check(_lib.lib().siu3r_m2f_attn_mask(_p(mask_logits), _p(out), _p(ws), B, T, IH, IW, OH, OW, Q, ld, _stream()))
    return out


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float):
    """Drop-in for the reference's pybind `curope.rope_2d` (curope.cpp:49-65): tokens [B,N,H,D] in place."""
    if tokens.dim() != 4:
        raise RuntimeError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise RuntimeError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise RuntimeError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise RuntimeError("seq_length differs between tokens & positions")
    if positions.size(2) != 2:
        raise RuntimeError("positions.shape[2] must be equal to 2")
    if tokens.is_cuda != positions.is_cuda:
        raise RuntimeError("tokens and positions are not on the same device")
    _gpu(tokens)
    B, N, H, D = tokens.shape
    if not (tokens.stride(3) == 1 and tokens.stride(2) == D):
        raise RuntimeError("tokens are not contiguous")  # kernels.cu:91
    if not positions.is_contiguous():
        raise RuntimeError("positions are not contiguous")
    if positions.dtype != torch.int64:
        raise RuntimeError("positions must be int64")
    code = {torch.float16: 2, torch.float64: 3}.get(tokens.dtype)  # the seam also takes the reference's half / double (kernels.cu:101)
    check(_lib.lib().siu3r_rope2d(_p(tokens), code if code is not None else _dt(tokens), B, N, H, D, tokens.stride(0), tokens.stride(1),
                                  tokens.stride(2), _p(positions), float(base), float(fwd), _stream()))
    return None
