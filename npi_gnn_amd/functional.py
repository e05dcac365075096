"""Functional form of the conv hot path on MI355X: thin wrappers over the C ABI plus the
``torch.autograd.Function``s that give ``loss.backward()`` (reference
``src/train_with_twoDataset.PY:54``) the same gradients PyG 1.4.2's autograd graph produces.

Formulas (PyG 1.4.2):
  SAGEConv : out = mean_{j in N(i) U {i}} x_j @ W + b          (aggregate at F_in, then project)
  GCNConv  : out = sum_e norm_e (x @ W)[src e] + b,  norm_e = d^-1/2[src] w_e d^-1/2[dst],
             d = weighted out-degree incl. the self loop      (project first, aggregate at F_out)
"""
from __future__ import annotations

import collections
import weakref
from typing import Optional

import torch

from ._lib import (NPI_BF16, NPI_F32, NPI_GEMM_A_ZERO_PADDED, NPI_GEMM_EXACT_F32, NPI_GEMM_RESERVE_CUS, NPI_GEMM_SPLIT_F16X2,
                   NPI_GEMM_WORKSPACE_PREPARED, NPI_PREPARE_F16X2, check, load, ptr, require_gpu, stream_ptr)
from .graph import CSRGraph, CSRSide, as_graph
from .schedule import DEFAULT, Schedule


_PROFILE = None     # bench.py sets this to a list to collect (start, end) events per segsum launch
_PROFILE_TAGS = None  # bench.py: dict tag -> list of (start, end) events around the GAT aggregation launches


class _tag_events:
    """HIP events on the launch stream around one aggregation launch, filed under ``tag`` (a no-op unless bench.py asked)"""

    def __init__(self, tag, dev):
        self.rec = _PROFILE_TAGS
        if self.rec is not None:
            self.tag, self.dev = tag, dev
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.rec is not None:
            self.e0.record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            self.e1.record(torch.cuda.current_stream(self.dev))
            self.rec.setdefault(self.tag, []).append((self.e0, self.e1))
        return False

_PROFILE_GEMM = None  # likewise (name, flops, start, end) per projection GEMM, on the stream it is launched on


class _gemm_events:
    """HIP events around one GEMM entry point when bench.py asked for them (a no-op otherwise)."""

    def __init__(self, name, flops, dev):
        self.rec = _PROFILE_GEMM
        if self.rec is not None:
            self.name, self.flops, self.dev = name, flops, dev
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.rec is not None:
            self.e0.record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            self.e1.record(torch.cuda.current_stream(self.dev))
            self.rec.append((self.name, self.flops, self.e0, self.e1))
        return False

# dW = agg^T dOut (MFMA-bound, launched as ONE workgroup per CU so that it leaves wave slots, LDS and
# registers free) runs on a side stream under the dX chain, whose aggregation is HBM-bound: the two then
# share every CU instead of queueing.  Measured at C4: 8.0 -> 7.1 ms per step.  (With the dW grid
# filling the chip twice over, as before, the same overlap gained 1 %.)  Only for graphs large enough
# for the kernels to outlast the stream bookkeeping (Schedule.overlap_streams / overlap_min_rows).
_SIDE_STREAMS = {}


def _overlaps(sch: Schedule, rows: int) -> bool:
    return sch.overlap_streams and rows >= sch.overlap_min_rows


def _side_stream(dev, k: int = 0) -> "torch.cuda.Stream":
    s = _SIDE_STREAMS.get((dev, k))
    if s is None:
        s = torch.cuda.Stream(device=dev)          # (a high-priority stream made no difference)
        _SIDE_STREAMS[(dev, k)] = s
    return s


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


def _fc(t: torch.Tensor, name: str, like: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float32 or bfloat16 storage (bf16: f32 accumulation inside the kernels); all operands alike."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"{name} must be float32 or bfloat16 (got {t.dtype})")
    if like is not None and t.dtype != like.dtype:
        raise TypeError(f"{name} is {t.dtype} but the other operand is {like.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _fcp(t: torch.Tensor, name: str) -> torch.Tensor:
    """as ``_fc``, but a 2-D operand may keep a row pitch (a column block of a wider buffer): the kernels take a leading
    dimension"""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.size(1):
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"{name} must be float32 or bfloat16 (got {t.dtype})")
        return t
    return _fc(t, name)


def _code(t: torch.Tensor) -> int:
    return NPI_BF16 if t.dtype == torch.bfloat16 else NPI_F32


def _check_out(out: torch.Tensor, rows: int, cols: int, like: torch.Tensor, what: str) -> None:
    """``out=`` buffers go straight to a kernel that writes ``rows`` rows of ``cols`` elements: anything else would be an
    out-of-bounds device write, so it is an error here (a slice of a larger buffer with a row pitch is fine)"""
    if (out.dim() != 2 or out.size(0) != rows or out.size(1) != cols or out.dtype != like.dtype or out.stride(1) != 1
            or out.device != like.device):
        raise ValueError(f"{what}: out must be [{rows}, {cols}] {like.dtype} with unit column stride on {like.device} "
                         f"(got {tuple(out.shape)} {out.dtype}, strides {tuple(out.stride())}, {out.device})")


# ---------------------------------------------------------------------------------------------
# raw ops
# ---------------------------------------------------------------------------------------------
def segsum_scales_ok(side: CSRSide, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> bool:
    """can ``segsum(..., scales_out=)`` write the finished rows' power-of-two scales (f32 rows of 256 columns, 16-byte aligned)?"""
    return (x.dtype == torch.float32 and x.size(1) == 256 and side.nnz_max > 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and (out is None or (out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0)))


def segsum(graph: CSRGraph, side: CSRSide, x: torch.Tensor, w: Optional[torch.Tensor] = None,
           mean: bool = False, bias: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, x2: Optional[torch.Tensor] = None,
           scales_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = scale_i * sum_{p in row i} w[p] * x[col[p]] (+ bias): fused gather + segmented
    reduction (``npi_segsum``).  ``x2``: second part of a two-part table -- entries with
    ``col >= x.size(0)`` read ``x2[col - x.size(0)]`` (``npi_segsum_ex``; the sharded layers).
    ``scales_out`` ``[n_rows]`` f32: also the power-of-two scale of every finished row (what ``row_scales(out)`` would compute in
    a pass of its own; ``segsum_scales_ok``) for the fp16 x 2 projection behind the aggregation."""
    dev = require_gpu(x, w, bias, x2)
    x = _fcp(x, "x")
    if bias is not None:
        bias = _fc(bias, "bias", x)
    N, F = side.n_rows, x.size(1)                  # `graph` may be None for a stand-alone (sharded) side
    split = x.size(0)
    if x2 is not None:
        x2 = _fc(x2, "x2", x)
        if x2.size(1) != F:
            raise ValueError("x2 must have the width of x")
        if x2.stride(0) != x.stride(0):
            x2 = x2.contiguous()
            if x2.stride(0) != x.stride(0):
                x = x.contiguous()
    if side.n_cols > split + (x2.size(0) if x2 is not None else 0):
        raise ValueError(f"the table has {split + (x2.size(0) if x2 is not None else 0)} rows, the adjacency indexes {side.n_cols}")
    if out is None:
        out = torch.empty((N, F), dtype=x.dtype, device=dev)
    else:
        _check_out(out, N, F, x, "segsum")
    carry = side.carry(F)
    prof = _PROFILE
    if prof is not None:        # bench.py: HIP events on the launch stream around this launch
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
    if scales_out is not None and (scales_out.dtype != torch.float32 or scales_out.numel() != N or not scales_out.is_contiguous()):
        raise ValueError("segsum: scales_out must be a contiguous float32 vector with one element per output row")
    check(load().npi_segsum_ex(ptr(side.rowptr), ptr(side.col), ptr(side.item_row), side.item, ptr(w), N, side.nnz_max,
                                ptr(x), x.stride(0), ptr(x2), split if x2 is not None else 0, ptr(out), out.stride(0), F,
                                _code(x), 1 if mean else 0, ptr(bias), ptr(carry), ptr(scales_out), stream_ptr(dev)), "npi_segsum")
    if prof is not None:
        ev1.record(torch.cuda.current_stream(dev))
        prof.append((ev0, ev1))
    return out


# Arithmetic of the f32 projection GEMMs: an argument of every call (``flags``: 0 = the split on the 16-bit matrix cores,
# NPI_GEMM_EXACT_F32 = the exact-f32 MFMA kernels), and of every layer (``Schedule.f16x2_min_rows``: from how many rows on the
# layers' GEMMs take two fp16 pieces per operand).  Neither the library (ABI 3 on) nor this module keeps a switch.


def _gflags(sch: Schedule) -> int:
    """the ``flags`` a layer under schedule ``sch`` hands to its projection GEMMs (``Schedule.gemm_exact_f32``)"""
    return NPI_GEMM_EXACT_F32 if sch.gemm_exact_f32 else 0


def _f16x2(sch: Schedule, rows: int, K: int, N: int, dtype) -> bool:
    """does a layer under schedule ``sch`` run its f32 projection GEMMs (contraction K, output width N, ``rows`` rows) on two fp16
    pieces per operand (NPI_GEMM_SPLIT_F16X2: three matrix products per tile pair instead of six, the same f32-rounding-level error)?"""
    return (sch.f16x2_min_rows is not None and not sch.gemm_exact_f32 and rows >= sch.f16x2_min_rows and dtype == torch.float32
            and f16x2_shape(rows, K, N))


def _gemm_workspace(K: int, N: int, dev) -> torch.Tensor:
    """caller-owned scratch for the re-laid weight matrix: the library allocates nothing"""
    return torch.empty(int(load().npi_linear_workspace_bytes(K, N)), dtype=torch.uint8, device=dev)


def _pad128(k: int) -> int:
    return (int(k) + 127) // 128 * 128


def padded_aggregate_buffer(x: torch.Tensor, K: int, rows: int, bf16_ok: bool = False):
    """A zeroed ``[rows, Kp]`` f32 buffer (Kp = K rounded up to 128) whose first K columns the aggregation fills, or None.
    The reference's feature width is 178: with the aggregate kept 256 wide and its pad columns zero, the layer's forward GEMM and
    its weight-gradient GEMM take the matrix-core kernels (``NPI_GEMM_A_ZERO_PADDED``) instead of the guarded ones -- what
    ``InteractionGraph.batch`` does for extracted batches, here for features the caller hands over 178 wide.  Only when the
    padding costs less than half as much again (178 -> 256 yes, 65 -> 128 no).  (bf16 storage has no padded-operand flag: the
    layer pads its weight matrix with zero rows instead, ``_SageConvFn``.)"""
    if (x.dtype not in ((torch.float32, torch.bfloat16) if bf16_ok else (torch.float32,)) or x.size(1) != K or K % 128 == 0
            or 2 * _pad128(K) > 3 * K or rows < 128):
        return None
    return torch.zeros((rows, _pad128(K)), dtype=x.dtype, device=x.device)


def f16x2_shape(M: int, K: int, N: int) -> bool:
    """shapes whose f32 projection GEMMs (contraction K, output width N, M rows) take the matrix-core kernel completely, i.e. where
    ``NPI_GEMM_SPLIT_F16X2`` applies to every output tile"""
    return M >= 128 and K % 32 == 0 and N % 128 == 0


def _check_scales(scales: Optional[torch.Tensor], rows: int, what: str) -> None:
    """the kernels read one scale per row of the left operand: anything else would be an out-of-bounds device read"""
    if scales is not None and (scales.dtype != torch.float32 or scales.dim() != 1 or scales.numel() != rows or not scales.is_contiguous()):
        raise ValueError(f"{what}: the row scales must be a contiguous float32 vector with one element per row ({rows}), "
                         f"got {tuple(scales.shape)} {scales.dtype}")


def row_scales(a: torch.Tensor) -> torch.Tensor:
    """``[M]`` power-of-two scales of the rows of ``a`` (``npi_row_scales``): the ``a_scales`` of ``linear_fwd`` /
    ``linear_bwd_data`` under ``NPI_GEMM_SPLIT_F16X2`` (two fp16 pieces per operand, three matrix products instead of six)"""
    dev = require_gpu(a)
    if a.dtype != torch.float32 or a.dim() != 2 or a.stride(1) != 1:
        raise TypeError("row_scales: a float32 matrix with unit column stride")
    out = torch.empty(a.size(0), dtype=torch.float32, device=dev)
    check(load().npi_row_scales(ptr(a), a.stride(0), a.size(0), a.size(1), ptr(out), stream_ptr(dev)), "npi_row_scales")
    return out


class Planes:
    """A re-laid copy of a weight matrix for the matrix-core GEMMs (``npi_linear_prepare``) and the arithmetic it was laid out for:
    three bf16 planes (``f16`` False; bf16 storage: one k-block-major copy) or the two fp16 planes + column scales of
    ``NPI_GEMM_SPLIT_F16X2`` calls.  The library cannot tell the two layouts apart, so the tag travels with the buffer and
    ``linear_fwd`` / ``linear_bwd_data`` refuse a copy of the other kind."""
    __slots__ = ("buf", "f16")

    def __init__(self, buf: torch.Tensor, f16: bool = False):
        self.buf, self.f16 = buf, bool(f16)


def _planes_for(ws: Optional["Planes"], f16: bool, what: str) -> torch.Tensor:
    if not isinstance(ws, Planes):
        raise TypeError(f"{what}: ws must come from prepare_weight (a functional.Planes)")
    if ws.f16 != f16:
        raise ValueError(f"{what}: the prepared weight copy holds {'fp16 x 2' if ws.f16 else 'bf16 x 3'} planes, this call runs on "
                         f"{'fp16 x 2 (row scales given)' if f16 else 'bf16 x 3 (no row scales)'}: prepare_weight(..., f16={f16})")
    return ws.buf


def prepare_weight(weight: torch.Tensor, backward: bool = True, f16: bool = False):
    """The re-laid copies of ``weight [K, N]`` the matrix-core GEMMs read (three bf16 planes for f32, a k-block-major copy for
    bf16; ``f16``: the two fp16 planes + column scales of ``NPI_GEMM_SPLIT_F16X2`` calls) for ``linear_fwd(..., ws=)`` and --
    ``backward`` -- ``linear_bwd_data(..., ws=)``, written by ONE launch
    (``npi_linear_prepare``) instead of one in front of every GEMM: ``(ws_fwd, ws_bwd or None)``, or ``(None, None)`` when the
    shape does not take those kernels anyway.  Valid while ``weight`` is unchanged (a layer's forward and its backward)."""
    K, N = weight.shape
    if (weight.dtype not in (torch.float32, torch.bfloat16) or K % 32 or N % 32 or weight.stride(1) != 1
            or not weight.is_cuda or (f16 and weight.dtype != torch.float32)):
        return None, None
    lib = load()
    dev = weight.device
    one = int(lib.npi_linear_workspace_bytes(K, N))
    ws = torch.empty(one * (2 if backward else 1), dtype=torch.uint8, device=dev)
    check(lib.npi_linear_prepare(ptr(weight), weight.stride(0), K, N, (3 if backward else 1) | (NPI_PREPARE_F16X2 if f16 else 0),
                                 _code(weight), ptr(ws), ws.numel(), stream_ptr(dev)), "npi_linear_prepare")
    return Planes(ws[:one], f16), (Planes(ws[one:], f16) if backward else None)


def linear_fwd(a: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
               rowscale: Optional[torch.Tensor] = None, relu: bool = False, flags: Optional[int] = None,
               out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None, reserve_cus: int = 0,
               a_scales: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``a @ weight + bias``.  ``a`` may be wider than ``weight`` has rows: ``[M, Kp]`` with Kp = K rounded up to 128 and
    the columns K.. ZERO (``NPI_GEMM_A_ZERO_PADDED``: the matrix-core kernel on Kp instead of the guarded one on an odd K).
    ``out``: write into this ``[M, N]`` tensor (rows may have a pitch; same dtype) instead of a new one.  ``ws``: the forward
    copy of ``prepare_weight(weight)`` -- no preparation launch in front of the GEMM.  ``reserve_cus``: leave that many CUs (a
    multiple of 8) to a kernel that runs beside the GEMM (``NPI_GEMM_RESERVE_CUS``; the sharded layers, ``Schedule.gemm_reserve_cus``).
    ``a_scales``: ``row_scales(a)`` -- the GEMM then runs on two fp16 pieces per operand (``NPI_GEMM_SPLIT_F16X2``: half the matrix
    work, the same f32-level accuracy); a ``ws`` handed over with it must come from ``prepare_weight(..., f16=True)``."""
    dev = require_gpu(a, weight, bias, rowscale, a_scales)
    a = _fc(a, "a")
    weight = _fc(weight, "weight", a)
    if bias is not None:
        bias = _fc(bias, "bias", a)
    M, Ka = a.shape
    K, N = weight.shape
    fl = int(flags or 0) | NPI_GEMM_RESERVE_CUS(reserve_cus)
    if Ka != K:
        if Ka != _pad128(K) or a.dtype != torch.float32:
            raise ValueError(f"a has {Ka} columns, weight {K} rows (a zero-padded a must be f32 and {_pad128(K)} wide)")
        fl |= NPI_GEMM_A_ZERO_PADDED
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=dev)
    elif out.shape != (M, N) or out.dtype != a.dtype or out.stride(1) != 1 or out.device != a.device:
        raise ValueError(f"linear_fwd: out must be [{M}, {N}] {a.dtype} with unit column stride on the operands' device")
    _check_scales(a_scales, M, "linear_fwd")
    if a_scales is not None and (a.dtype != torch.float32 or Ka != K or fl & NPI_GEMM_EXACT_F32):
        a_scales = None                                         # (storage / flags the fp16 x 2 kernel does not serve)
    if a_scales is not None:
        fl |= NPI_GEMM_SPLIT_F16X2
    if ws is not None and Ka == K:
        ws = _planes_for(ws, a_scales is not None, "linear_fwd")
        fl |= NPI_GEMM_WORKSPACE_PREPARED
    else:
        ws = _gemm_workspace(Ka, N, dev)
    with _gemm_events("fwd", 2.0 * M * K * N, dev):
        check(load().npi_linear_fwd_ex(ptr(a), a.stride(0), ptr(weight), weight.stride(0), ptr(bias), ptr(rowscale),
                                        ptr(out), out.stride(0), M, K, N, 1 if relu else 0, _code(a),
                                        fl, ptr(ws), ws.numel(), ptr(a_scales), stream_ptr(dev)),
              "npi_linear_fwd")
    return out


def linear_bwd_data(dc: torch.Tensor, weight: torch.Tensor,
                    rowscale: Optional[torch.Tensor] = None, flags: Optional[int] = None,
                    out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None, reserve_cus: int = 0,
                    dc_scales: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``rowscale * (dc @ weight.T)``; ``out``: write into this ``[M, K]`` tensor (a row block of a larger buffer); ``ws``: the
    backward copy of ``prepare_weight(weight)``; ``dc_scales``: ``row_scales(dc)`` -- the fp16 x 2 arithmetic, as in ``linear_fwd``."""
    dev = require_gpu(dc, weight, rowscale, dc_scales)
    dc = _fc(dc, "dC")
    weight = _fc(weight, "weight", dc)
    M, N = dc.shape
    K = weight.size(0)
    if out is None:
        da = torch.empty((M, K), dtype=dc.dtype, device=dev)
    else:
        if out.shape != (M, K) or out.dtype != dc.dtype or out.stride(1) != 1 or out.device != dc.device:
            raise ValueError(f"linear_bwd_data: out must be [{M}, {K}] {dc.dtype} with unit column stride on the operands' device")
        da = out
    fl = int(flags or 0) | NPI_GEMM_RESERVE_CUS(reserve_cus)
    _check_scales(dc_scales, M, "linear_bwd_data")
    if dc_scales is not None and (dc.dtype != torch.float32 or fl & NPI_GEMM_EXACT_F32):
        dc_scales = None
    if dc_scales is not None:
        fl |= NPI_GEMM_SPLIT_F16X2
    if ws is not None:
        ws = _planes_for(ws, dc_scales is not None, "linear_bwd_data")
        fl |= NPI_GEMM_WORKSPACE_PREPARED
    else:
        ws = _gemm_workspace(K, N, dev)
    with _gemm_events("bwd_data", 2.0 * M * K * N, dev):
        check(load().npi_linear_bwd_data_ex(ptr(dc), dc.stride(0), ptr(weight), weight.stride(0), ptr(rowscale),
                                             ptr(da), da.stride(0), M, K, N, _code(dc),
                                             fl, ptr(ws), ws.numel(), ptr(dc_scales), stream_ptr(dev)),
              "npi_linear_bwd_data")
    return da


def linear_fwd_scores_ok(a: torch.Tensor, weight: torch.Tensor) -> bool:
    """can ``linear_fwd_scores`` take these operands (f32, aligned, one column tile of the split kernel covers the output)?"""
    return (a.dtype == torch.float32 and weight.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1
            and weight.stride(1) == 1 and a.stride(0) % 4 == 0 and weight.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0
            and weight.data_ptr() % 16 == 0
            and bool(load().npi_linear_fwd_scores_supported(a.size(0), weight.size(0), weight.size(1))))


def linear_fwd_scores(a: torch.Tensor, weight: torch.Tensor, att2: torch.Tensor, a_scales: Optional[torch.Tensor] = None):
    """``(h, a_dst, a_src)``: ``h = a @ weight`` and the row dots ``h @ att[:N]``, ``h @ att[N:]`` taken from the accumulators in
    the GEMM's store epilogue (``npi_linear_fwd_scores``; one head, ``att2`` holds ``2 N`` values)."""
    dev = require_gpu(a, weight, att2)
    M, K = a.shape
    N = weight.size(1)
    att2 = _f32c(att2.reshape(-1), "att")
    if att2.numel() != 2 * N:
        raise ValueError("linear_fwd_scores: att must hold 2 N values")
    _check_scales(a_scales, M, "linear_fwd_scores")
    h = torch.empty((M, N), dtype=torch.float32, device=dev)
    a_dst = torch.empty((M, 1), dtype=torch.float32, device=dev)
    a_src = torch.empty((M, 1), dtype=torch.float32, device=dev)
    ws = _gemm_workspace(K, N, dev)
    with _gemm_events("fwd", 2.0 * M * K * N, dev):
        check(load().npi_linear_fwd_scores(ptr(a), a.stride(0), ptr(weight), weight.stride(0), ptr(att2), ptr(h), h.stride(0),
                                               ptr(a_dst), ptr(a_src), M, K, N, ptr(ws), ws.numel(), ptr(a_scales), stream_ptr(dev)),
              "npi_linear_fwd_scores")
    return h, a_dst, a_src


def linear_bwd_data_rank2_ok(dc: torch.Tensor, weight: torch.Tensor) -> bool:
    """can ``linear_bwd_data_rank2`` take these operands (f32, aligned, a shape the split kernel covers completely)?"""
    M, N = dc.shape
    return (dc.dtype == torch.float32 and weight.dtype == torch.float32 and dc.stride(1) == 1 and weight.stride(1) == 1
            and dc.stride(0) % 4 == 0 and weight.stride(0) % 4 == 0 and dc.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0
            and bool(load().npi_linear_bwd_data_rank2_supported(M, weight.size(0), N)))


def linear_bwd_data_rank2(dc: torch.Tensor, weight: torch.Tensor, row0: torch.Tensor, row1: torch.Tensor,
                          col0: torch.Tensor, col1: torch.Tensor, dc_scales: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``dc @ weight.T + row0 (x) col0 + row1 (x) col1`` with the rank-2 term added in the GEMM's store epilogue
    (``npi_linear_bwd_data_rank2``; ``row*`` are ``[M]``, ``col*`` ``[K]``)."""
    dev = require_gpu(dc, weight, row0, row1, col0, col1)
    M, N = dc.shape
    K = weight.size(0)
    if row0.numel() != M or row1.numel() != M or col0.numel() != K or col1.numel() != K:
        raise ValueError("linear_bwd_data_rank2: row vectors must have M entries, column vectors K")
    row0, row1, col0, col1 = (_f32c(t.reshape(-1), "rank-2 vector") for t in (row0, row1, col0, col1))
    _check_scales(dc_scales, M, "linear_bwd_data_rank2")
    da = torch.empty((M, K), dtype=torch.float32, device=dev)
    ws = _gemm_workspace(K, N, dev)
    with _gemm_events("bwd_data", 2.0 * M * K * N, dev):
        check(load().npi_linear_bwd_data_rank2(ptr(dc), dc.stride(0), ptr(weight), weight.stride(0), ptr(row0), ptr(row1),
                                               ptr(col0), ptr(col1), ptr(da), da.stride(0), M, K, N, ptr(ws), ws.numel(), ptr(dc_scales),
                                               stream_ptr(dev)), "npi_linear_bwd_data_rank2")
    return da


def gat_rank2_cols(weight: torch.Tensor, att2: torch.Tensor) -> torch.Tensor:
    """``U [2, K] = [att_dst ; att_src] @ weight.T`` (one head; ``att2`` is ``[1, 2C]`` or ``[2, C]``): the column vectors of
    ``linear_bwd_data_rank2`` (``npi_gat_rank2_cols``)"""
    dev = require_gpu(weight, att2)
    weight, att2 = _f32c(weight, "weight"), _f32c(att2.reshape(2, -1), "att")
    K, C = weight.shape
    U = torch.empty((2, K), dtype=torch.float32, device=dev)
    check(load().npi_gat_rank2_cols(ptr(weight), weight.stride(0), ptr(att2), K, C, ptr(U), stream_ptr(dev)), "npi_gat_rank2_cols")
    return U


def gat_rank2_tail(P: torch.Tensor, weight: torch.Tensor, att2: torch.Tensor, dw: Optional[torch.Tensor], want_datt: bool):
    """``dw += P.T @ att`` in place (``dw`` None: skipped) and, when asked, ``datt [2, C] = P @ weight`` -- one launch
    (``npi_gat_rank2_tail``); P is ``[2, K]`` = x^T [g_dst g_src]"""
    dev = require_gpu(P, weight, att2, dw)
    P, weight, att2 = _f32c(P, "P"), _f32c(weight, "weight"), _f32c(att2.reshape(2, -1), "att")
    K, C = weight.shape
    if dw is not None and (dw.shape != (K, C) or dw.dtype != torch.float32 or dw.stride(1) != 1):
        raise ValueError("gat_rank2_tail: dw must be a [K, C] float32 tensor with unit column stride")
    datt = torch.empty((2, C), dtype=torch.float32, device=dev) if want_datt else None
    if dw is None and datt is None:
        return None
    check(load().npi_gat_rank2_tail(ptr(P), ptr(weight), weight.stride(0), ptr(att2), K, C, ptr(dw),
                                    dw.stride(0) if dw is not None else 0, ptr(datt), stream_ptr(dev)), "npi_gat_rank2_tail")
    return datt


#: id(row-scale tensor) -> (weak reference to it, (its version, columns), its column scales): at most 8 (LRU)
_COL_SCALES_KEPT: "collections.OrderedDict" = collections.OrderedDict()


def col_scales(a: Optional[torch.Tensor] = None, row_scales: Optional[torch.Tensor] = None, cols: Optional[int] = None) -> torch.Tensor:
    """``[K]`` power-of-two COLUMN scales for ``linear_bwd_weight(a_cs=, dc_cs=)`` (``npi_col_scales``): from the column maxima
    of ``a`` (one pass over it -- for a matrix that does not change between steps, once), or -- ``a`` None -- the smallest of the
    ``row_scales`` its producer wrote, for each of ``cols`` columns alike (no pass over the matrix)."""
    lib = load()
    if a is not None:
        dev = require_gpu(a)
        if a.dtype != torch.float32 or a.dim() != 2 or a.stride(1) != 1:
            raise TypeError("col_scales: a float32 matrix with unit column stride")
        M, K = a.shape
    else:
        if row_scales is None or cols is None:
            raise ValueError("col_scales: a matrix, or its row scales and its column count")
        dev = require_gpu(row_scales)
        _check_scales(row_scales, row_scales.numel(), "col_scales")
        M, K = int(row_scales.numel()), int(cols)
        # the scales of a matrix that does not change between steps (a layer's input features) are the same tensor object every
        # step: its column scales are kept beside a WEAK reference to it (an id reused by another tensor misses; recomputed when
        # it was written in place since) -- no attribute on the tensor, no strong reference
        kept = _COL_SCALES_KEPT.get(id(row_scales))
        if kept is not None and kept[0]() is row_scales and kept[1] == (row_scales._version, K):
            _COL_SCALES_KEPT.move_to_end(id(row_scales))
            return kept[2]
    n_ws = int(lib.npi_col_scales_workspace_elems(M, K))
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    out = torch.empty(K, dtype=torch.float32, device=dev)
    check(lib.npi_col_scales(ptr(a), a.stride(0) if a is not None else 0, M, K, ptr(row_scales) if a is None else 0, ptr(out), ptr(ws),
                             n_ws, stream_ptr(dev)), "npi_col_scales")
    if a is None and not torch.cuda.is_current_stream_capturing():
        _COL_SCALES_KEPT[id(row_scales)] = (weakref.ref(row_scales), (row_scales._version, K), out)
        _COL_SCALES_KEPT.move_to_end(id(row_scales))
        while len(_COL_SCALES_KEPT) > 8:
            _COL_SCALES_KEPT.popitem(last=False)
    return out


def dw_f16x2_shape(M: int, K: int, N: int) -> bool:
    """shapes whose f32 weight-gradient GEMM takes the fp16 x 2 matrix-core kernel (``linear_bwd_weight(a_cs=, dc_cs=)``)"""
    return M >= 4096 and K % 128 == 0 and N % 128 == 0


def linear_bwd_weight(a: torch.Tensor, dc: torch.Tensor, want_bias: bool = True, shared: bool = False,
                      flags: Optional[int] = None, k_valid: Optional[int] = None, a_cs: Optional[torch.Tensor] = None,
                      dc_cs: Optional[torch.Tensor] = None):
    """``shared``: the GEMM will run beside an HBM-bound kernel on another stream (smaller grid; a per-call argument of
    ``npi_linear_bwd_weight_ex``, no process-wide switch is touched).  ``k_valid``: ``a`` is ``[M, Kp]`` with only the
    first ``k_valid`` columns data and the rest ZERO (see ``linear_fwd``); dW then has ``k_valid`` rows.
    ``a_cs`` / ``dc_cs`` (both or none; ``col_scales``): the column scales of the two operands -- the GEMM then runs on two fp16
    pieces per operand (``NPI_GEMM_SPLIT_F16X2``: three matrix products per tile instead of six; ``dw_f16x2_shape``)."""
    dev = require_gpu(a, dc)
    a = _fc(a, "a")
    dc = _fc(dc, "dC", a)
    M, Ka = a.shape
    N = dc.size(1)
    K = Ka if k_valid is None else int(k_valid)
    fl = int(flags or 0)
    if K != Ka:
        if Ka != _pad128(K) or a.dtype != torch.float32:
            raise ValueError(f"a zero-padded a must be f32 and {_pad128(K)} wide (got {Ka})")
        fl |= NPI_GEMM_A_ZERO_PADDED
    if (a_cs is None) != (dc_cs is None):
        raise ValueError("linear_bwd_weight: a_cs and dc_cs come together")
    if a_cs is not None and (a.dtype != torch.float32 or K != Ka or fl & NPI_GEMM_EXACT_F32 or not dw_f16x2_shape(M, K, N)):
        a_cs = dc_cs = None                                     # (storage / shape / flags the fp16 x 2 kernel does not serve)
    if a_cs is not None:
        for v, n, what in ((a_cs, K, "a_cs"), (dc_cs, N, "dc_cs")):
            if v.dtype != torch.float32 or v.numel() != n or not v.is_contiguous() or v.device != a.device:
                raise ValueError(f"linear_bwd_weight: {what} must be a contiguous float32 vector with {n} entries on the operands' device")
        fl |= NPI_GEMM_SPLIT_F16X2
    lib = load()
    n_ws = int(lib.npi_linear_bwd_weight_workspace_elems(M, Ka, N))
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    dw = torch.empty((K, N), dtype=a.dtype, device=dev)
    db = torch.empty(N, dtype=a.dtype, device=dev) if want_bias else None
    with _gemm_events("bwd_weight", 2.0 * M * K * N, dev):
        check(lib.npi_linear_bwd_weight_ex(ptr(a), a.stride(0), ptr(dc), dc.stride(0), ptr(dw), dw.stride(0), ptr(db),
                                           M, K, N, ptr(ws), n_ws, _code(a), fl,
                                           1 if shared else 0, ptr(a_cs), ptr(dc_cs), stream_ptr(dev)),
              "npi_linear_bwd_weight")
    return dw, db


def _layer_calls_ok() -> bool:
    """the one-call-per-layer entry points (npi_conv_fwd / npi_conv_bwd) issue the same launches as the per-op calls; the per-op
    calls are kept while bench.py's per-launch event hooks are on (the hooks sit around the single launches)"""
    return _PROFILE is None and _PROFILE_GEMM is None and _PROFILE_TAGS is None


def conv_fwd(side: CSRSide, x: torch.Tensor, w_entry: Optional[torch.Tensor], mean: bool, weight: torch.Tensor,
             bias: Optional[torch.Tensor], relu: bool, want_bwd_copy: bool):
    """aggregate-then-project as ONE call (``npi_conv_fwd``): ``(agg, out, ws_bwd or None)``.  ``agg`` is ``[N, Ka]`` with Ka = the
    weight's row count, or that rounded up to 128 with zero pad columns (``padded_aggregate_buffer``).  The same launches as
    ``segsum`` + ``prepare_weight`` + ``linear_fwd``."""
    dev = require_gpu(x, weight, bias, w_entry)
    x = _fcp(x, "x")
    weight = _fc(weight, "weight", x)
    if bias is not None:
        bias = _fc(bias, "bias", x)
    N, F = side.n_rows, x.size(1)
    K, Nout = weight.shape
    if side.n_cols > x.size(0):
        raise ValueError(f"the table has {x.size(0)} rows, the adjacency indexes {side.n_cols}")
    agg = padded_aggregate_buffer(x, K, N)
    if agg is None:
        agg = torch.empty((N, F), dtype=x.dtype, device=dev)
    Ka = agg.size(1)
    if Ka != K and (x.dtype != torch.float32 or Ka != _pad128(K) or F not in (K, Ka)):
        # (F == Ka: x is itself the zero-padded base of the caller's features, sage_conv(pad_base=))
        raise ValueError(f"x has {F} columns, weight {K} rows (a zero-padded x / aggregate must be f32 and {_pad128(K)} wide)")
    lib = load()
    flags = NPI_GEMM_A_ZERO_PADDED if Ka != K else 0
    can_prepare = (Ka == K and K % 32 == 0 and Nout % 32 == 0 and weight.stride(1) == 1)
    which = (3 if want_bwd_copy else 1) if can_prepare else 0
    one = int(lib.npi_linear_workspace_bytes(Ka, Nout))
    ws = torch.empty(one * (2 if which == 3 else 1), dtype=torch.uint8, device=dev)
    out = torch.empty((N, Nout), dtype=x.dtype, device=dev)
    check(lib.npi_conv_fwd(ptr(side.rowptr), ptr(side.col), ptr(side.item_row), side.item, ptr(w_entry), N, side.nnz_max, ptr(x),
                           x.stride(0), F, 1 if mean else 0, ptr(agg), agg.stride(0), ptr(side.carry(F)), ptr(weight), weight.stride(0),
                           ptr(bias), ptr(out), out.stride(0), K, Nout, 1 if relu else 0, _code(x), flags, which, ptr(ws), ws.numel(),
                           stream_ptr(dev)), "npi_conv_fwd")
    return agg, out, (Planes(ws[one:], False) if which == 3 else None)


def conv_bwd(tside: CSRSide, grad_out: torch.Tensor, out_relu: Optional[torch.Tensor], agg: torch.Tensor, weight: torch.Tensor,
             rowscale: Optional[torch.Tensor], t_w: Optional[torch.Tensor], ws_bwd: Optional[torch.Tensor], want_x: bool,
             want_w: bool, want_bias: bool):
    """the backward of ``conv_fwd`` as ONE call (``npi_conv_bwd``): ``(dx, dw, db)``.  The same launches as ``relu_backward`` +
    ``linear_bwd_weight`` + ``linear_bwd_data`` + ``segsum`` over the transposed side, on one stream."""
    dev = require_gpu(grad_out, agg, weight, rowscale, t_w)
    lib = load()
    N, Nout = grad_out.shape
    K = weight.size(0)
    Ka = agg.size(1)
    flags = NPI_GEMM_A_ZERO_PADDED if Ka != K else 0
    dz = torch.empty((N, Nout), dtype=torch.float32, device=dev) if out_relu is not None else None
    dw = db = dws = dagg = dx = None
    n_ws = 0
    if want_w:
        n_ws = int(lib.npi_linear_bwd_weight_workspace_elems(N, Ka, Nout))
        dws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        dw = torch.empty((K, Nout), dtype=agg.dtype, device=dev)
        db = torch.empty(Nout, dtype=agg.dtype, device=dev) if want_bias else None
    prepared = ws_bwd is not None
    ws_bwd = _planes_for(ws_bwd, False, "conv_bwd") if prepared else None
    if want_x:
        dagg = torch.empty((N, K), dtype=grad_out.dtype, device=dev)
        dx = torch.empty((N, K), dtype=grad_out.dtype, device=dev)
        if ws_bwd is None:
            ws_bwd = _gemm_workspace(K, Nout, dev)
    check(lib.npi_conv_bwd(ptr(grad_out), grad_out.stride(0), ptr(out_relu), out_relu.stride(0) if out_relu is not None else 0, ptr(dz),
                           Nout, N, K, Nout, _code(grad_out), flags, ptr(agg), agg.stride(0), ptr(dw), Nout, ptr(db), ptr(dws), n_ws,
                           ptr(weight), weight.stride(0), ptr(rowscale), ptr(dagg), K, ptr(ws_bwd),
                           ws_bwd.numel() if ws_bwd is not None else 0, 1 if prepared else 0, ptr(tside.rowptr), ptr(tside.col),
                           ptr(tside.item_row), tside.item, ptr(t_w), tside.nnz_max, ptr(dx), K,
                           ptr(tside.carry(K)) if want_x else 0, stream_ptr(dev)), "npi_conv_bwd")
    return dx, dw, db



class _L2NormalizeFn(torch.autograd.Function):
    """``F.normalize(x, p=2, dim=-1)`` for f32 rows (``npi_l2_normalize_rows`` / ``_bwd``)"""

    @staticmethod
    def forward(ctx, x, eps: float):
        dev = require_gpu(x)
        x = _f32c(x, "x")
        M, F = x.shape
        y = torch.empty((M, F), dtype=torch.float32, device=dev)
        nrm = torch.empty(M, dtype=torch.float32, device=dev)
        check(load().npi_l2_normalize_rows(ptr(x), x.stride(0), M, F, float(eps), ptr(y), y.stride(0), ptr(nrm), stream_ptr(dev)),
              "npi_l2_normalize_rows")
        ctx.eps = float(eps)
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        dev = y.device
        dy = _f32c(dy, "grad_out")
        M, F = y.shape
        dx = torch.empty((M, F), dtype=torch.float32, device=dev)
        check(load().npi_l2_normalize_rows_bwd(ptr(dy), dy.stride(0), ptr(y), y.stride(0), ptr(nrm), M, F, ctx.eps, ptr(dx),
                                               dx.stride(0), stream_ptr(dev)), "npi_l2_normalize_rows_bwd")
        return dx, None


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """``torch.nn.functional.normalize(x, p=2.0, dim=-1)`` (what ``SAGEConv(normalize=True)`` ends with); f32 2-D through the
    package's own kernels, other storage types through the torch op"""
    if x.dtype != torch.float32 or x.dim() != 2:
        return torch.nn.functional.normalize(x, p=2.0, dim=-1, eps=eps)
    return _L2NormalizeFn.apply(x, eps)


def relu_backward(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """dz = dy where y > 0 else 0 (``y``: the output of a fused ReLU); f32 through ``npi_relu_backward``, other storage types
    through the equivalent torch op"""
    if dy.dtype != torch.float32 or y.dtype != torch.float32 or dy.dim() != 2:
        return torch.ops.aten.threshold_backward(dy, y, 0)
    dev = require_gpu(dy, y)
    if dy.stride(1) != 1:
        dy = dy.contiguous()
    if y.stride(1) != 1:
        y = y.contiguous()
    M, F = dy.shape
    dz = torch.empty((M, F), dtype=torch.float32, device=dev)
    check(load().npi_relu_backward(ptr(dy), dy.stride(0), ptr(y), y.stride(0), M, F, ptr(dz), dz.stride(0), stream_ptr(dev)),
          "npi_relu_backward")
    return dz


def colsum(x: torch.Tensor) -> torch.Tensor:
    dev = require_gpu(x)
    x = _f32c(x, "x")
    M, N = x.shape
    n_ws = int(load().npi_colsum_workspace_elems(M, N))
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    out = torch.empty(N, dtype=torch.float32, device=dev)
    check(load().npi_colsum(ptr(x), x.stride(0), M, N, ptr(out), ptr(ws), n_ws, stream_ptr(dev)), "npi_colsum")
    return out


def entry_weights(graph: CSRGraph, edge_weight: Optional[torch.Tensor], fill: float = 1.0):
    """``edge_weight [E]`` in the entry order of both orientations, self loops included
    (``add_remaining_self_loops(edge_index, edge_weight, fill, N)``: an existing self loop's weight becomes that
    node's loop weight, every other node's loop weighs ``fill``).  Returns ``[by_dst, by_src]``."""
    lib = load()
    dev = graph.device
    N = graph.num_nodes
    s = stream_ptr(dev)
    loop_w = None
    if edge_weight is not None:
        require_gpu(edge_weight)
        edge_weight = _f32c(edge_weight.detach(), "edge_weight")
        if edge_weight.numel() != graph.num_edges:
            raise ValueError(f"edge_weight has {edge_weight.numel()} entries, edge_index {graph.num_edges} columns")
    src, dst = graph._src, graph._dst
    m = (src == dst) & (src >= 0)                        # (-1, -1) columns are padding (npi_filter_adj), not self loops
    if bool(m.any()):
        loop_w = torch.full((N,), fill, dtype=torch.float32, device=dev)
        loop_w[src[m]] = edge_weight[m] if edge_weight is not None else 1.0
    out = []
    for side in (graph.by_dst, graph.by_src):
        we = torch.empty(max(side.nnz_max, 1), dtype=torch.float32, device=dev)
        check(lib.npi_entry_weights(ptr(side.eid), ptr(side.rowidx), ptr(side.rowptr), ptr(edge_weight),
                                    ptr(loop_w), fill, N, side.nnz_max, ptr(we), s), "npi_entry_weights")
        out.append(we)
    return out


def mean_bwd_weights(graph: CSRGraph, tside: CSRSide, w_src: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-entry weights of the TRANSPOSED aggregation of a mean layer when it runs on dOut itself: ``scatter_mean``'s divisor
    belongs to the target, i.e. to the GATHERED row of the by-source side -- ``w[p] = inv_count[col[p]]`` (times the entry's edge
    weight, ``w_src``).  ``npi_entry_col_scale``; without edge weights computed once per graph and side (kept on the graph)."""
    cache = graph.__dict__.setdefault("_mean_t_w", {})
    key = id(tside)
    if w_src is None and key in cache:
        return cache[key]
    dev = graph.device
    inv = graph.inv_count(graph.by_dst)
    out = torch.empty(max(tside.nnz_max, 1), dtype=torch.float32, device=dev)
    check(load().npi_entry_col_scale(ptr(tside.col), ptr(tside.rowptr), ptr(inv), ptr(w_src), tside.n_rows, inv.numel(), tside.nnz_max,
                                     ptr(out), stream_ptr(dev)), "npi_entry_col_scale")
    if w_src is None:
        cache[key] = out
    return out


def _aggregate_first_ok(sch: Schedule, f16: bool, weight: torch.Tensor, grad_out: torch.Tensor, tside: CSRSide, ws_bwd) -> bool:
    """can the backward of an aggregate-then-project layer run as ``dX = (A^T dOut) W^T`` with the GEMM on fp16 x 2 -- the
    transposed aggregation on dOut itself, writing the row scales of its output (``_backward_aggregate_first``)?"""
    return (sch.aggregate_first_backward and f16 and isinstance(ws_bwd, Planes) and ws_bwd.f16 and grad_out.dtype == torch.float32 and grad_out.size(1) == 256
            and segsum_scales_ok(tside, grad_out) and f16x2_shape(grad_out.size(0), grad_out.size(1), weight.size(0)))


def _backward_aggregate_first(graph, tside: CSRSide, w_t: Optional[torch.Tensor], agg, weight, grad_out, ws_bwd: "Planes", want_w: bool,
                              has_bias: bool, overlap: bool, k_valid=None):
    """``dX = (A_w^T dOut) W^T`` instead of ``A_w^T (dOut W^T)`` -- the same number up to fp32 rounding (the aggregation is linear;
    GCNConv's forward uses the same freedom).  What it buys on large graphs: the transposed aggregation now WRITES
    the left operand of the backward's data GEMM, so it writes that operand's row scales too (``segsum(scales_out=)``) and the
    GEMM runs on two fp16 pieces per operand -- three matrix products instead of six -- with no pass over dOut for its scales
    (which arrive from outside the layer).  dW = agg^T dOut needs neither and goes first on the launch stream, so that it is
    resident on every CU before the aggregation -- on the side stream -- fills the remaining wave slots (as before).
    ``w_t``: per-entry weights of the transposed side (``mean_bwd_weights`` / the GCN norm).  Returns ``(dx, dw, db)``."""
    dev = grad_out.device
    N, Nout = grad_out.shape
    t = torch.empty((N, Nout), dtype=torch.float32, device=dev)
    t_scales = torch.empty(N, dtype=torch.float32, device=dev)
    dw = db = None
    if overlap and want_w:
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        side.wait_stream(main)                                   # dOut (and the buffers above) are ready for the side stream
        dw, db = linear_bwd_weight(agg, grad_out, want_bias=has_bias, shared=True, k_valid=k_valid)
        with torch.cuda.stream(side):
            segsum(graph, tside, grad_out, w=w_t, mean=False, out=t, scales_out=t_scales)
        for buf in (grad_out, t, t_scales):
            buf.record_stream(side)                              # allocated on main, used on side
        main.wait_stream(side)
    else:
        if want_w:
            dw, db = linear_bwd_weight(agg, grad_out, want_bias=has_bias, k_valid=k_valid)
        segsum(graph, tside, grad_out, w=w_t, mean=False, out=t, scales_out=t_scales)
    dx = linear_bwd_data(t, weight, ws=ws_bwd, dc_scales=t_scales)
    return dx, dw, db


# ---------------------------------------------------------------------------------------------
# SAGEConv
# ---------------------------------------------------------------------------------------------
class _SageConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, graph: CSRGraph, w_entry=None, relu: bool = False, sch: Schedule = DEFAULT):
        ctx.graph = graph
        ctx.w_src = w_entry[1] if w_entry else None
        ctx.has_bias = bias is not None
        ctx.relu = relu
        ctx.sch = sch
        ctx.k_rows = None
        f16 = (x.dtype == weight.dtype and x.size(1) == weight.size(0)
               and _f16x2(sch, graph.by_dst.n_rows, weight.size(0), weight.size(1), x.dtype) and segsum_scales_ok(graph.by_dst, x))
        ctx.f16 = f16
        if f16:
            # large graphs, 256 features: the projection on two fp16 pieces per operand -- the aggregation writes the row scales of
            # agg itself (a wave maximum per finished row); where it cannot, a pass over agg would cost what the GEMM saves
            agg = torch.empty((graph.by_dst.n_rows, x.size(1)), dtype=x.dtype, device=x.device)
            scales = torch.empty(agg.size(0), dtype=torch.float32, device=x.device)
            segsum(graph, graph.by_dst, x, w=w_entry[0] if w_entry else None, mean=True, out=agg, scales_out=scales)
            # both fp16 x 2 copies of W in one call: the backward runs aggregate-first (_backward_aggregate_first), so its data GEMM
            # takes the row scales the transposed aggregation writes -- no pass over dOut, which arrives from outside the layer
            wsf, ctx.ws_bwd = prepare_weight(weight, backward=ctx.needs_input_grad[0], f16=True)
            out = linear_fwd(agg, weight, bias, relu=relu, ws=wsf, a_scales=scales)
            ctx.k_valid = None
            ctx.save_for_backward(agg, weight, *([out] if relu else []))
            return out
        fl = _gflags(sch)                                       # (exact-f32 MFMA kernels on request: the per-op calls carry the flag)
        if not fl and _layer_calls_ok() and x.dtype == weight.dtype and x.dtype in (torch.float32, torch.bfloat16) and not (
                x.dtype == torch.bfloat16 and weight.size(0) % 128 != 0 and 2 * _pad128(weight.size(0)) <= 3 * weight.size(0)):
            # the whole layer call as one entry point (the same launches; one trip through the C ABI instead of three)
            agg, out, ctx.ws_bwd = conv_fwd(graph.by_dst, x, w_entry[0] if w_entry else None, True, weight, bias, relu,
                                            ctx.needs_input_grad[0])
            ctx.k_valid = weight.size(0) if agg.size(1) != weight.size(0) else None
            ctx.save_for_backward(agg, weight, *([out] if relu else []))
            return out
        # a2-a4: gather + scatter_mean (w_entry: PyG's `edge_weight.view(-1, 1) * x_j`; the mean still divides by the count)
        # x may be the zero-padded base of the caller's features (sage_conv): agg then keeps the padded width -- zero
        # columns stay zero under a weighted mean -- and both GEMMs run on it (linear_fwd / linear_bwd_weight)
        agg = None if fl else padded_aggregate_buffer(x, weight.size(0), graph.by_dst.n_rows, bf16_ok=True)
        if agg is None:
            agg = segsum(graph, graph.by_dst, x, w=w_entry[0] if w_entry else None, mean=True)
        else:
            segsum(graph, graph.by_dst, x, w=w_entry[0] if w_entry else None, mean=True, out=agg[:, : x.size(1)])
        if agg.size(1) != weight.size(0) and agg.dtype == torch.bfloat16:
            # bf16 storage, zero-padded aggregate: W gets zero ROWS to match (one small launch) and all three GEMMs of the layer
            # run the aligned bf16 matrix-core kernels; dAgg and dW are computed padded and cut back to the true width
            ctx.k_rows = weight.size(0)
            weight = torch.nn.functional.pad(weight.detach(), (0, 0, 0, agg.size(1) - weight.size(0)))
        # both re-laid copies of W (for this GEMM and for dAgg = dOut W^T of the backward) in one launch
        wsf, ctx.ws_bwd = prepare_weight(weight, backward=ctx.needs_input_grad[0]) if (
            agg.size(1) == weight.size(0) and agg.dtype == weight.dtype and not fl) else (None, None)
        out = linear_fwd(agg, weight, bias, relu=relu, ws=wsf, flags=fl)  # a5: agg @ W + b (ReLU in the epilogue on request)
        ctx.k_valid = weight.size(0) if agg.size(1) != weight.size(0) else None
        ctx.save_for_backward(agg, weight, *([out] if relu else []))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        agg, weight = ctx.saved_tensors[:2]
        graph: CSRGraph = ctx.graph
        grad_out = _fc(grad_out, "grad_out", agg)
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        want_x = ctx.needs_input_grad[0]
        # symmetric edge list, no per-entry weights: A^T has the rows of A (graph.CSRGraph.symmetric) -- skip the second sort
        tside = (lambda: graph.by_dst) if (graph.symmetric and ctx.w_src is None) else (lambda: graph.by_src)
        overlap = want_w and want_x and _overlaps(ctx.sch, grad_out.size(0))
        fl = _gflags(ctx.sch)
        if not fl and not overlap and ctx.k_rows is None and _layer_calls_ok() and (want_w or want_x) and not ctx.f16:
            # one stream: the whole backward as one entry point (ReLU mask, dW + db, dAgg GEMM, transposed aggregation)
            out_relu = None
            if ctx.relu:
                if grad_out.dtype == torch.float32:
                    out_relu = ctx.saved_tensors[2]
                else:
                    grad_out = relu_backward(grad_out, ctx.saved_tensors[2])
            dx, dw, db = conv_bwd(tside() if want_x else graph.by_dst, grad_out, out_relu, agg, weight, graph.inv_count(graph.by_dst),
                                  ctx.w_src, ctx.ws_bwd, want_x, want_w, ctx.has_bias)
            return dx, dw, db, None, None, None, None
        if ctx.relu:                                            # threshold_backward, as F.relu's autograd does it
            grad_out = relu_backward(grad_out, ctx.saved_tensors[2])
        dx = dw = db = None
        if want_x and _aggregate_first_ok(ctx.sch, ctx.f16, weight, grad_out, tside(), ctx.ws_bwd):
            ts = tside()
            dx, dw, db = _backward_aggregate_first(graph, ts, mean_bwd_weights(graph, ts, ctx.w_src), agg, weight, grad_out, ctx.ws_bwd,
                                                   want_w, ctx.has_bias, overlap, ctx.k_valid)
            return dx, dw, db, None, None, None, None
        ws_bwd = ctx.ws_bwd if not (isinstance(ctx.ws_bwd, Planes) and ctx.ws_bwd.f16) else None     # (fp16 x 2 planes: not for this order)
        if want_w and not overlap:
            dw, db = linear_bwd_weight(agg, grad_out, want_bias=ctx.has_bias, k_valid=ctx.k_valid, flags=fl)   # aggT dOut, colsum
        if want_x:
            # dAgg = dOut W^T, pre-divided by the in-count of its row (fused epilogue), then
            # dX[j] = sum over the entries whose SOURCE is j  ==  segsum over the by-source CSR
            dagg = linear_bwd_data(grad_out, weight, rowscale=graph.inv_count(graph.by_dst), ws=ws_bwd, flags=fl)
            if ctx.k_rows is not None:
                dagg = dagg[:, : ctx.k_rows]                                 # (the pad columns of dAgg: dOut times zero rows)
            if overlap:
                # dW is independent of the dX chain.  It is launched on THIS stream right behind dAgg's GEMM, one
                # workgroup per CU, so that it is resident everywhere before the aggregation -- sent to a second HIP
                # stream -- fills the remaining wave slots; the two then share every CU.  (The other way round the
                # aggregation wins the race, takes every register of every SIMD, and dW only starts when it is over.)
                dev = grad_out.device
                main = torch.cuda.current_stream(dev)
                side = _side_stream(dev)
                side.wait_stream(main)                               # dAgg is complete for the side stream
                dw, db = linear_bwd_weight(agg, grad_out, want_bias=ctx.has_bias, shared=True, k_valid=ctx.k_valid, flags=fl)
                with torch.cuda.stream(side):
                    dx = segsum(graph, tside(), dagg, w=ctx.w_src, mean=False)
                dagg.record_stream(side)                             # allocated on main, read on side
                dx.record_stream(main)                               # allocated on side, consumed on main
                main.wait_stream(side)
            else:
                dx = segsum(graph, tside(), dagg, w=ctx.w_src, mean=False)
        if dw is not None and ctx.k_rows is not None:
            dw = dw[: ctx.k_rows]                                             # the gradient of the zero rows is not W's
        return dx, dw, db, None, None, None, None


class _SageConcatFn(torch.autograd.Function):
    """``SAGEConv(concat=True)`` (PyG 1.4.2): ``out = [x | mean_{j -> i} x_j] @ W[2F, Fo] + b`` over the edge list AS IT IS (no
    self loop appended).  The mean is written straight into the right half of the ``[N, 2F]`` operand of ONE projection GEMM
    (K = 2F); the backward splits ``dOut W^T`` into its two halves: the left one is x's own share, the right one -- divided by
    the in-count -- goes through the transposed aggregation."""

    @staticmethod
    def forward(ctx, x, weight, bias, graph: CSRGraph, w_entry=None):
        N, F = x.shape
        cat = torch.empty((N, 2 * F), dtype=x.dtype, device=x.device)
        cat[:, :F].copy_(x)
        segsum(graph, graph.by_dst, x, w=w_entry[0] if w_entry else None, mean=True, out=cat[:, F:])
        out = linear_fwd(cat, weight, bias)
        ctx.graph, ctx.F = graph, F
        ctx.w_src = w_entry[1] if w_entry else None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(cat, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        cat, weight = ctx.saved_tensors
        graph: CSRGraph = ctx.graph
        F = ctx.F
        grad_out = _fc(grad_out, "grad_out", cat)
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = linear_bwd_weight(cat, grad_out, want_bias=ctx.has_bias)
        if ctx.needs_input_grad[0]:
            dcat = linear_bwd_data(grad_out, weight)                              # [N, 2F]
            dagg = dcat[:, F:] * graph.inv_count(graph.by_dst).view(-1, 1).to(dcat.dtype)     # scatter_mean's divisor, per TARGET
            dx = segsum(graph, graph.by_src, dagg.contiguous(), w=ctx.w_src, mean=False)
            dx += dcat[:, :F]
        return dx, dw, db, None, None


def sage_conv(x: torch.Tensor, edge_index, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
              normalize: bool = False, edge_weight: Optional[torch.Tensor] = None, relu: bool = False,
              pad_base: Optional[torch.Tensor] = None, schedule: Schedule = DEFAULT, concat: bool = False) -> torch.Tensor:
    """PyG 1.4.2 ``SAGEConv(normalize=False, concat=False).forward`` on MI355X
    (call sites: reference ``src/classes.py:62,66,70``).  ``edge_weight [E]`` scales the messages
    (no gradient flows to it, as in the reference's use of the layer).  ``pad_base``: the wider buffer ``x`` is the leading
    columns of, its other columns zero (``GraphBatch.pad_base``)."""
    require_gpu(x, weight, bias)
    if edge_weight is not None and edge_weight.requires_grad:
        raise NotImplementedError("gradients w.r.t. edge_weight are not implemented")      # as gcn_conv: never silently detached
    if concat:
        # PyG 1.4.2: ``concat=True`` skips add_remaining_self_loops -- the edge list as it is, weight [2 F_in, F_out]
        if isinstance(edge_index, CSRGraph):
            graph = edge_index
            if graph.self_loops or not graph.keep_equal:
                raise ValueError("sage_conv(concat=True) aggregates over the edge list as it is: build the graph with "
                                 "CSRGraph(edge_index, N, self_loops=False, keep_equal=True)")
            if graph.num_nodes != x.size(0):
                raise ValueError(f"CSRGraph was built for {graph.num_nodes} nodes, x has {x.size(0)}")
        else:
            ei = edge_index.edge_index if hasattr(edge_index, "edge_index") else edge_index      # a GraphBatch or the [2, E] tensor
            graph = CSRGraph(ei, x.size(0), self_loops=False, keep_equal=True)
        if weight.size(0) != 2 * x.size(1):
            raise ValueError(f"sage_conv(concat=True): weight must have {2 * x.size(1)} rows (got {weight.size(0)})")
        w_entry = entry_weights(graph, edge_weight, 1.0) if edge_weight is not None else None
        out = _SageConcatFn.apply(x, weight, bias, graph, w_entry)
        if relu:
            out = torch.relu(out)
        return l2_normalize(out) if normalize else out
    graph = as_graph(edge_index, x.size(0))
    w_entry = entry_weights(graph, edge_weight, 1.0) if edge_weight is not None else None
    # features that are a view of a wider buffer whose extra columns are zero (InteractionGraph.batch: 178 -> 256): the
    # layer runs on the padded buffer, so that its GEMMs take the matrix-core kernels instead of the guarded ones
    base = pad_base
    if (base is not None and not x.requires_grad and x.dtype == torch.float32 and base.size(0) == x.size(0)
            and base.size(1) == _pad128(x.size(1)) and weight.size(0) == x.size(1) and base.data_ptr() == x.data_ptr()):
        x = base
    if relu and normalize:
        raise ValueError("sage_conv: relu=True applies to the projection's output; normalize=True comes after it in PyG")
    out = _SageConvFn.apply(x, weight, bias, graph, w_entry, relu, schedule)
    if normalize:
        out = l2_normalize(out)
    return out


# ---------------------------------------------------------------------------------------------
# GCNConv
# ---------------------------------------------------------------------------------------------
class GCNNorm:
    """Per-entry symmetric normalisation for both orientations (``GCNConv.norm``), cacheable."""

    def __init__(self, graph: CSRGraph, edge_weight: Optional[torch.Tensor] = None, improved: bool = False):
        lib = load()
        dev = graph.device
        N = graph.num_nodes
        fill = 2.0 if improved else 1.0
        s = stream_ptr(dev)
        sides = (graph.by_dst, graph.by_src)
        w_entry = entry_weights(graph, edge_weight, fill) if (edge_weight is not None or improved) else [None, None]
        deg = None
        if w_entry[1] is not None:      # weighted degree over SOURCE rows
            deg = torch.empty(N, dtype=torch.float32, device=dev)
            check(lib.npi_row_weight_sum(ptr(graph.by_src.rowptr), ptr(w_entry[1]), N, ptr(deg), s),
                  "npi_row_weight_sum")
        self.norm = []
        for k, side in enumerate(sides):
            nrm = torch.empty(max(side.nnz_max, 1), dtype=torch.float32, device=dev)
            check(lib.npi_gcn_norm(ptr(side.rowidx), ptr(side.col), ptr(side.rowptr), ptr(w_entry[k]), ptr(deg),
                                   ptr(graph.by_src.rowptr), N, side.nnz_max, ptr(nrm), s), "npi_gcn_norm")
            self.norm.append(nrm)
        self.graph = graph

    @property
    def by_dst(self) -> torch.Tensor:
        return self.norm[0]

    @property
    def by_src(self) -> torch.Tensor:
        return self.norm[1]


class PlainWeights:
    """``GCNConv(normalize=False)`` (PyG 1.4.2: ``norm = edge_weight``): the per-entry weights of both orientations over the edge
    list AS IT IS -- no self loop appended, existing ones ordinary entries -- in the shape the GCN functions take (``.graph``,
    ``.by_dst``, ``.by_src``; None = unweighted)."""

    def __init__(self, edge_index, num_nodes: int, edge_weight: Optional[torch.Tensor] = None):
        if isinstance(edge_index, CSRGraph):
            if edge_index.self_loops or not edge_index.keep_equal:
                raise ValueError("GCNConv(normalize=False) aggregates over the edge list as it is: build the graph with "
                                 "CSRGraph(edge_index, N, self_loops=False, keep_equal=True)")
            self.graph = edge_index
        else:
            ei = edge_index.edge_index if hasattr(edge_index, "edge_index") else edge_index
            self.graph = CSRGraph(ei, num_nodes, self_loops=False, keep_equal=True)
        w = entry_weights(self.graph, edge_weight, 1.0) if edge_weight is not None else [None, None]
        self.by_dst, self.by_src = w


class _GcnConvFn(torch.autograd.Function):
    """PyG's literal order: project, then aggregate at width F_out (used when F_in > F_out, e.g. C1's 178 -> 64)."""

    @staticmethod
    def forward(ctx, x, weight, bias, norm: GCNNorm):
        graph = norm.graph
        xw = linear_fwd(x, weight)                                            # project first
        out = segsum(graph, graph.by_dst, xw, w=norm.by_dst, bias=bias)       # sum_e norm_e xw[src] + b
        ctx.norm = norm
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        norm: GCNNorm = ctx.norm
        graph = norm.graph
        grad_out = _fc(grad_out, "grad_out", x)
        dx = dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # the column-sum kernel is f32: bf16 storage (BASELINE.json configs[1] style training) widens dOut for it
            db = colsum(grad_out) if grad_out.dtype == torch.float32 else colsum(grad_out.float()).to(grad_out.dtype)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dxw = segsum(graph, graph.by_src, grad_out, w=norm.by_src)        # A_hat^T dOut
            if ctx.needs_input_grad[1]:
                dw, _ = linear_bwd_weight(x, dxw, want_bias=False)
            if ctx.needs_input_grad[0]:
                dx = linear_bwd_data(dxw, weight)
        return dx, dw, db, None


class _GcnAggFirstFn(torch.autograd.Function):
    """The same GCNConv evaluated as ``(A_hat x) W + b`` instead of ``A_hat (x W) + b`` -- identical up to fp32 rounding.
    Aggregating FIRST (at width F_in <= F_out) gives the layer SAGEConv's schedule: the aggregate is saved, so the
    weight gradient ``agg^T dOut`` (with ``db`` fused) no longer waits for the backward aggregation and runs on the matrix
    cores UNDER it (second stream), and the aggregations run at the narrower width (C3's first layer: 178 instead of 256)."""

    @staticmethod
    def forward(ctx, x, weight, bias, norm: GCNNorm, sch: Schedule = DEFAULT):
        graph = norm.graph
        ctx.norm = norm
        ctx.has_bias = bias is not None
        ctx.sch = sch
        if (x.dtype == weight.dtype and x.size(1) == weight.size(0) and segsum_scales_ok(graph.by_dst, x)
                and _f16x2(sch, graph.by_dst.n_rows, weight.size(0), weight.size(1), x.dtype)):
            # large graphs, 256 features: the projection on two fp16 pieces per operand, the row scales written by the aggregation
            # (as in _SageConvFn.forward; the backward aggregates first as well)
            agg = torch.empty((graph.by_dst.n_rows, x.size(1)), dtype=x.dtype, device=x.device)
            scales = torch.empty(agg.size(0), dtype=torch.float32, device=x.device)
            segsum(graph, graph.by_dst, x, w=norm.by_dst, out=agg, scales_out=scales)
            wsf, ctx.ws_bwd = prepare_weight(weight, backward=ctx.needs_input_grad[0], f16=True)
            out = linear_fwd(agg, weight, bias, ws=wsf, a_scales=scales)
            ctx.k_valid = None
            ctx.f16 = True
            ctx.save_for_backward(agg, weight)
            return out
        ctx.f16 = False
        fl = _gflags(sch)
        if not fl and _layer_calls_ok() and x.dtype == weight.dtype == torch.float32:
            agg, out, ctx.ws_bwd = conv_fwd(graph.by_dst, x, norm.by_dst, False, weight, bias, False, ctx.needs_input_grad[0])
            ctx.k_valid = weight.size(0) if agg.size(1) != weight.size(0) else None
            ctx.save_for_backward(agg, weight)
            return out
        agg = None if fl else padded_aggregate_buffer(x, weight.size(0), graph.by_dst.n_rows)
        if agg is None:
            agg = segsum(graph, graph.by_dst, x, w=norm.by_dst)                # sum_e norm_e x[src]
        else:
            segsum(graph, graph.by_dst, x, w=norm.by_dst, out=agg[:, : x.size(1)])
        wsf, ctx.ws_bwd = prepare_weight(weight, backward=ctx.needs_input_grad[0]) if (
            agg.size(1) == weight.size(0) and agg.dtype == weight.dtype and not fl) else (None, None)
        out = linear_fwd(agg, weight, bias, ws=wsf, flags=fl)
        ctx.k_valid = weight.size(0) if agg.size(1) != weight.size(0) else None
        ctx.save_for_backward(agg, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        agg, weight = ctx.saved_tensors
        norm: GCNNorm = ctx.norm
        graph = norm.graph
        grad_out = _fc(grad_out, "grad_out", agg)
        dx = dw = db = None
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        want_x = ctx.needs_input_grad[0]
        overlap = want_w and want_x and _overlaps(ctx.sch, grad_out.size(0))
        fl = _gflags(ctx.sch)
        if not fl and not overlap and _layer_calls_ok() and (want_w or want_x) and grad_out.dtype == torch.float32 and not ctx.f16:
            dx, dw, db = conv_bwd(graph.by_src if want_x else graph.by_dst, grad_out, None, agg, weight, None, norm.by_src,
                                  ctx.ws_bwd, want_x, want_w, ctx.has_bias)
            return dx, dw, db, None, None
        if want_x and _aggregate_first_ok(ctx.sch, ctx.f16, weight, grad_out, graph.by_src, ctx.ws_bwd):
            dx, dw, db = _backward_aggregate_first(graph, graph.by_src, norm.by_src, agg, weight, grad_out, ctx.ws_bwd, want_w,
                                                   ctx.has_bias, overlap, ctx.k_valid)
            return dx, dw, db, None, None
        ws_bwd = ctx.ws_bwd if not (isinstance(ctx.ws_bwd, Planes) and ctx.ws_bwd.f16) else None
        if want_w and not overlap:
            dw, db = linear_bwd_weight(agg, grad_out, want_bias=ctx.has_bias, k_valid=ctx.k_valid, flags=fl)
        if want_x:
            dagg = linear_bwd_data(grad_out, weight, ws=ws_bwd, flags=fl)
            if overlap:                                                        # see _SageConvFn.backward
                dev = grad_out.device
                main = torch.cuda.current_stream(dev)
                side = _side_stream(dev)
                side.wait_stream(main)
                dw, db = linear_bwd_weight(agg, grad_out, want_bias=ctx.has_bias, shared=True, k_valid=ctx.k_valid, flags=fl)
                with torch.cuda.stream(side):
                    dx = segsum(graph, graph.by_src, dagg, w=norm.by_src)
                dagg.record_stream(side)
                dx.record_stream(main)
                main.wait_stream(side)
            else:
                dx = segsum(graph, graph.by_src, dagg, w=norm.by_src)
        return dx, dw, db, None, None


def gcn_conv(x: torch.Tensor, edge_index, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
             edge_weight: Optional[torch.Tensor] = None, improved: bool = False,
             norm: Optional[GCNNorm] = None, schedule: Schedule = DEFAULT, normalize: bool = True) -> torch.Tensor:
    """PyG 1.4.2 ``GCNConv.forward`` (normalize=True) on MI355X.  Evaluated as ``(A_hat x) W + b`` when the input is not wider
    than the output (``_GcnAggFirstFn``: the aggregation at the narrower width, dW under the backward aggregation), in PyG's
    literal order ``A_hat (x W) + b`` otherwise -- the same number up to fp32 rounding."""
    require_gpu(x, weight, bias)
    if edge_weight is not None and edge_weight.requires_grad:
        raise NotImplementedError("gradients w.r.t. edge_weight are not implemented")
    if norm is None:
        norm = GCNNorm(as_graph(edge_index, x.size(0)), edge_weight, improved) if normalize else \
            PlainWeights(edge_index, x.size(0), edge_weight)            # normalize=False: norm = edge_weight, no self loops
    if weight.size(0) <= weight.size(1):
        return _GcnAggFirstFn.apply(x, weight, bias, norm, schedule)
    return _GcnConvFn.apply(x, weight, bias, norm)


# ---------------------------------------------------------------------------------------------
# GATConv
# ---------------------------------------------------------------------------------------------
def _transpose_map(graph: CSRGraph) -> torch.Tensor:
    """by-target position of every by-source entry (same directed edge; loop -> loop), cached."""
    tm = getattr(graph, "_tmap", None)
    if tm is None:
        lib = load()
        dev, N, E = graph.device, graph.num_nodes, graph.num_edges
        d, s_ = graph.by_dst, graph.by_src
        st = stream_ptr(dev)
        pos = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        check(lib.npi_edge_positions(ptr(d.eid), ptr(d.rowptr), N, d.nnz_max, E, ptr(pos), st), "npi_edge_positions")
        tm = torch.empty(max(s_.nnz_max, 1), dtype=torch.int32, device=dev)
        check(lib.npi_entry_transpose_map(ptr(s_.eid), ptr(s_.rowidx), ptr(s_.rowptr), ptr(d.rowptr), ptr(pos), N,
                                          s_.nnz_max, ptr(tm), st), "npi_entry_transpose_map")
        graph._tmap = tm
    return tm


def _inverse_transpose_map(graph: CSRGraph) -> torch.Tensor:
    """by-source position of every by-target entry (inverse of ``_transpose_map``), cached; index work in torch, once per graph"""
    inv = getattr(graph, "_tmap_inv", None)
    if inv is None:
        tm = _transpose_map(graph)
        n = tm.numel()
        idx = torch.arange(n, device=tm.device)
        valid = idx < graph.by_src.rowptr[-1]                       # entries past nnz (dropped edges / padding) hold garbage
        inv = torch.zeros(n + 1, dtype=torch.int32, device=tm.device)
        inv.scatter_(0, torch.where(valid, tm.long(), torch.full_like(idx, n)), idx.to(torch.int32))
        inv = inv[:n].contiguous()
        graph._tmap_inv = inv
    return inv


def gat_fused_shape(H: int, C: int) -> bool:
    """Shapes the fused backward serves (``npi_gat_backward_fused_heads``): ONE gather pass over the by-source entries yields
    the aggregation half of d hfeat AND every entry's score gradient -- one head of <= 256 channels, or 2 / 4 / 8 heads of
    32 / 64 / 128 channels with H C <= 256.  Everything else takes ``npi_gat_edge_grad`` (a gather pass over the by-target
    entries) followed by the by-source aggregation."""
    return C % 4 == 0 and H * C <= 256 and (H == 1 or (H in (2, 4, 8) and C in (32, 64, 128)))


def _gat_aggregate(graph, side, x, H, C, a_dst, a_src, m, s, slope, by_source, bias=None, g_dst=None,
                   g_src=None, att=None, alpha=None, alpha_map=None, x2=None, out=None):
    """``x2``: second part of a two-part table (see ``segsum``); ``out``: write into this ``[n_rows, H C]`` buffer."""
    dev = x.device
    x = _f32c(x, "x")
    if x2 is not None:
        x2 = _f32c(x2, "x2")
        if x2.stride(0) != x.stride(0):
            raise ValueError("the two parts of the table must share the row pitch")
    if out is None:
        out = torch.empty((side.n_rows, H * C), dtype=torch.float32, device=dev)
    else:
        _check_out(out, side.n_rows, H * C, x, "gat_aggregate")
    check(load().npi_gat_aggregate_ex(ptr(side.rowptr), ptr(side.col), ptr(side.item_row), side.item, side.n_rows, side.nnz_max,
                                      ptr(x), x.stride(0), ptr(x2), x.size(0) if x2 is not None else 0,
                                      ptr(out), out.stride(0), H, C, ptr(a_dst), ptr(a_src),
                                      ptr(m), ptr(s), float(slope), 1 if by_source else 0, ptr(bias), ptr(g_dst),
                                      ptr(g_src), ptr(att), ptr(alpha), ptr(alpha_map), ptr(side.carry(H * C)),
                                      stream_ptr(dev)), "npi_gat_aggregate")
    return out


def gat_scores(hfeat, att2, H, C):
    """a_dst[i,h] = <hfeat[i,h,:], att[h,:C]>, a_src[i,h] = <hfeat[i,h,:], att[h,C:]>"""
    dev = hfeat.device
    N = hfeat.size(0)
    a_dst = torch.empty((N, H), dtype=torch.float32, device=dev)
    a_src = torch.empty((N, H), dtype=torch.float32, device=dev)
    check(load().npi_gat_scores(ptr(hfeat), hfeat.stride(0), ptr(att2), N, H, C, ptr(a_dst), ptr(a_src), stream_ptr(dev)),
          "npi_gat_scores")
    return a_dst, a_src


def gat_softmax_stats(side: CSRSide, a_row, a_col, H, slope, want_scores: bool = False):
    """(m, s) [n_rows, H]: row max and sum of exp(. - max) of leaky_relu(a_row[row] + a_col[col]) over the entries of
    every row of ``side`` (an empty row gets m = 0, s = 0); item-parallel segmented scan (``npi_gat_softmax_stats_ex``).
    ``want_scores``: also the score of every entry ``[nnz_max, H]`` (entry order), for ``npi_gat_aggregate_scores`` --
    returns (m, s, scores)."""
    lib = load()
    dev = a_row.device
    m = torch.empty((side.n_rows, H), dtype=torch.float32, device=dev)
    s = torch.empty((side.n_rows, H), dtype=torch.float32, device=dev)
    n_ws = int(lib.npi_seg_scan_workspace_elems(side.nnz_max, H))
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    e = torch.empty((max(side.nnz_max, 1), H), dtype=torch.float32, device=dev) if want_scores else None
    check(lib.npi_gat_softmax_stats_ex(ptr(side.rowptr), ptr(side.col), ptr(side.rowidx), ptr(a_row), ptr(a_col),
                                       side.n_rows, side.nnz_max, H, float(slope), ptr(m), ptr(s), ptr(e), ptr(ws), n_ws,
                                       stream_ptr(dev)), "npi_gat_softmax_stats_ex")
    return (m, s, e) if want_scores else (m, s)


def gat_rowdot(a, b, bias, H, C):
    """D[i,h] = <a[i,h,:], b[i,h,:] - bias[h,:]>"""
    dev = a.device
    N = a.size(0)
    D = torch.empty((N, H), dtype=torch.float32, device=dev)
    check(load().npi_gat_rowdot(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(bias), N, H, C, ptr(D), stream_ptr(dev)),
          "npi_gat_rowdot")
    return D


def gat_rowdot_colsum(a, b, bias, H, C, want_colsum: bool = True, relu_mask: bool = False):
    """(D, colsum(a) or None): ``gat_rowdot`` and the bias gradient in one pass over ``a`` (= dOut) and ``b`` (= out);
    falls back on the two separate kernels for unaligned rows or H C > 1024.  ``relu_mask``: ``b`` is the output of a fused
    ReLU -- the gradient is masked first (``a' = a`` where ``b > 0``), D and the column sums use ``a'``, and ``a'`` is returned
    as a third value (the gradient of the pre-activation, which the rest of the backward consumes)."""
    lib = load()
    dev = a.device
    N = a.size(0)
    ok = (C % 4 == 0 and H * C <= 1024 and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0
          and b.data_ptr() % 16 == 0 and (bias is None or bias.data_ptr() % 16 == 0))
    if not ok:
        if relu_mask:
            a = relu_backward(a, b)
        res = (gat_rowdot(a, b, bias, H, C), (colsum(a) if want_colsum else None))
        return res + (a,) if relu_mask else res
    D = torch.empty((N, H), dtype=torch.float32, device=dev)
    cs = torch.empty(H * C, dtype=torch.float32, device=dev) if want_colsum else None
    am = torch.empty((N, H * C), dtype=torch.float32, device=dev) if relu_mask else None
    n_ws = int(lib.npi_gat_rowdot_colsum_workspace_elems(N, H, C)) if want_colsum else 0
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dev)
    check(lib.npi_gat_rowdot_colsum_relu(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(bias), N, H, C, ptr(D), ptr(cs), ptr(am),
                                         am.stride(0) if am is not None else 0, ptr(ws), n_ws, stream_ptr(dev)),
          "npi_gat_rowdot_colsum")
    return (D, cs, am) if relu_mask else (D, cs)


def gat_aggregate_scores(side: CSRSide, table, table2, C, scores, m, s, bias=None, out=None, relu: bool = False):
    """One head: out[r] = sum_p exp(scores[p] - m[r]) / (s[r] + 1e-16) table[col p] (+ bias) over the entries of row r, the
    per-entry scores coming from ``gat_softmax_stats(..., want_scores=True)``; ``table2``: second part of a two-part table."""
    dev = table.device
    table = _f32c(table, "table")
    if table2 is not None:
        table2 = _f32c(table2, "table2")
        if table2.stride(0) != table.stride(0):
            raise ValueError("the two parts of the table must share the row pitch")
    if out is None:
        out = torch.empty((side.n_rows, C), dtype=torch.float32, device=dev)
    else:
        _check_out(out, side.n_rows, C, table, "gat_aggregate_scores")
    if side.nnz_max == 0:
        out = out.zero_() if bias is None else out.copy_(bias.view(1, -1).expand_as(out))
        return out.clamp_(min=0) if relu else out
    with _tag_events("gat_fwd_aggregate", dev):
        check(load().npi_gat_aggregate_scores(ptr(side.rowptr), ptr(side.col), ptr(side.item_row), side.item, side.n_rows,
                                              side.nnz_max, ptr(table), table.stride(0), ptr(table2),
                                              table.size(0) if table2 is not None else 0, ptr(out), out.stride(0), C, ptr(scores),
                                              ptr(m), ptr(s), ptr(bias), 1 if relu else 0, ptr(side.carry(C)), stream_ptr(dev)),
              "npi_gat_aggregate_scores")
    return out


def gat_aggregate_fused(side: CSRSide, table, table2, C, a_dst, att2, slope, bias=None, out=None, relu: bool = False,
                        scales_out: Optional[torch.Tensor] = None):
    """One head of at most 256 channels: ``(out, m, s)`` -- the forward aggregation together with the softmax statistics of every
    row, in ONE launch (``npi_gat_aggregate_fused``): an online softmax whose scores' source half is recomputed from the gathered
    rows (``att2``: the layer's ``[1, 2C]`` attention vector) -- no statistics pass, no per-entry score array, no gather of
    ``a_src``.  ``a_dst`` ``[n_rows]``; ``table2``: second part of a two-part table.  ``scales_out`` ``[n_rows]`` (256 channels): also
    the power-of-two scale of every stored row (bias and ReLU applied) -- the ``x_scales`` of the next layer's projection."""
    dev = table.device
    table = _f32c(table, "table")
    if table2 is not None:
        table2 = _f32c(table2, "table2")
        if table2.stride(0) != table.stride(0):
            raise ValueError("the two parts of the table must share the row pitch")
    if out is None:
        out = torch.empty((side.n_rows, C), dtype=torch.float32, device=dev)
    else:
        _check_out(out, side.n_rows, C, table, "gat_aggregate_fused")
    m = torch.empty((side.n_rows, 1), dtype=torch.float32, device=dev)
    s = torch.empty((side.n_rows, 1), dtype=torch.float32, device=dev)
    with _tag_events("gat_fwd_aggregate", dev):
        check(load().npi_gat_aggregate_fused(ptr(side.rowptr), ptr(side.col), ptr(side.rowidx), ptr(side.item_row), side.item,
                                                 side.n_rows, side.nnz_max, ptr(table), table.stride(0), ptr(table2),
                                                 table.size(0) if table2 is not None else 0, ptr(out), out.stride(0), C,
                                                 ptr(a_dst.contiguous()), ptr(_f32c(att2.reshape(-1), "att")), float(slope), ptr(bias),
                                                 1 if relu else 0, ptr(m), ptr(s), ptr(side.carry(C)), ptr(scales_out), stream_ptr(dev)),
              "npi_gat_aggregate_fused")
    return out, m, s


def gat_pack_targets(a_dst, m, s, D, out=None):
    """[n, 4] = (a_dst, m, 1 / (s + 1e-16), D) of every target node (one head): what ``gat_backward_fused_packed`` gathers.
    ``out``: a contiguous ``[n, 4]`` f32 tensor to fill (a slice of a larger table)."""
    dev = a_dst.device
    n = a_dst.numel()
    if out is not None and (out.shape != (n, 4) or out.dtype != torch.float32 or not out.is_contiguous()):
        raise ValueError(f"gat_pack_targets: out must be a contiguous [{n}, 4] float32 tensor")
    t = out if out is not None else torch.empty((n, 4), dtype=torch.float32, device=dev)
    check(load().npi_gat_pack_targets(ptr(a_dst.contiguous()), ptr(m.contiguous()), ptr(s.contiguous()), ptr(D.contiguous()), n,
                                      ptr(t), stream_ptr(dev)), "npi_gat_pack_targets")
    return t


def gat_backward_fused_packed(side: CSRSide, dout, dout2, hrow, C, tpack, a_src_rows, slope, out=None, H: int = 1,
                              scales_out: Optional[torch.Tensor] = None, rowsum_out: Optional[torch.Tensor] = None):
    """By-SOURCE side: (out [n_rows, H C] = sum_q alpha_q dout[col q], dz [nnz_max, H] per entry and head) in one gather pass;
    ``dout2``: second part of the gathered table; ``tpack`` [n_cols H, 4] indexed by (column id, head); ``hrow`` /
    ``a_src_rows``: features and source scores of the ROW nodes.  H in {1, 2, 4, 8} (npi_gat_backward_fused_heads).
    ``scales_out`` ``[n_rows]`` (H C == 256): also the power-of-two scale of every row of ``out``, for the fp16 x 2 GEMM behind it.
    ``rowsum_out`` ``[n_rows]`` (one head): also the row sums of dz -- ``seg_rowsum(side, dz, 1)`` -- from the lanes that compute dz."""
    dev = dout.device
    dout = _f32c(dout, "dout")
    if dout2 is not None:
        dout2 = _f32c(dout2, "dout2")
        if dout2.stride(0) != dout.stride(0):
            raise ValueError("the two parts of the table must share the row pitch")
    hrow = _f32c(hrow, "hrow")
    if out is None:
        out = torch.empty((side.n_rows, H * C), dtype=torch.float32, device=dev)
    else:
        _check_out(out, side.n_rows, H * C, dout, "gat_backward_fused_packed")
    dz = torch.empty(max(side.nnz_max, 1) * H, dtype=torch.float32, device=dev)
    if rowsum_out is not None and (H != 1 or rowsum_out.numel() != side.n_rows or rowsum_out.dtype != torch.float32
                                   or not rowsum_out.is_contiguous()):
        raise ValueError("gat_backward_fused_packed: rowsum_out must be a contiguous float32 [n_rows] tensor (one head)")
    if side.nnz_max == 0:
        if scales_out is not None:
            scales_out.fill_(1.0)
        if rowsum_out is not None:
            rowsum_out.zero_()
        return out.zero_(), dz
    n_ws = int(load().npi_seg_scan_workspace_elems(side.nnz_max, 1)) if rowsum_out is not None else 0
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev) if n_ws else None
    with _tag_events("gat_bwd_fused", dev):
        check(load().npi_gat_backward_fused_heads(ptr(side.rowptr), ptr(side.col), ptr(side.rowidx), ptr(side.item_row), side.item,
                                                      side.n_rows, side.nnz_max, ptr(dout), dout.stride(0), ptr(dout2),
                                                      dout.size(0) if dout2 is not None else 0, ptr(hrow), hrow.stride(0), ptr(out),
                                                      out.stride(0), H, C, ptr(tpack), ptr(a_src_rows.contiguous()), float(slope), ptr(dz),
                                                      ptr(side.carry(H * C)), ptr(scales_out), ptr(rowsum_out), ptr(ws), n_ws,
                                                      stream_ptr(dev)),
              "npi_gat_backward_fused_heads")
    return out, dz


def gat_rank1_add(dh, g_dst, g_src, att2, H, C):
    """dh[j, h, :] += g_dst[j, h] att[h, :C] + g_src[j, h] att[h, C:] in place"""
    check(load().npi_gat_rank1_add(ptr(dh), dh.stride(0), ptr(g_dst), ptr(g_src), ptr(att2), dh.size(0), H, C,
                                   stream_ptr(dh.device)), "npi_gat_rank1_add")
    return dh


def gat_edge_grad(side: CSRSide, col_feat, col_feat2, row_feat, H, C, a_dst, a_src, m, s, D, slope, swap,
                  alpha_out=None):
    """dz per entry of ``side`` (``npi_gat_edge_grad_ex``).  swap = 0: rows are targets (row_feat = dOut rows; a_dst,
    m, s, D by row; a_src by column; col_feat = gathered hfeat).  swap = 1: rows are sources (row_feat = their hfeat;
    a_src by row; a_dst, m, s, D by column; col_feat = gathered dOut)."""
    dev = row_feat.device
    dz = torch.empty((max(side.nnz_max, 1), H), dtype=torch.float32, device=dev)
    col_feat = _f32c(col_feat, "col_feat")
    if col_feat2 is not None:
        col_feat2 = _f32c(col_feat2, "col_feat2")
        if col_feat2.stride(0) != col_feat.stride(0):
            raise ValueError("the two parts of the table must share the row pitch")
    check(load().npi_gat_edge_grad_ex(ptr(side.rowptr), ptr(side.col), ptr(side.rowidx), side.n_rows, side.nnz_max,
                                      ptr(col_feat), col_feat.stride(0), ptr(col_feat2),
                                      col_feat.size(0) if col_feat2 is not None else 0,
                                      ptr(row_feat), row_feat.stride(0), H, C, ptr(a_dst), ptr(a_src), ptr(m), ptr(s),
                                      ptr(D), float(slope), int(swap), ptr(dz), ptr(alpha_out), stream_ptr(dev)),
          "npi_gat_edge_grad")
    return dz


def seg_rowsum(side: CSRSide, vals, H, map_=None):
    """out[r, h] = sum over the entries p of row r of vals[map ? map[p] : p, h]"""
    lib = load()
    dev = vals.device
    out = torch.empty((side.n_rows, H), dtype=torch.float32, device=dev)
    n_ws = int(lib.npi_seg_scan_workspace_elems(side.nnz_max, H))
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    check(lib.npi_seg_rowsum_ex(ptr(side.rowptr), ptr(side.rowidx), ptr(vals), ptr(map_), side.n_rows, side.nnz_max, H,
                                ptr(out), ptr(ws), n_ws, stream_ptr(dev)), "npi_seg_rowsum_ex")
    return out


def gat_att_grad(hfeat, g_dst, g_src, H, C):
    lib = load()
    dev = hfeat.device
    N = hfeat.size(0)
    n_ws = int(lib.npi_gat_att_grad_workspace_elems(N, H, C))
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    datt = torch.empty((H, 2 * C), dtype=torch.float32, device=dev)
    check(lib.npi_gat_att_grad(ptr(hfeat), hfeat.stride(0), ptr(g_dst), ptr(g_src), N, H, C, ptr(datt), ptr(ws), n_ws,
                               stream_ptr(dev)), "npi_gat_att_grad")
    return datt


class _GatConvFn(torch.autograd.Function):
    """Returns the concatenated heads [N, H*C] (bias fused when given)."""

    @staticmethod
    def forward(ctx, x, weight, att, bias, graph: CSRGraph, heads: int, slope: float, relu: bool = False,
                sch: Schedule = DEFAULT, x_scales: Optional[torch.Tensor] = None, want_scales: bool = False):
        H = int(heads)
        C = weight.size(1) // H
        att2 = _f32c(att.reshape(H, 2 * C), "att")
        d = graph.by_dst
        # x_scales (row_scales(x): computed once for a feature matrix that does not change between steps, or handed on by the
        # layer in front): the projection on two fp16 pieces per operand
        _check_scales(x_scales, x.size(0), "gat_conv(x_scales=)")
        xs = x_scales if (x_scales is not None and _f16x2(sch, x.size(0), weight.size(0), weight.size(1), x.dtype)) else None
        fl = _gflags(sch)                                       # (exact-f32 kernels on request: no store-epilogue fusions)
        if H == 1 and sch.gat_scores_epilogue and not fl and linear_fwd_scores_ok(x, weight):
            hfeat, a_dst, a_src = linear_fwd_scores(x, weight, att2, a_scales=xs)   # x @ W, both scores in its store epilogue
        else:
            hfeat = linear_fwd(x, weight, a_scales=xs, flags=fl)             # x @ W
            a_dst, a_src = gat_scores(hfeat, att2, H, C)
        out_scales = None
        if H == 1 and C % 4 == 0 and C <= 256 and d.nnz_max > 0 and sch.gat_fused_stats:
            # the statistics inside the aggregation launch: every item computes its entries' scores, rows cut by an item boundary
            # merge their parts' (max, sum exp) where cut rows are resolved (round 5: one pass over col / rowidx and a launch less)
            if want_scales and C == 256:
                out_scales = torch.empty(d.n_rows, dtype=torch.float32, device=x.device)
            out, m, s = gat_aggregate_fused(d, hfeat, None, C, a_dst, att2, slope, bias=bias, relu=relu, scales_out=out_scales)
        elif H == 1 and C % 4 == 0 and d.nnz_max > 0:
            # the statistics pass leaves the score of every entry; the aggregation reads it back (one coalesced load per
            # 64 entries) instead of gathering a_src[j] per entry and redoing the leaky_relu
            m, s, scores = gat_softmax_stats(d, a_dst, a_src, H, slope, want_scores=True)
            out = gat_aggregate_scores(d, hfeat, None, C, scores, m, s, bias=bias, relu=relu)
            del scores
        else:
            m, s = gat_softmax_stats(d, a_dst, a_src, H, slope)
            out = _gat_aggregate(graph, d, hfeat, H, C, a_dst, a_src, m, s, slope, False, bias=bias)
            if relu:                                     # several heads / odd widths: the ReLU as its own pass
                out = torch.relu_(out)
        ctx.relu = bool(relu)
        ctx.graph, ctx.H, ctx.C, ctx.slope = graph, H, C, float(slope)
        ctx.has_bias = bias is not None
        ctx.sch = sch
        ctx.x_scales = xs                                    # (the backward's dW GEMM takes its column scales from them)
        ctx.save_for_backward(x, weight, att2, hfeat, a_dst, a_src, m, s, out,
                              bias if bias is not None else torch.empty(0, device=x.device))
        if want_scales:
            if out_scales is None:                                           # (a shape whose aggregation cannot write them)
                out_scales = torch.empty(0, dtype=torch.float32, device=x.device)
            ctx.mark_non_differentiable(out_scales)
            return out, out_scales
        return out

    @staticmethod
    def backward(ctx, grad_out, _grad_scales=None):
        x, weight, att2, hfeat, a_dst, a_src, m, s, out, bias = ctx.saved_tensors
        graph: CSRGraph = ctx.graph
        H, C, slope, sch = ctx.H, ctx.C, ctx.slope, ctx.sch
        dev = x.device
        grad_out = _f32c(grad_out, "grad_out")
        d, sr = graph.by_dst, graph.by_src
        N = graph.num_nodes
        # D_i = <dOut_i, out_i - b> = sum_p alpha_p dalpha_p  (softmax backward) and db = column sums of dOut: one pass
        # (a fused ReLU: the same pass masks the gradient first and hands the masked gradient on)
        if ctx.relu:
            D, db, grad_out = gat_rowdot_colsum(grad_out, out, bias if ctx.has_bias else None, H, C,
                                                want_colsum=ctx.has_bias and ctx.needs_input_grad[3], relu_mask=True)
        else:
            D, db = gat_rowdot_colsum(grad_out, out, bias if ctx.has_bias else None, H, C,
                                      want_colsum=ctx.has_bias and ctx.needs_input_grad[3])
        if gat_fused_shape(H, C):
            # (a_dst, m, 1/s, D) of every (target, head) in one float4; alpha is recomputed per entry by the lane that owns it
            tpack = gat_pack_targets(a_dst, m, s, D)                           # [N H, 4]
            dh = torch.empty((N, H * C), dtype=torch.float32, device=dev)
            rank2 = (sch.gat_rank2_epilogue and not sch.gemm_exact_f32 and H == 1 and N >= sch.gat_rank2_min_rows and ctx.needs_input_grad[0]
                     and x.dtype == torch.float32 and weight.size(0) % 4 == 0 and linear_bwd_data_rank2_ok(dh, weight))
            # dX's GEMM on two fp16 pieces per operand: the row scales of d hfeat from the pass that writes it (256 channels)
            dh_scales = (torch.empty(N, dtype=torch.float32, device=dev)
                         if rank2 and H * C == 256 and _f16x2(sch, N, H * C, weight.size(0), x.dtype) else None)
            # one head: the by-source row sums of dz (g_src) come out of the same launch
            g_src = torch.empty((N, 1), dtype=torch.float32, device=dev) if (H == 1 and sch.gat_src_rowsum_fused) else None
            dh, dz = gat_backward_fused_packed(sr, grad_out, None, hfeat, C, tpack, a_src, slope, out=dh, H=H, scales_out=dh_scales,
                                               rowsum_out=g_src)
            dz = dz.view(-1, H)
            if rank2:
                return _GatConvFn._backward_rank2(ctx, x, weight, att2, dh, dz, db, C, dh_scales, g_src)
            if g_src is None:
                g_src = seg_rowsum(sr, dz, H)                                     # dz is in by-source entry order here
            g_dst = seg_rowsum(d, dz, H, map_=_inverse_transpose_map(graph))
            # d hfeat_j = sum_i alpha_ij dOut_i + g_dst[j] att[:C] + g_src[j] att[C:]
            gat_rank1_add(dh, g_dst, g_src, att2, H, C)
        else:
            # dz per by-target entry, then its row sums in both orientations;
            # one head: keep the alpha this kernel computes; the by-source pass reads it back through the transpose map
            alpha = torch.empty((max(d.nnz_max, 1), H), dtype=torch.float32, device=dev) if H == 1 else None
            dz = gat_edge_grad(d, hfeat, None, grad_out, H, C, a_dst, a_src, m, s, D, slope, 0, alpha_out=alpha)
            g_dst = seg_rowsum(d, dz, H)
            g_src = seg_rowsum(sr, dz, H, map_=_transpose_map(graph))
            dh = _gat_aggregate(graph, sr, grad_out, H, C, a_dst, a_src, m, s, slope, True,
                                g_dst=g_dst, g_src=g_src, att=att2, alpha=alpha,
                                alpha_map=_transpose_map(graph) if alpha is not None else None)
        # datt streams hfeat once (HBM-bound) and depends only on g_dst / g_src; the two GEMMs that follow are MFMA-bound:
        # on large graphs the attention gradient runs on the side stream UNDER them
        overlap = (ctx.needs_input_grad[2] and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and _overlaps(sch, N))
        datt = None
        if overlap:
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                datt = gat_att_grad(hfeat, g_dst, g_src, H, C).view(1, H, 2 * C)
            for t in (hfeat, g_dst, g_src):
                t.record_stream(side)
            datt.record_stream(main)
        elif ctx.needs_input_grad[2]:
            datt = gat_att_grad(hfeat, g_dst, g_src, H, C).view(1, H, 2 * C)
        dw = linear_bwd_weight(x, dh, want_bias=False, flags=_gflags(sch))[0] if ctx.needs_input_grad[1] else None
        dx = linear_bwd_data(dh, weight, flags=_gflags(sch)) if ctx.needs_input_grad[0] else None
        if overlap:
            main.wait_stream(side)
        return dx, dw, datt, db, None, None, None, None, None, None, None

    @staticmethod
    def _backward_rank2(ctx, x, weight, att2, dh, dz, db, C, dh_scales=None, g_src=None):
        """The tail of the one-head backward without ever forming d hfeat' = dh + g_dst (x) a1 + g_src (x) a2 (a1 = att[:C],
        a2 = att[C:]; g_dst / g_src = row sums of dz by target / by source).  With P = [x^T g_dst; x^T g_src] ([2, K], ONE
        pass over x):
            dX   = dh W^T + g_dst (x) (W a1) + g_src (x) (W a2)      the rank-2 term in the GEMM's store epilogue
            dW   = x^T dh + P^T [a1; a2]                              a [K, C] outer-product correction
            datt = [P W]                                              since hfeat = x W
        Only dX needs g_dst / g_src; dW = x^T dh needs neither.  So the MFMA-bound dW GEMM goes on the launch stream and the
        HBM-bound passes -- the by-target row sum of dz (a gather through the transpose map) and the pass over x -- run on
        the side stream beside it (both fit next to a dW workgroup on a CU: <= 70 VGPRs, <= 8 KB LDS); dX waits for the row
        sums only.  The [2, .] products
        are two small launches (``npi_gat_rank2_cols`` in front, ``npi_gat_rank2_tail`` behind)."""
        dev = x.device
        graph: CSRGraph = ctx.graph
        K = weight.size(0)
        A = att2.view(2, C)                                                   # rows a1, a2
        U = gat_rank2_cols(weight, A)                                         # [2, K]: W a1, W a2
        main = torch.cuda.current_stream(dev)
        overlap = _overlaps(ctx.sch, x.size(0))
        side = _side_stream(dev) if overlap else main
        # dW = x^T dh is EXPOSED here (the pass that produced dh is the layer's big HBM-bound kernel): two fp16 pieces per operand
        # when both operands come with row scales -- x's handed in (gat_conv(x_scales=)), dh's written by the fused pass -- from
        # which their column scales follow without a pass over either matrix (col_scales(row_scales=))
        dw_kw = {}
        if (dh_scales is not None and ctx.x_scales is not None and ctx.needs_input_grad[1]
                and dw_f16x2_shape(x.size(0), x.size(1), dh.size(1)) and x.dtype == torch.float32):
            dw_kw = dict(a_cs=col_scales(row_scales=ctx.x_scales, cols=x.size(1)), dc_cs=col_scales(row_scales=dh_scales, cols=dh.size(1)))
        tmap = _inverse_transpose_map(graph)                                  # cached; built on the launch stream
        # g_src handed in: the fused pass has already summed dz by source row (Schedule.gat_src_rowsum_fused).  Otherwise: dz is in
        # by-source entry order, its by-source row sum a coalesced, latency-bound pass (0.09 ms alone, 0.31 ms with one wave per
        # SIMD beside a dW workgroup), so it stays in front of dW unless the schedule asks for it beside dW
        src_beside = overlap and ctx.sch.gat_src_rowsum_beside_dw and g_src is None
        if not src_beside and g_src is None:
            g_src = seg_rowsum(graph.by_src, dz, 1)
        if overlap:
            side.wait_stream(main)
            # resident before the side stream's passes ask for wave slots
            dw = linear_bwd_weight(x, dh, want_bias=False, **dw_kw)[0] if ctx.needs_input_grad[1] else None
        with torch.cuda.stream(side):
            if src_beside:
                g_src = seg_rowsum(graph.by_src, dz, 1)
            g_dst = seg_rowsum(graph.by_dst, dz, 1, map_=tmap)                # 128-byte lines of a 4-byte permutation
            have_g = torch.cuda.Event()
            have_g.record(side)
            P = gat_att_grad(x, g_dst, g_src, 1, K).view(2, K)                # x^T g_dst, x^T g_src
        if not overlap:
            dw = linear_bwd_weight(x, dh, want_bias=False, **dw_kw)[0] if ctx.needs_input_grad[1] else None
        main.wait_event(have_g)
        dx = linear_bwd_data_rank2(dh, weight, g_dst, g_src, U[0], U[1], dc_scales=dh_scales)
        if overlap:
            for t in (x, dz, tmap, g_src):
                t.record_stream(side)
            for t in (P, g_dst) + ((g_src,) if src_beside else ()):
                t.record_stream(main)
            main.wait_stream(side)
        # dW += P^T [a1; a2] and d att = P W in one small launch behind the GEMMs
        datt = gat_rank2_tail(P, weight, A, dw, ctx.needs_input_grad[2])
        if datt is not None:
            datt = datt.view(1, 1, 2 * C)
        return dx, dw, datt, db, None, None, None, None, None, None, None


def _valid_entries(side: CSRSide) -> torch.Tensor:
    """``[nnz_max, 1]`` bool: entry slots that hold an entry (capacity past ``rowptr[N]`` -- dropped self loops, out-of-range ids -- is
    never written by the CSR build and never read by a kernel; element-wise torch ops over whole entry arrays must mask it)"""
    return (torch.arange(max(side.nnz_max, 1), device=side.rowptr.device) < side.rowptr[-1]).view(-1, 1)


class _GatDropoutFn(torch.autograd.Function):
    """GATConv with ATTENTION DROPOUT in training mode (PyG 1.4.2 ``GATConv.message``: ``alpha = softmax(alpha, edge_index_i)``,
    ``alpha = F.dropout(alpha, p, training)``, ``x_j * alpha``).  ``keep`` ``[nnz_max, H]``, in by-target entry order, holds
    ``0`` for a dropped attention weight and ``1 / (1 - p)`` for a kept one.  The layer's fast kernels never hold alpha (both
    directions recompute it per entry), so this variant -- a regulariser for small graphs, used by nothing in the reference --
    materialises alpha per entry and head and is COMPOSED from the per-op kernels: the score / statistics pass, one weighted
    aggregation per head in each direction (``npi_segsum_ex`` with per-entry weights over a column block), the per-entry dots
    ``<dOut_i, h_j>`` from ``npi_gat_edge_grad`` (called with alpha = 1, D = 0, slope = 1: its dz IS the dot), segmented row sums
    and the element-wise softmax backward on ``[nnz, H]`` arrays in torch.  With dropout the softmax term is unchanged:
    ``sum_j alpha_ij dalpha_ij = sum_j alpha'_ij <dOut_i, h_j> = <dOut_i, out_i - b>`` (alpha' = alpha keep)."""

    @staticmethod
    def forward(ctx, x, weight, att, bias, graph: CSRGraph, heads: int, slope: float, keep: torch.Tensor):
        H = int(heads)
        C = weight.size(1) // H
        att2 = _f32c(att.reshape(H, 2 * C), "att")
        d = graph.by_dst
        N = d.n_rows
        keep = _f32c(keep, "keep")
        if keep.shape != (max(d.nnz_max, 1), H):
            raise ValueError(f"gat_conv: keep must be [{max(d.nnz_max, 1)}, {H}] (by-target entry order), got {tuple(keep.shape)}")
        hfeat = linear_fwd(x, weight)
        a_dst, a_src = gat_scores(hfeat, att2, H, C)
        m, s, e = gat_softmax_stats(d, a_dst, a_src, H, slope, want_scores=True)
        ri = d.rowidx.long().clamp_(0, max(N - 1, 0))                       # (slots past nnz hold garbage row ids)
        valid = _valid_entries(d)
        alpha = torch.where(valid, torch.exp(e - m[ri]) / (s[ri] + 1e-16), torch.zeros((), device=x.device))
        alpha_k = torch.where(valid, alpha * keep, torch.zeros((), device=x.device))
        out = torch.empty((N, H * C), dtype=torch.float32, device=x.device)
        for h in range(H):
            blk = slice(h * C, (h + 1) * C)
            segsum(graph, d, hfeat[:, blk], w=alpha_k[:, h].contiguous(), out=out[:, blk],
                   bias=bias[blk].contiguous() if bias is not None else None)
        ctx.graph, ctx.H, ctx.C, ctx.slope, ctx.has_bias = graph, H, C, float(slope), bias is not None
        ctx.save_for_backward(x, weight, att2, hfeat, e, alpha, alpha_k, keep, out,
                              bias if bias is not None else torch.empty(0, device=x.device))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, att2, hfeat, e, alpha, alpha_k, keep, out, bias = ctx.saved_tensors
        graph: CSRGraph = ctx.graph
        H, C, slope = ctx.H, ctx.C, ctx.slope
        dev = x.device
        grad_out = _f32c(grad_out, "grad_out")
        d, sr = graph.by_dst, graph.by_src
        N = d.n_rows
        D, db = gat_rowdot_colsum(grad_out, out, bias if ctx.has_bias else None, H, C,
                                  want_colsum=ctx.has_bias and ctx.needs_input_grad[3])
        zeros, ones = torch.zeros((N, H), device=dev), torch.ones((N, H), device=dev)
        dots = gat_edge_grad(d, hfeat, None, grad_out, H, C, zeros, zeros, zeros, ones, zeros, 1.0, 0)    # <dOut_i, h_j>
        ri = d.rowidx.long().clamp_(0, max(N - 1, 0))
        zero = torch.zeros((), device=dev)
        gprime = torch.where(e > 0, torch.ones((), device=dev), torch.full((), slope, device=dev))        # e = leaky_relu(z)
        dz = torch.where(_valid_entries(d), alpha * (keep * dots - D[ri]) * gprime, zero)
        tm = _transpose_map(graph)
        g_dst = seg_rowsum(d, dz, H)
        g_src = seg_rowsum(sr, dz, H, map_=tm)
        # d hfeat_j = sum_i alpha'_ij dOut_i + g_dst[j] att[:C] + g_src[j] att[C:]
        alpha_src = torch.where(_valid_entries(sr), alpha_k[tm.long().clamp_(0, alpha_k.size(0) - 1)], zero)
        dh = torch.empty((N, H * C), dtype=torch.float32, device=dev)
        for h in range(H):
            blk = slice(h * C, (h + 1) * C)
            segsum(graph, sr, grad_out[:, blk], w=alpha_src[:, h].contiguous(), out=dh[:, blk])
        gat_rank1_add(dh, g_dst, g_src, att2, H, C)
        datt = gat_att_grad(hfeat, g_dst, g_src, H, C).view(1, H, 2 * C) if ctx.needs_input_grad[2] else None
        dw = linear_bwd_weight(x, dh, want_bias=False)[0] if ctx.needs_input_grad[1] else None
        dx = linear_bwd_data(dh, weight) if ctx.needs_input_grad[0] else None
        return dx, dw, datt, db, None, None, None, None


def gat_dropout_keep(graph: CSRGraph, heads: int, p: float) -> torch.Tensor:
    """A fresh ``keep`` array for ``gat_conv(..., keep=)``: Bernoulli(1 - p) / (1 - p) per by-target entry and head (torch's
    generator: the distribution of ``F.dropout``, not its random bits for a given seed)."""
    if not 0.0 <= p < 1.0:
        raise ValueError("gat_dropout_keep: p must be in [0, 1)")
    d = graph.by_dst
    return torch.empty((max(d.nnz_max, 1), int(heads)), dtype=torch.float32, device=d.rowptr.device).bernoulli_(1.0 - p).div_(1.0 - p)


def gat_conv(x: torch.Tensor, edge_index, weight: torch.Tensor, att: torch.Tensor,
             bias: Optional[torch.Tensor] = None, heads: int = 1, concat: bool = True,
             negative_slope: float = 0.2, relu: bool = False, schedule: Schedule = DEFAULT,
             keep: Optional[torch.Tensor] = None, x_scales: Optional[torch.Tensor] = None, return_scales: bool = False):
    """PyG 1.4.2 ``GATConv.forward`` on MI355X; ``att`` is ``[1, H, 2C]``.  ``relu=True`` (an extension, as in
    ``sage_conv``): ``F.relu(conv(x, edge_index))`` with the ReLU in the aggregation's row epilogue and its backward mask in the
    pass that computes the softmax term -- one head; other shapes apply it as a separate pass.  ``keep``: attention dropout in
    training mode (``_GatDropoutFn``; ``gat_dropout_keep`` draws one) -- the composed, slower variant of the layer.
    ``x_scales``: ``row_scales(x)`` -- the projection ``x @ W`` then runs on two fp16 pieces per operand (large graphs; EXPERIMENTS
    A34).  Computed ONCE for a feature matrix that does not change between steps, or taken from the layer in front:
    ``return_scales=True`` returns ``(out, out_scales)`` with the scales of the output's rows written by the aggregation launch
    itself (one head of 256 channels; otherwise ``out_scales`` is None) -- what the next layer takes as its ``x_scales``."""
    require_gpu(x, weight, att, bias, x_scales)
    graph = as_graph(edge_index, x.size(0))
    if keep is not None:
        out = _GatDropoutFn.apply(x, weight, att, bias if concat else None, graph, heads, negative_slope, keep)
        if not concat:
            out = out.view(x.size(0), heads, -1).mean(dim=1)
            out = out + bias if bias is not None else out
        out = torch.relu(out) if relu else out
        return (out, None) if return_scales else out
    if concat:
        if return_scales:
            out, sc = _GatConvFn.apply(x, weight, att, bias, graph, heads, negative_slope, relu, schedule, x_scales, True)
            return out, (sc if sc.numel() else None)
        return _GatConvFn.apply(x, weight, att, bias, graph, heads, negative_slope, relu, schedule, x_scales, False)
    out = _GatConvFn.apply(x, weight, att, None, graph, heads, negative_slope, False, schedule, x_scales, False)
    out = out.view(x.size(0), heads, -1).mean(dim=1)          # head average (concat=False)
    out = out + bias if bias is not None else out
    out = torch.relu(out) if relu else out
    return (out, None) if return_scales else out
