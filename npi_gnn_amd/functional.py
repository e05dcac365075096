"""Functional form of the conv hot path on MI355X: thin wrappers over the C ABI plus the
``torch.autograd.Function``s that give ``loss.backward()`` (reference
``src/train_with_twoDataset.PY:54``) the same gradients PyG 1.4.2's autograd graph produces.

Formulas (PyG 1.4.2):
  SAGEConv : out = mean_{j in N(i) U {i}} x_j @ W + b          (aggregate at F_in, then project)
  GCNConv  : out = sum_e norm_e (x @ W)[src e] + b,  norm_e = d^-1/2[src] w_e d^-1/2[dst],
             d = weighted out-degree incl. the self loop      (project first, aggregate at F_out)
"""
from __future__ import annotations

from typing import Optional

import torch

from ._lib import NPI_F32, check, load, ptr, require_gpu, stream_ptr
from .graph import CSRGraph, CSRSide, as_graph


_PROFILE = None     # bench.py sets this to a list to collect (start, end) events per segsum launch


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# raw ops
# ---------------------------------------------------------------------------------------------
def segsum(graph: CSRGraph, side: CSRSide, x: torch.Tensor, w: Optional[torch.Tensor] = None,
           mean: bool = False, bias: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = scale_i * sum_{p in row i} w[p] * x[col[p]] (+ bias): fused gather + segmented
    reduction (``npi_segsum``)."""
    dev = require_gpu(x, w, bias)
    x = _f32c(x, "x")
    N, F = side.n_rows, x.size(1)                  # `graph` may be None for a stand-alone (sharded) side
    if x.size(0) != side.n_cols:
        raise ValueError(f"x has {x.size(0)} rows, the adjacency indexes a table of {side.n_cols}")
    if out is None:
        out = torch.empty((N, F), dtype=torch.float32, device=dev)
    carry = side.carry(F)
    prof = _PROFILE
    if prof is not None:        # bench.py: HIP events on the launch stream around this launch
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
    check(load().npi_segsum(ptr(side.rowptr), ptr(side.col), ptr(side.item_row), ptr(w), N, side.nnz_max,
                            ptr(x), x.stride(0), ptr(out), out.stride(0), F, NPI_F32, 1 if mean else 0,
                            ptr(bias), ptr(carry), stream_ptr(dev)), "npi_segsum")
    if prof is not None:
        ev1.record(torch.cuda.current_stream(dev))
        prof.append((ev0, ev1))
    return out


def linear_fwd(a: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
               rowscale: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    dev = require_gpu(a, weight, bias, rowscale)
    a, weight = _f32c(a, "a"), _f32c(weight, "weight")
    M, K = a.shape
    N = weight.size(1)
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    check(load().npi_linear_fwd(ptr(a), a.stride(0), ptr(weight), weight.stride(0), ptr(bias), ptr(rowscale),
                                ptr(out), out.stride(0), M, K, N, 1 if relu else 0, stream_ptr(dev)),
          "npi_linear_fwd")
    return out


def linear_bwd_data(dc: torch.Tensor, weight: torch.Tensor,
                    rowscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    dev = require_gpu(dc, weight, rowscale)
    dc, weight = _f32c(dc, "dC"), _f32c(weight, "weight")
    M, N = dc.shape
    K = weight.size(0)
    da = torch.empty((M, K), dtype=torch.float32, device=dev)
    check(load().npi_linear_bwd_data(ptr(dc), dc.stride(0), ptr(weight), weight.stride(0), ptr(rowscale),
                                     ptr(da), da.stride(0), M, K, N, stream_ptr(dev)), "npi_linear_bwd_data")
    return da


def linear_bwd_weight(a: torch.Tensor, dc: torch.Tensor, want_bias: bool = True):
    dev = require_gpu(a, dc)
    a, dc = _f32c(a, "a"), _f32c(dc, "dC")
    M, K = a.shape
    N = dc.size(1)
    lib = load()
    n_ws = int(lib.npi_linear_bwd_weight_workspace_elems(M, K, N))
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    dw = torch.empty((K, N), dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    check(lib.npi_linear_bwd_weight(ptr(a), a.stride(0), ptr(dc), dc.stride(0), ptr(dw), dw.stride(0), ptr(db),
                                    M, K, N, ptr(ws), n_ws, stream_ptr(dev)), "npi_linear_bwd_weight")
    return dw, db


def colsum(x: torch.Tensor) -> torch.Tensor:
    dev = require_gpu(x)
    x = _f32c(x, "x")
    M, N = x.shape
    n_ws = ((max(M, 1) + 2047) // 2048) * N
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    out = torch.empty(N, dtype=torch.float32, device=dev)
    check(load().npi_colsum(ptr(x), x.stride(0), M, N, ptr(out), ptr(ws), n_ws, stream_ptr(dev)), "npi_colsum")
    return out


# ---------------------------------------------------------------------------------------------
# SAGEConv
# ---------------------------------------------------------------------------------------------
class _SageConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, graph: CSRGraph):
        agg = segsum(graph, graph.by_dst, x, mean=True)                  # a2-a4: gather + scatter_mean
        out = linear_fwd(agg, weight, bias)                               # a5: agg @ W + b
        ctx.graph = graph
        ctx.has_bias = bias is not None
        ctx.save_for_backward(agg, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        agg, weight = ctx.saved_tensors
        graph: CSRGraph = ctx.graph
        grad_out = _f32c(grad_out, "grad_out")
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = linear_bwd_weight(agg, grad_out, want_bias=ctx.has_bias)       # aggT dOut, colsum
        if ctx.needs_input_grad[0]:
            # dAgg = dOut W^T, pre-divided by the in-count of its row (fused epilogue), then
            # dX[j] = sum over the entries whose SOURCE is j  ==  segsum over the by-source CSR
            dagg = linear_bwd_data(grad_out, weight, rowscale=graph.inv_count(graph.by_dst))
            dx = segsum(graph, graph.by_src, dagg, mean=False)
        return dx, dw, db, None


def sage_conv(x: torch.Tensor, edge_index, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
              normalize: bool = False) -> torch.Tensor:
    """PyG 1.4.2 ``SAGEConv(normalize=False, concat=False).forward`` on MI355X
    (call sites: reference ``src/classes.py:62,66,70``)."""
    require_gpu(x, weight, bias)
    graph = as_graph(edge_index, x.size(0))
    out = _SageConvFn.apply(x, weight, bias, graph)
    if normalize:
        out = torch.nn.functional.normalize(out, p=2.0, dim=-1)
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
        w_entry = [None, None]
        if edge_weight is not None or improved:
            loop_w = None
            if edge_weight is not None:
                require_gpu(edge_weight)
                edge_weight = _f32c(edge_weight.detach(), "edge_weight")
            # add_remaining_self_loops: an existing self loop's weight (GCNConv.norm passes ones when
            # edge_weight is None) becomes that node's loop weight instead of `fill`
            src, dst = graph._src, graph._dst
            m = src == dst
            if bool(m.any()):
                loop_w = torch.full((N,), fill, dtype=torch.float32, device=dev)
                loop_w[src[m]] = edge_weight[m] if edge_weight is not None else 1.0
            for k, side in enumerate(sides):
                we = torch.empty(max(side.nnz_max, 1), dtype=torch.float32, device=dev)
                check(lib.npi_entry_weights(ptr(side.eid), ptr(side.rowidx), ptr(side.rowptr), ptr(edge_weight),
                                            ptr(loop_w), fill, N, side.nnz_max, ptr(we), s), "npi_entry_weights")
                w_entry[k] = we
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


class _GcnConvFn(torch.autograd.Function):
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
        grad_out = _f32c(grad_out, "grad_out")
        dx = dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(grad_out)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dxw = segsum(graph, graph.by_src, grad_out, w=norm.by_src)        # A_hat^T dOut
            if ctx.needs_input_grad[1]:
                dw, _ = linear_bwd_weight(x, dxw, want_bias=False)
            if ctx.needs_input_grad[0]:
                dx = linear_bwd_data(dxw, weight)
        return dx, dw, db, None


def gcn_conv(x: torch.Tensor, edge_index, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
             edge_weight: Optional[torch.Tensor] = None, improved: bool = False,
             norm: Optional[GCNNorm] = None) -> torch.Tensor:
    """PyG 1.4.2 ``GCNConv.forward`` (normalize=True) on MI355X."""
    require_gpu(x, weight, bias)
    if edge_weight is not None and edge_weight.requires_grad:
        raise NotImplementedError("gradients w.r.t. edge_weight are not implemented")
    if norm is None:
        norm = GCNNorm(as_graph(edge_index, x.size(0)), edge_weight, improved)
    return _GcnConvFn.apply(x, weight, bias, norm)
