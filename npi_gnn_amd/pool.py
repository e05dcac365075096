"""Between-layer steps of the reference's ``Net_1`` on MI355X: PyG 1.4.2
``TopKPooling(in_channels, ratio)`` and ``global_max_pool`` / ``global_mean_pool``
(reference ``src/classes.py:49,51,53,63-64,67-68,71-72``), forward and
backward.

With these and ``npi_gnn_amd.nn.SAGEConv`` the whole ``Net_1`` (``src/classes.py:59-82``) is
GPU-resident for the train loop (``src/train_with_twoDataset.PY:52-54``) and for the evaluation /
case-study flows (``src/methods.py:87-96``, ``src/test.py``, ``src/case_study*.py``).  Gradients flow
to the layer input (kept rows only, as in PyG: the selection itself is not differentiable), to
``TopKPooling.weight`` and through ``[max || mean]``; ``edge_index`` / ``batch`` / ``perm`` are index
outputs.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
from torch import nn
from torch.nn import Parameter

from ._lib import check, load, ptr, require_gpu, stream_ptr


# ``padded_edges=True`` (an explicit argument of topk_pool / TopKPooling; net1.Net_1 sets it) AND the sizes of the batch's
# graphs known on the host (batch._npi_sizes): TopKPooling makes no device read, see _topk_pool_fwd.  The edge_index it
# returns is then PADDED to the input's length with (-1, -1) columns -- NOT PyG's contract, which is why it is opt-in.
NO_SYNC = True
LDS_SORT_MAX_NODES = 16384       # npi_topk_select sorts one graph's (score, index) keys in LDS; larger graphs: radix selection


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# the pooled graph's CSR by filtering the parent's instead of a fresh sort (NPI_DERIVE_CSR=0: always sort)
DERIVE_CSR = os.environ.get("NPI_DERIVE_CSR", "1") != "0"


def graph_ptr(batch: torch.Tensor, num_graphs: Optional[int] = None) -> torch.Tensor:
    """int32 ``[B+1]`` segment starts of a PyG ``batch`` vector (non-decreasing int64)."""
    dev = require_gpu(batch)
    if batch.dtype != torch.int64:
        raise TypeError("batch must be a LongTensor")
    N = batch.numel()
    known = getattr(batch, "_npi_graph_ptr", None)          # left by topk_pool on the batch vector it produced
    if known is not None and (num_graphs is None or int(num_graphs) == known.numel() - 1):
        return known
    sizes = getattr(batch, "_npi_sizes", None)             # host-known graph sizes (net1.KeyLoader): no device read
    if num_graphs is None and sizes is not None:
        num_graphs = int(sizes.numel())
    B = int(num_graphs) if num_graphs is not None else (int(batch[-1].item()) + 1 if N else 0)
    gp = torch.empty(B + 1, dtype=torch.int32, device=dev)
    check(load().npi_graph_bounds(ptr(batch.contiguous()), N, B, ptr(gp), stream_ptr(dev)), "npi_graph_bounds")
    return gp


def _topk_pool_fwd(x: torch.Tensor, edge_index: torch.Tensor, batch: torch.Tensor, weight: torch.Tensor,
                   ratio: float = 0.5, num_graphs: Optional[int] = None, padded_edges: bool = False):
    """forward kernels; returns the public tuple plus (score [N], perm int32 [n_out], remap int32 [N]: old -> new id or -1)
    for the backward"""
    lib = load()
    dev = require_gpu(x, edge_index, batch, weight)
    x = _f32(x.detach())
    w = _f32(weight.detach().reshape(-1))
    N, F = x.shape
    st = stream_ptr(dev)
    i32 = dict(dtype=torch.int32, device=dev)
    gp = graph_ptr(batch, num_graphs)
    B = gp.numel() - 1
    score = torch.empty(N, dtype=torch.float32, device=dev)
    check(lib.npi_topk_score(ptr(x), x.stride(0), ptr(w), N, F, ptr(score), st), "npi_topk_score")
    out_ptr = torch.empty(B + 1, **i32)
    perm = torch.empty(max(N, 1), **i32)
    remap = torch.empty(max(N, 1), **i32)
    status = torch.empty(1, **i32)
    sizes = getattr(batch, "_npi_sizes", None)               # host-known graph sizes: the largest one picks the kernels
    max_nodes = int(sizes.max()) if sizes is not None and sizes.numel() == B and B > 0 else 0

    def select_sorted():
        """graphs above the LDS sort's 16,384 nodes: two device-wide stable radix sorts (npi_topk_select_sorted)"""
        n_ws = int(lib.npi_topk_sorted_workspace_bytes(N))
        wsb = torch.empty(n_ws, dtype=torch.uint8, device=dev)
        check(lib.npi_topk_select_sorted(ptr(score), ptr(batch.contiguous()), ptr(gp), N, B, float(ratio), ptr(out_ptr),
                                         ptr(perm), ptr(remap), ptr(wsb), n_ws, st), "npi_topk_select_sorted")
    if max_nodes > LDS_SORT_MAX_NODES:                      # known on the host: straight to the radix selection
        status.zero_()
        select_sorted()
    else:
        check(lib.npi_topk_select_ex(ptr(score), ptr(gp), N, B, float(ratio), ptr(out_ptr), ptr(perm), ptr(remap),
                                     ptr(status), max_nodes, st), "npi_topk_select")
    # filter_adj needs only the old->new id map, so it runs before the sizes are known
    E = edge_index.size(1)
    src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
    out_ei = torch.empty((2, max(E, 1)), dtype=torch.int64, device=dev)
    count = torch.empty(1, **i32)
    ws = torch.empty(int(lib.npi_filter_adj_workspace_elems(E)), **i32)
    # Host-known graph sizes (``batch._npi_sizes``, left by net1.KeyLoader or by the previous pooling layer): the kept
    # node count is ceil(ratio n_g) per graph -- computable on the host -- and the surviving edges stay in an array of the
    # input's length whose tail is (-1, -1) padding (npi_filter_adj_ex), which every consumer drops.  No device read at
    # all: the layer, and with it the whole Net_1 step, runs without a host synchronisation and captures into a HIP graph.
    kept = _kept_sizes(batch, ratio) if (B > 0 and padded_edges) else None
    nosync = kept is not None and kept.numel() == B
    check(lib.npi_filter_adj_ex(ptr(src), ptr(dst), E, ptr(remap), ptr(out_ei[0]), ptr(out_ei[1]), ptr(count), ptr(ws),
                                1 if nosync else 0, st), "npi_filter_adj")
    if nosync:
        n_out, e_out = int(kept.sum()), E                # kept = ceil(ratio n_g) in the kernel's f32 arithmetic
    else:
        kept = None
        # sizes of the outputs are data dependent: ONE device read per pooling layer (PyG's own implementation has several)
        # (the same read also reports graph builds that dropped out-of-range node ids, graph.pending_status)
        from . import graph as _graph
        pend = _graph.pending_status(dev)                   # this device's words only; the others stay pending
        vals = torch.cat([out_ptr[-1:], status, count] + pend).tolist() if B else [0, 0, 0] + (torch.cat(pend).tolist() if pend else [])
        n_out, flags, e_out = vals[:3]
        _graph.raise_on_status(vals[3:])
        if flags & 2:
            # A graph with more than 16,384 nodes does not fit the LDS sort of npi_topk_select and nobody told us its size
            # beforehand: select again with the radix kernels (same rule: score descending, lower index first among equals,
            # ceil(ratio n) per graph), then filter_adj again with the new map -- all on the device.
            select_sorted()
            check(lib.npi_filter_adj(ptr(src), ptr(dst), E, ptr(remap), ptr(out_ei[0]), ptr(out_ei[1]), ptr(count), ptr(ws), st),
                  "npi_filter_adj")
            n_out, e_out = (int(v) for v in torch.cat([out_ptr[-1:], count]).tolist())
    xo = torch.empty((n_out, F), dtype=torch.float32, device=dev)
    batch_o = torch.empty(n_out, dtype=torch.int64, device=dev)
    score_o = torch.empty(n_out, dtype=torch.float32, device=dev)
    perm64 = torch.empty(n_out, dtype=torch.int64, device=dev)           # the LongTensor PyG returns, written by the gather
    check(lib.npi_topk_gather_ex(ptr(x), x.stride(0), ptr(score), ptr(batch.contiguous()), ptr(perm), ptr(out_ptr), B, F,
                                 n_out, ptr(xo), xo.stride(0), ptr(batch_o), ptr(score_o), ptr(perm64), st), "npi_topk_gather")
    # the kept-row offsets ARE the segment starts of the pooled batch vector: the readout and the next pooling
    # layer take them from here instead of searching `batch_o` again
    batch_o._npi_graph_ptr = out_ptr
    if kept is not None:
        batch_o._npi_sizes = kept                        # the next pooling layer knows its sizes as well
    ei_out = out_ei[:, :e_out]
    if getattr(edge_index, "_npi_symmetric", False):
        ei_out._npi_symmetric = True                     # both directions of a pair survive or fall together
    # The conv in front of this layer left the CSR of `edge_index` on it (graph.as_graph): the pooled graph's CSR is that
    # one filtered -- four launches, no sort -- and rides on the edge list this layer returns, where the next conv finds it.
    from .graph import cached_graph
    parent = cached_graph(edge_index, N)                 # None if the edge list was written to since its CSR was built
    if DERIVE_CSR and nosync and parent is not None and parent.self_loops:
        off = int(lib.npi_filter_adj_newpos_offset(E))
        # derived when (and only if) a conv asks for it: the last pooling layer's graph feeds no conv at all
        ei_out._npi_graph_from = (parent.by_dst, perm, remap, ws[off:off + E], n_out, e_out, ei_out._version)
    return (xo, ei_out, None, batch_o, perm64, score_o), (score, perm[:n_out], remap)


class _TopKPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, edge_index, batch, ratio, num_graphs, holder=None, padded_edges=False):
        (xo, ei_o, _, batch_o, perm, score_o), (score, perm32, remap) = _topk_pool_fwd(x, edge_index, batch, weight, ratio,
                                                                                       num_graphs, padded_edges)
        if holder is not None:                     # attributes do not survive the way out of an autograd Function
            holder["graph"] = getattr(ei_o, "_npi_graph_from", None)
        ctx.save_for_backward(x.detach(), weight.detach(), score, perm32, remap)
        out_ptr = batch_o._npi_graph_ptr
        ctx.mark_non_differentiable(ei_o, batch_o, perm, out_ptr)
        # no zero tensors for the gradients of outputs nobody used (score[perm], and autograd materialises them even for the
        # four integer outputs: five fill launches per layer and step)
        ctx.set_materialize_grads(False)
        return xo, score_o, ei_o, batch_o, perm, out_ptr

    @staticmethod
    def backward(ctx, dxo, dscore_o, *_unused):
        x, weight, score, perm, remap = ctx.saved_tensors
        lib = load()
        dev = x.device
        x = _f32(x)
        w = _f32(weight.reshape(-1))
        N, F = x.shape
        n_out = perm.numel()
        st = stream_ptr(dev)
        if dxo is None and dscore_o is None:
            return None, None, None, None, None, None, None, None
        dxo = _f32(dxo) if dxo is not None else torch.zeros((n_out, F), dtype=torch.float32, device=dev)
        dso = _f32(dscore_o) if dscore_o is not None else None
        dx = torch.empty((N, F), dtype=torch.float32, device=dev)          # dropped rows: zeros written by the kernel
        dzv = torch.empty(max(n_out, 1), dtype=torch.float32, device=dev)
        dzz = torch.empty(max(n_out, 1), dtype=torch.float32, device=dev)
        check(lib.npi_topk_gather_bwd_ex(ptr(x), x.stride(0), ptr(score), ptr(w), ptr(remap), N, F, ptr(dxo),
                                         dxo.stride(0), ptr(dso), ptr(dx), dx.stride(0), ptr(dzv), ptr(dzz), st),
              "npi_topk_gather_bwd")
        dw = None
        if ctx.needs_input_grad[1]:
            dw = torch.empty(F, dtype=torch.float32, device=dev)
            n_ws = int(lib.npi_topk_weight_grad_workspace_elems(n_out, F))
            ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
            check(lib.npi_topk_weight_grad(ptr(x), x.stride(0), ptr(perm), ptr(dzv), ptr(dzz), n_out, F, ptr(w), ptr(dw),
                                           ptr(ws), n_ws, st), "npi_topk_weight_grad")
            dw = dw.view_as(weight)
        return (dx if ctx.needs_input_grad[0] else None), dw, None, None, None, None, None, None


def _kept_sizes(batch, ratio):
    sizes = getattr(batch, "_npi_sizes", None)
    if not NO_SYNC or sizes is None or sizes.numel() == 0:
        return None
    return torch.ceil(torch.tensor(float(ratio), dtype=torch.float32) * sizes.to(torch.float32)).to(torch.int64)


def topk_pool(x: torch.Tensor, edge_index: torch.Tensor, batch: torch.Tensor, weight: torch.Tensor,
              ratio: float = 0.5, num_graphs: Optional[int] = None, padded_edges: bool = False):
    """``TopKPooling.forward`` -> ``(x', edge_index', None, batch', perm, score[perm])``; differentiable in
    ``x`` and ``weight``.

    ``padded_edges=False`` (default): PyG's contract -- ``edge_index'`` is exactly the surviving edges (one device read
    per call for the two data-dependent sizes).  ``padded_edges=True`` and ``batch._npi_sizes`` present (the per-graph
    node counts on the host, net1.KeyLoader): NO device read -- ``edge_index'`` then keeps the INPUT's length, the
    surviving edges first, in order, and a tail of ``(-1, -1)`` columns.  Only this package's consumers (the convs' CSR
    build, the next ``filter_adj``, ``entry_weights``) drop that tail; ``x[edge_index[0]]`` or ``edge_index.size(1)``
    in foreign code would not -- hence opt-in."""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        holder = {}
        xo, score_o, ei_o, batch_o, perm, out_ptr = _TopKPoolFn.apply(x, weight, edge_index, batch, ratio, num_graphs, holder,
                                                                      padded_edges)
        # (the tensors an autograd Function hands back need not be the objects its forward created: re-attach)
        batch_o._npi_graph_ptr = out_ptr
        kept = _kept_sizes(batch, ratio) if padded_edges else None
        if kept is not None:
            batch_o._npi_sizes = kept
        if getattr(edge_index, "_npi_symmetric", False):
            ei_o._npi_symmetric = True
        if holder.get("graph") is not None:
            ei_o._npi_graph_from = holder["graph"][:6] + (ei_o._version,)
        return xo, ei_o, None, batch_o, perm, score_o
    return _topk_pool_fwd(x, edge_index, batch, weight, ratio, num_graphs, padded_edges)[0]


def _readout_fwd(x: torch.Tensor, batch: torch.Tensor, num_graphs: Optional[int]):
    dev = require_gpu(x, batch)
    x = _f32(x.detach())
    gp = graph_ptr(batch, num_graphs)
    B, F = gp.numel() - 1, x.size(1)
    out = torch.empty((B, 2 * F), dtype=torch.float32, device=dev)
    check(load().npi_readout_max_mean(ptr(x), x.stride(0), ptr(gp), B, F, ptr(out), stream_ptr(dev)),
          "npi_readout_max_mean")
    return out, x, gp


class _ReadoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, batch, num_graphs):
        out, xc, gp = _readout_fwd(x, batch, num_graphs)
        ctx.save_for_backward(xc, gp, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gp, out = ctx.saved_tensors
        dev = x.device
        dout = _f32(dout)
        B, F = gp.numel() - 1, x.size(1)
        if B == 0:
            return torch.zeros_like(x), None, None
        dx = torch.empty_like(x)                  # rows outside every graph (none in PyG batches): zeroed by the kernel
        check(load().npi_readout_max_mean_bwd_ex(ptr(x), x.stride(0), ptr(gp), B, F, ptr(out), ptr(dout), ptr(dx),
                                                 dx.stride(0), x.size(0), stream_ptr(dev)), "npi_readout_max_mean_bwd")
        return dx, None, None


def global_max_mean_pool(x: torch.Tensor, batch: torch.Tensor, num_graphs: Optional[int] = None) -> torch.Tensor:
    """``cat([global_max_pool(x, batch), global_mean_pool(x, batch)], dim=1)`` -> ``[B, 2F]``; differentiable in ``x``."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _ReadoutFn.apply(x, batch, num_graphs)
    return _readout_fwd(x, batch, num_graphs)[0]


def global_max_pool(x, batch, size=None):
    return global_max_mean_pool(x, batch, size)[:, : x.size(1)]


def global_mean_pool(x, batch, size=None):
    return global_max_mean_pool(x, batch, size)[:, x.size(1):]


class TopKPooling(nn.Module):
    """``TopKPooling(in_channels, ratio=0.5)``: parameter ``weight [1, in_channels]`` (PyG 1.4.2 layout,
    so ``pool1.weight`` of the reference checkpoints loads unchanged)."""

    def __init__(self, in_channels: int, ratio: float = 0.5, padded_edges: bool = False, **kwargs):
        """``padded_edges``: see ``topk_pool`` -- with host-known graph sizes the layer reads nothing back and returns an
        edge list of the input's length with a ``(-1, -1)`` tail (for pipelines made of this package's layers only)."""
        super().__init__()
        self.in_channels, self.ratio, self.padded_edges = in_channels, ratio, bool(padded_edges)
        self.weight = Parameter(torch.empty(1, in_channels))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        bound = 1.0 / math.sqrt(self.in_channels)
        self.weight.data.uniform_(-bound, bound)

    def forward(self, x, edge_index, edge_attr=None, batch=None):
        if edge_attr is not None:
            raise NotImplementedError("TopKPooling: edge_attr is not used by NPI-GNN")
        if batch is None:
            batch = torch.zeros(x.size(0), dtype=torch.int64, device=x.device)
        return topk_pool(x, edge_index, batch, self.weight, self.ratio, padded_edges=self.padded_edges)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, ratio={self.ratio})"
