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
from typing import Optional

import torch
from torch import nn
from torch.nn import Parameter

from ._lib import check, load, ptr, require_gpu, stream_ptr
from .graph import CSRRecipe, GraphBatch


# ``padded_edges=True`` (an explicit argument of topk_pool_batch / TopKPooling; net1.Net_1 sets it) AND the sizes of the
# batch's graphs known on the host (GraphBatch.sizes): TopKPooling makes no device read, see _select.  The edge_index it
# returns is then PADDED to the input's length with (-1, -1) columns -- NOT PyG's contract, which is why it is opt-in.
NO_SYNC = True
LDS_SORT_MAX_NODES = 16384       # npi_topk_select sorts one graph's (score, index) keys in LDS; larger graphs: radix selection


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def graph_ptr(batch: torch.Tensor, num_graphs: Optional[int] = None) -> torch.Tensor:
    """int32 ``[B+1]`` segment starts of a PyG ``batch`` vector (non-decreasing int64).  ``num_graphs=None``: one device
    read (``batch[-1] + 1``, as PyG's ``global_*_pool`` do)."""
    dev = require_gpu(batch)
    if batch.dtype != torch.int64:
        raise TypeError("batch must be a LongTensor")
    N = batch.numel()
    B = int(num_graphs) if num_graphs is not None else (int(batch[-1].item()) + 1 if N else 0)
    gp = torch.empty(B + 1, dtype=torch.int32, device=dev)
    check(load().npi_graph_bounds(ptr(batch.contiguous()), N, B, ptr(gp), stream_ptr(dev)), "npi_graph_bounds")
    return gp


def _kept_sizes(sizes: Optional[torch.Tensor], ratio: float) -> Optional[torch.Tensor]:
    """ceil(ratio n_g) per graph in the selection kernel's f32 arithmetic, on the host"""
    if not NO_SYNC or sizes is None or sizes.numel() == 0:
        return None
    return torch.ceil(torch.tensor(float(ratio), dtype=torch.float32) * sizes.to(torch.float32)).to(torch.int64)


class _Selection:
    """What the index half of TopKPooling (no gradients: PyG's selection is not differentiable either) leaves for the
    gather, for the backward and for the pooled ``GraphBatch``."""
    __slots__ = ("score", "perm", "remap", "out_ptr", "n_out", "num_graphs", "batch_in", "batch_out", "perm64", "edge_index",
                 "kept", "recipe")


def _select(gb: GraphBatch, weight: torch.Tensor, ratio: float, padded_edges: bool) -> _Selection:
    """scores, per-graph top-k, filter_adj and the sizes of the outputs"""
    lib = load()
    x, edge_index, batch = gb.x, gb.edge_index, gb.batch_vector()
    dev = require_gpu(x, edge_index, batch, weight)
    x = _f32(x.detach())
    w = _f32(weight.detach().reshape(-1))
    N, F = x.shape
    st = stream_ptr(dev)
    i32 = dict(dtype=torch.int32, device=dev)
    gp = gb.segment_ptr()
    B = gp.numel() - 1
    batch = batch.contiguous()
    score = torch.empty(N, dtype=torch.float32, device=dev)
    check(lib.npi_topk_score(ptr(x), x.stride(0), ptr(w), N, F, ptr(score), st), "npi_topk_score")
    out_ptr = torch.empty(B + 1, **i32)
    perm = torch.empty(max(N, 1), **i32)
    remap = torch.empty(max(N, 1), **i32)
    status = torch.empty(1, **i32)
    sizes = gb.sizes                                         # host-known graph sizes: the largest one picks the kernels
    max_nodes = int(sizes.max()) if sizes is not None and sizes.numel() == B and B > 0 else 0

    def select_sorted():
        """graphs above the LDS sort's 16,384 nodes: two device-wide stable radix sorts (npi_topk_select_sorted)"""
        n_ws = int(lib.npi_topk_sorted_workspace_bytes(N))
        wsb = torch.empty(n_ws, dtype=torch.uint8, device=dev)
        check(lib.npi_topk_select_sorted(ptr(score), ptr(batch), ptr(gp), N, B, float(ratio), ptr(out_ptr),
                                         ptr(perm), ptr(remap), ptr(wsb), n_ws, st), "npi_topk_select_sorted")
    if max_nodes > LDS_SORT_MAX_NODES:                      # known on the host: straight to the radix selection
        status.zero_()
        select_sorted()
    else:
        check(lib.npi_topk_select(ptr(score), ptr(gp), N, B, float(ratio), ptr(out_ptr), ptr(perm), ptr(remap),
                                     ptr(status), max_nodes, st), "npi_topk_select")
    # filter_adj needs only the old->new id map, so it runs before the sizes are known
    E = edge_index.size(1)
    src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
    out_ei = torch.empty((2, max(E, 1)), dtype=torch.int64, device=dev)
    count = torch.empty(1, **i32)
    ws = torch.empty(int(lib.npi_filter_adj_workspace_elems(E)), **i32)
    # Host-known graph sizes (``GraphBatch.sizes``, from net1.KeyLoader or from the previous pooling layer): the kept
    # node count is ceil(ratio n_g) per graph -- computable on the host -- and the surviving edges stay in an array of the
    # input's length whose tail is (-1, -1) padding (npi_filter_adj), which every consumer drops.  No device read at
    # all: the layer, and with it the whole Net_1 step, runs without a host synchronisation and captures into a HIP graph.
    kept = _kept_sizes(sizes, ratio) if (B > 0 and padded_edges) else None
    nosync = kept is not None and kept.numel() == B
    check(lib.npi_filter_adj(ptr(src), ptr(dst), E, ptr(remap), ptr(out_ei[0]), ptr(out_ei[1]), ptr(count), ptr(ws),
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
            check(lib.npi_filter_adj(ptr(src), ptr(dst), E, ptr(remap), ptr(out_ei[0]), ptr(out_ei[1]), ptr(count), ptr(ws), 0, st),
                  "npi_filter_adj")
            n_out, e_out = (int(v) for v in torch.cat([out_ptr[-1:], count]).tolist())
    sel = _Selection()
    sel.score, sel.perm, sel.remap, sel.out_ptr = score, perm[:n_out], remap, out_ptr
    sel.n_out, sel.num_graphs, sel.batch_in, sel.kept = int(n_out), B, batch, kept
    # written by the gather launch together with the kept rows
    sel.batch_out = torch.empty(n_out, dtype=torch.int64, device=dev)
    sel.perm64 = torch.empty(n_out, dtype=torch.int64, device=dev)           # the LongTensor PyG returns
    sel.edge_index = out_ei[:, :e_out]
    # The conv in front of this layer left the CSR of `edge_index` in the batch (GraphBatch.graph): the pooled graph's CSR is
    # that one filtered -- four launches, no sort.  Only the recipe is written here: it is carried out when (and only if) a
    # conv asks for the pooled graph -- the last pooling layer's graph feeds no conv at all.
    sel.recipe = None
    parent = gb.peek_graph()                                # None if the edge list was written to since its CSR was built
    if nosync and parent is not None and parent.self_loops:
        off = int(lib.npi_filter_adj_newpos_offset(E))
        sel.recipe = CSRRecipe(parent.by_dst, sel.perm, remap, ws[off:off + E], sel.n_out, int(e_out),
                               sel.edge_index._version)
    return sel


def _gather(x: torch.Tensor, sel: _Selection):
    """``x[perm] * score[perm]``, ``score[perm]`` -- and, in the same launch, ``batch[perm]`` and ``perm`` as int64"""
    dev = x.device
    x = _f32(x.detach())
    F = x.size(1)
    xo = torch.empty((sel.n_out, F), dtype=torch.float32, device=dev)
    score_o = torch.empty(sel.n_out, dtype=torch.float32, device=dev)
    check(load().npi_topk_gather(ptr(x), x.stride(0), ptr(sel.score), ptr(sel.batch_in), ptr(sel.perm), ptr(sel.out_ptr),
                                    sel.num_graphs, F, sel.n_out, ptr(xo), xo.stride(0), ptr(sel.batch_out), ptr(score_o),
                                    ptr(sel.perm64), stream_ptr(dev)), "npi_topk_gather")
    return xo, score_o


class _TopKGatherFn(torch.autograd.Function):
    """The differentiable half: gradients to the kept rows of ``x`` and, through the scores, to ``weight``."""

    @staticmethod
    def forward(ctx, x, weight, sel):
        xo, score_o = _gather(x, sel)
        ctx.save_for_backward(x.detach(), weight.detach(), sel.score, sel.perm, sel.remap)
        # no zero tensor for the gradient of an output nobody used (score[perm])
        ctx.set_materialize_grads(False)
        return xo, score_o

    @staticmethod
    def backward(ctx, dxo, dscore_o):
        x, weight, score, perm, remap = ctx.saved_tensors
        lib = load()
        dev = x.device
        x = _f32(x)
        w = _f32(weight.reshape(-1))
        N, F = x.shape
        n_out = perm.numel()
        st = stream_ptr(dev)
        if dxo is None and dscore_o is None:
            return None, None, None
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
        return (dx if ctx.needs_input_grad[0] else None), dw, None


def topk_pool_batch(gb: GraphBatch, weight: torch.Tensor, ratio: float = 0.5, padded_edges: bool = False):
    """``TopKPooling.forward`` on a ``GraphBatch`` -> ``(pooled GraphBatch, perm, score[perm])``; differentiable in
    ``gb.x`` and ``weight``.

    ``padded_edges=False`` (default): PyG's contract -- the pooled ``edge_index`` is exactly the surviving edges (one device
    read per call for the two data-dependent sizes).  ``padded_edges=True`` and ``gb.sizes`` present (the per-graph node
    counts on the host, net1.KeyLoader): NO device read -- the pooled ``edge_index`` then keeps the INPUT's length, the
    surviving edges first, in order, and a tail of ``(-1, -1)`` columns.  Only this package's consumers (the convs' CSR
    build, the next ``filter_adj``, ``entry_weights``) drop that tail; ``x[edge_index[0]]`` or ``edge_index.size(1)``
    in foreign code would not -- hence opt-in.  The pooled batch carries its segment starts, its host sizes (when the
    input's were known) and, when a conv had built the input's CSR, the recipe for its own."""
    x = gb.x
    sel = _select(gb, weight, ratio, padded_edges)
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        xo, score_o = _TopKGatherFn.apply(x, weight, sel)
    else:
        xo, score_o = _gather(x, sel)
    # the kept-row offsets ARE the segment starts of the pooled batch vector: the readout and the next pooling layer take
    # them from here instead of searching the batch vector again; both directions of a pair survive or fall together
    out = GraphBatch(xo, sel.edge_index, sel.batch_out, sel.num_graphs, sizes=sel.kept, graph_ptr=sel.out_ptr,
                     symmetric=gb.symmetric, recipe=sel.recipe)
    return out, sel.perm64, score_o


def topk_pool(x: torch.Tensor, edge_index: torch.Tensor, batch: torch.Tensor, weight: torch.Tensor,
              ratio: float = 0.5, num_graphs: Optional[int] = None):
    """PyG's tensor form: ``(x', edge_index', None, batch', perm, score[perm])`` with exactly PyG's contract (the surviving
    edges, nothing padded)."""
    out, perm, score = topk_pool_batch(GraphBatch(x, edge_index, batch, num_graphs), weight, ratio)
    return out.x, out.edge_index, None, out.batch, perm, score


def _readout_fwd(x: torch.Tensor, gp: torch.Tensor):
    dev = require_gpu(x, gp)
    x = _f32(x.detach())
    B, F = gp.numel() - 1, x.size(1)
    out = torch.empty((B, 2 * F), dtype=torch.float32, device=dev)
    check(load().npi_readout_max_mean(ptr(x), x.stride(0), ptr(gp), B, F, ptr(out), stream_ptr(dev)),
          "npi_readout_max_mean")
    return out, x


class _ReadoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gp):
        out, xc = _readout_fwd(x, gp)
        ctx.save_for_backward(xc, gp, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gp, out = ctx.saved_tensors
        dev = x.device
        dout = _f32(dout)
        B, F = gp.numel() - 1, x.size(1)
        if B == 0:
            return torch.zeros_like(x), None
        dx = torch.empty_like(x)                  # rows outside every graph (none in PyG batches): zeroed by the kernel
        check(load().npi_readout_max_mean_bwd_ex(ptr(x), x.stride(0), ptr(gp), B, F, ptr(out), ptr(dout), ptr(dx),
                                                 dx.stride(0), x.size(0), stream_ptr(dev)), "npi_readout_max_mean_bwd")
        return dx, None


def global_max_mean_pool(x, batch: Optional[torch.Tensor] = None, num_graphs: Optional[int] = None) -> torch.Tensor:
    """``cat([global_max_pool(x, batch), global_mean_pool(x, batch)], dim=1)`` -> ``[B, 2F]``; differentiable in ``x``.
    ``x`` may be a ``GraphBatch`` (its features over its segments; the segment starts a pooling layer left are reused)."""
    if isinstance(x, GraphBatch):
        gp, x = x.segment_ptr(), x.x
    else:
        gp = graph_ptr(batch, num_graphs)
    if torch.is_grad_enabled() and x.requires_grad:
        return _ReadoutFn.apply(x, gp)
    return _readout_fwd(x, gp)[0]


def global_max_pool(x, batch, size=None):
    return global_max_mean_pool(x, batch, size)[:, : x.size(1)]


def global_mean_pool(x, batch, size=None):
    return global_max_mean_pool(x, batch, size)[:, x.size(1):]


class TopKPooling(nn.Module):
    """``TopKPooling(in_channels, ratio=0.5)``: parameter ``weight [1, in_channels]`` (PyG 1.4.2 layout,
    so ``pool1.weight`` of the reference checkpoints loads unchanged)."""

    def __init__(self, in_channels: int, ratio: float = 0.5, padded_edges: bool = False, **kwargs):
        """``padded_edges``: see ``topk_pool_batch`` -- with host-known graph sizes (``GraphBatch.sizes``) the layer reads
        nothing back and returns an edge list of the input's length with a ``(-1, -1)`` tail (for pipelines made of this
        package's layers only)."""
        super().__init__()
        self.in_channels, self.ratio, self.padded_edges = in_channels, ratio, bool(padded_edges)
        self.weight = Parameter(torch.empty(1, in_channels))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        bound = 1.0 / math.sqrt(self.in_channels)
        self.weight.data.uniform_(-bound, bound)

    def forward(self, x, edge_index=None, edge_attr=None, batch=None):
        """PyG's tensors in, PyG's 6-tuple out; or a ``GraphBatch`` in, ``(pooled GraphBatch, perm, score[perm])`` out
        (``padded_edges`` applies to this form only: it needs ``GraphBatch.sizes``)."""
        if edge_attr is not None:
            raise NotImplementedError("TopKPooling: edge_attr is not used by NPI-GNN")
        if isinstance(x, GraphBatch):
            if edge_index is not None or batch is not None:
                raise TypeError("TopKPooling: a GraphBatch carries its own edge_index and batch")
            return topk_pool_batch(x, self.weight, self.ratio, self.padded_edges)
        if batch is None:
            batch = torch.zeros(x.size(0), dtype=torch.int64, device=x.device)
        return topk_pool(x, edge_index, batch, self.weight, self.ratio)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, ratio={self.ratio})"
