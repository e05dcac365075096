"""One conv layer across the GPUs of a node: one process per GPU, RCCL over xGMI.

The reference is single-process (no distributed code); this is the build's own
multi-GPU form of the same layer (BASELINE.json configs[3]).

Partition: node i is OWNED by rank ``i % W`` as local row ``i // W`` (strided ownership).  On the
bipartite NPI graphs, where ncRNA ids come first and the ~10x fewer, ~10x heavier protein ids last,
striding gives every rank the same share of light and heavy rows, so rows AND entries balance
(a contiguous row split puts all proteins on one rank: 7x entry imbalance at C4).

Each rank keeps the CSR rows of the nodes it owns, with column ids that index the ALL-GATHERED
feature table (rank-major: row ``(j % W) * n_per + j // W``), so the gathered buffer is used as it
arrives -- no re-packing copy.  Per layer:

  forward :  table = all_gather(x_local)            [W*n_per, F]   <- the one exchange step
             agg   = segsum_mean(by_dst_local, table)               (same kernel as single GPU)
             out   = agg @ W + b                                     (local rows only)
  backward:  dagg  = (dOut @ W^T) / cnt                              (local rows)
             table = all_gather(dagg)                                <- exchange, overlapped with:
             dW,db = agg^T dOut, colsum(dOut);  all_reduce(dW, db)   (512 KiB)
             dX    = segsum(by_src_local, table)

No reduction of partial node embeddings is needed (a destination-row split computes whole rows),
which is what replaces the north-star's "all-reduce of partial embeddings": for an edge split the
[N, F] all-reduce costs 2x the bytes of this all-gather (SURVEY.md 8(e) cost table).

The local compute is a small backend object so that the partition + exchange logic can be
exercised on CPU with gloo (tests/test_dist_gloo.py injects a torch backend); the product backend
is ``HipBackend`` and there is no CPU fallback in this package.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn


class StridedPartition:
    def __init__(self, num_nodes: int, world: int):
        self.N, self.W = int(num_nodes), int(world)
        self.n_per = (self.N + self.W - 1) // self.W          # padded rows per rank in the gathered table

    def n_local(self, rank: int) -> int:
        return (self.N - rank + self.W - 1) // self.W if rank < self.N else 0

    def table_rows(self) -> int:
        return self.W * self.n_per

    def padded(self, ids: torch.Tensor) -> torch.Tensor:
        """global node id -> row of the all-gathered (rank-major, padded) table"""
        return (ids % self.W) * self.n_per + ids // self.W

    def shard(self, x_full: torch.Tensor, rank: int) -> torch.Tensor:
        return x_full[rank::self.W]

    def unshard(self, parts) -> torch.Tensor:
        out = torch.empty((self.N,) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype, device=parts[0].device)
        for r, p in enumerate(parts):
            out[r::self.W] = p
        return out


def local_edges(edge_index: torch.Tensor, part: StridedPartition, rank: int):
    """(key, val) of this rank's two CSR sides: keys are local rows, values rows of the gathered
    table; global self loops are removed here (the builder appends one loop per row)."""
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    md = keep & (dst % part.W == rank)
    ms = keep & (src % part.W == rank)
    by_dst = ((dst[md] // part.W).contiguous(), part.padded(src[md]).contiguous())
    by_src = ((src[ms] // part.W).contiguous(), part.padded(dst[ms]).contiguous())
    return by_dst, by_src


def all_gather_rows(x_local: torch.Tensor, part: StridedPartition, group=None, async_op: bool = False):
    """[n_local, F] -> [W * n_per, F] rank-major table (one pad row on the short ranks)."""
    F = x_local.size(1)
    if x_local.size(0) != part.n_per:
        buf = x_local.new_zeros((part.n_per, F))
        buf[: x_local.size(0)] = x_local
    else:
        buf = x_local.contiguous()
    table = x_local.new_empty((part.table_rows(), F))
    if part.W == 1:
        table.copy_(buf)
        return table, None
    if dist.get_backend(group) == "nccl":
        work = dist.all_gather_into_tensor(table, buf, group=group, async_op=async_op)
    else:
        chunks = list(table.view(part.W, part.n_per, F).unbind(0))
        work = dist.all_gather(chunks, buf, group=group, async_op=async_op)
    return table, (work if async_op else None)


class HipBackend:
    """Local compute of one rank on its MI355X through the C ABI."""

    def __init__(self, by_dst, by_src, n_local: int, table_rows: int, loop_col_offset: int):
        from .graph import build_side
        self.dst = build_side(by_dst[0], by_dst[1], n_local, table_rows, True, loop_col_offset, False)
        self.src = build_side(by_src[0], by_src[1], n_local, table_rows, True, loop_col_offset, False)
        self.local_nnz = int(by_dst[0].numel()) + n_local

    def aggregate_mean(self, table):
        from . import functional as NF
        return NF.segsum(None, self.dst, table, mean=True)

    def aggregate_t(self, table):
        from . import functional as NF
        return NF.segsum(None, self.src, table)

    def inv_count(self):
        return self.dst.inv_count()

    def linear_fwd(self, a, w, b):
        from . import functional as NF
        return NF.linear_fwd(a, w, b)

    def linear_bwd_data(self, dc, w, rowscale):
        from . import functional as NF
        return NF.linear_bwd_data(dc, w, rowscale)

    def linear_bwd_weight(self, a, dc, want_bias):
        from . import functional as NF
        return NF.linear_bwd_weight(a, dc, want_bias)


class ShardedGraph:
    """This rank's shard of the (self-loop-augmented) graph."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, rank: int, world: int, device,
                 backend_factory=None, group=None):
        self.part = StridedPartition(num_nodes, world)
        self.rank, self.world, self.group = rank, world, group
        self.n_local = self.part.n_local(rank)
        ei = edge_index.to(device)
        by_dst, by_src = local_edges(ei, self.part, rank)
        factory = backend_factory or HipBackend
        self.backend = factory(by_dst, by_src, self.n_local, self.part.table_rows(), rank * self.part.n_per)
        self.local_nnz = int(by_dst[0].numel()) + self.n_local

    def shard(self, x_full: torch.Tensor) -> torch.Tensor:
        return self.part.shard(x_full, self.rank)


class _ShardedSageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, weight, bias, sg: ShardedGraph):
        be = sg.backend
        table, _ = all_gather_rows(x_local, sg.part, sg.group)
        agg = be.aggregate_mean(table)
        out = be.linear_fwd(agg, weight, bias)
        ctx.sg = sg
        ctx.has_bias = bias is not None
        ctx.save_for_backward(agg, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        agg, weight = ctx.saved_tensors
        sg: ShardedGraph = ctx.sg
        be = sg.backend
        grad_out = grad_out.contiguous()
        dx = dw = db = None
        work = table = None
        if ctx.needs_input_grad[0]:
            dagg = be.linear_bwd_data(grad_out, weight, be.inv_count())
            table, work = all_gather_rows(dagg, sg.part, sg.group, async_op=sg.world > 1)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = be.linear_bwd_weight(agg, grad_out, ctx.has_bias)     # overlaps the all-gather
            if sg.world > 1:
                dist.all_reduce(dw, group=sg.group)
                if db is not None:
                    dist.all_reduce(db, group=sg.group)
        if ctx.needs_input_grad[0]:
            if work is not None:
                work.wait()
            dx = be.aggregate_t(table)
        return dx, dw, db, None


class ShardedSAGELayer(nn.Module):
    """SAGEConv (PyG 1.4.2 semantics, mean over in-neighbours and self, then ``@ W + b``) on a sharded
    graph.  Input and output are this rank's rows (nodes ``rank, rank + W, ...``); parameters are
    replicated and their gradients all-reduced, as data-parallel training expects."""

    def __init__(self, sg: ShardedGraph, weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
        super().__init__()
        self.sg = sg
        self.weight = nn.Parameter(weight.clone())
        self.bias = nn.Parameter(bias.clone()) if bias is not None else None

    def forward(self, x_local: torch.Tensor) -> torch.Tensor:
        return _ShardedSageFn.apply(x_local, self.weight, self.bias, self.sg)
