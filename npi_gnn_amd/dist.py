"""One conv layer across the GPUs of a node: one process per GPU, RCCL over xGMI.

The reference is single-process (no distributed code); this is the build's own
multi-GPU form of the same layer (BASELINE.json configs[3]).

Partition: a vertex cut with REPLICATED HUBS.  Nodes are split into hubs H (on the ncRNA-protein
graphs: the protein side, ~10x fewer and ~10x heavier nodes) and light nodes L (the ncRNAs).
Both sets are owned in strides (the k-th hub by rank ``k % W``, the k-th light node likewise), so
every rank holds the same share of light and heavy rows.  No light-light edge may exist (bipartite
graphs have none; ``auto_hubs`` promotes one endpoint of any such edge).  Per edge ``j -> i``:

  i light            computed by owner(i): x_j is local (never: j light) or a hub row      -> side A
  i hub,  j hub      computed by owner(i) from the gathered hub table                      -> side A
  i hub,  j light    PARTIAL sum computed by owner(j), reduce-scattered to owner(i)        -> side B

so the only rows that ever cross xGMI are hub rows: one all-gather of the hub features and one
reduce-scatter of the partial hub sums per direction -- 2 x |H| x F floats instead of the
|N| x F all-gather a plain destination-row split needs (C4: 0.2 GB instead of 1.02 GB per
direction).  With every node a hub (``hub_mask=None``) side B is empty and the scheme IS the plain
destination-row split with an all-gather of all rows; that is the fallback for graphs without a
small hub side.

Local row order on rank r: its light nodes first, then the hubs it owns.  Side A reads a table
``[gathered hub rows (W * h_per, rank-major, padded) ; local light rows]`` that the all-gather
fills in place; side B reads the rank's own rows directly, so it runs while the all-gather is in
flight, and the reduce-scatter of its result overlaps side A and (backward) the weight-gradient GEMM:

  forward :  hubs  = all_gather(x_own[hub rows])                     | part = segsum(B, x_own)
             hsum  = reduce_scatter(part)                            | agg  = segsum_mean(A, table)
             agg[hub rows] = (agg * cnt_A + hsum) / cnt ;  out = agg @ W + b
  backward:  dagg  = (dOut @ W^T) / cnt
             hubs  = all_gather(dagg[hub rows])                      | part = segsum(B^T, dagg)
             hsum  = reduce_scatter(part)                            | dW, db (+ all_reduce, 256 KiB)
             dX    = segsum(A^T, table) ;  dX[hub rows] += hsum

Sums of partials arrive in RCCL's order, so multi-GPU results match the single-GPU ones to fp32
rounding (tests: 1e-5), not bit for bit.

The local compute is a small backend object so that the partition + exchange logic can be
exercised on CPU with gloo (tests/test_dist_gloo.py injects a torch backend); the product backend
is ``HipBackend`` and there is no CPU fallback in this package.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import nn


def auto_hubs(edge_index: torch.Tensor, num_nodes: int, ratio: float = 2.0, max_fraction: float = 0.5):
    """Hub mask from degrees: nodes with total degree > ``ratio`` x mean, plus the heavier endpoint of
    every remaining light-light edge.  Returns None (all rows exchanged) when hubs would exceed
    ``max_fraction`` of the nodes."""
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    src, dst = src[keep], dst[keep]
    deg = torch.bincount(src, minlength=num_nodes) + torch.bincount(dst, minlength=num_nodes)
    hub = deg.float() > ratio * deg.float().mean()
    ll = ~hub[src] & ~hub[dst]
    if bool(ll.any()):
        s, d = src[ll], dst[ll]
        hub[torch.where(deg[s] > deg[d], s, d)] = True
    if int(hub.sum()) > max_fraction * num_nodes:
        return None
    return hub


class HubPartition:
    """Strided ownership of light nodes and of hubs; see the module docstring."""

    def __init__(self, num_nodes: int, world: int, hub_mask: Optional[torch.Tensor] = None, device=None):
        self.N, self.W = int(num_nodes), int(world)
        if hub_mask is None:
            hub_mask = torch.ones(self.N, dtype=torch.bool, device=device)
        self.hub = hub_mask.to(device=device, dtype=torch.bool)
        if self.hub.numel() != self.N:
            raise ValueError("hub_mask must have one entry per node")
        h = torch.cumsum(self.hub, 0) - 1
        l = torch.cumsum(~self.hub, 0) - 1
        self.index = torch.where(self.hub, h, l)                  # rank among hubs / among light nodes
        self.nH = int(self.hub.sum())
        self.nL = self.N - self.nH
        self.h_per = (self.nH + self.W - 1) // self.W             # padded hub rows per rank
        self.hub_rows = self.W * self.h_per

    def _count(self, total: int, rank: int) -> int:
        return (total - rank + self.W - 1) // self.W if rank < total else 0

    def n_light(self, rank: int) -> int:
        return self._count(self.nL, rank)

    def n_hub(self, rank: int) -> int:
        return self._count(self.nH, rank)

    def owner(self, ids: torch.Tensor) -> torch.Tensor:
        return self.index[ids] % self.W

    def local(self, ids: torch.Tensor) -> torch.Tensor:
        """position inside the owner's light block (light ids) or hub block (hub ids)"""
        return self.index[ids] // self.W

    def hub_row(self, ids: torch.Tensor) -> torch.Tensor:
        """hub id -> row of the all-gathered (rank-major, padded) hub table"""
        k = self.index[ids]
        return (k % self.W) * self.h_per + k // self.W

    def own_ids(self, rank: int) -> torch.Tensor:
        """global ids of the rows of rank ``rank`` in local order (light nodes, then hubs)"""
        ids = torch.arange(self.N, device=self.hub.device)
        mine = self.index % self.W == rank
        return torch.cat([ids[mine & ~self.hub], ids[mine & self.hub]])

    def shard(self, x_full: torch.Tensor, rank: int) -> torch.Tensor:
        return x_full[self.own_ids(rank).to(x_full.device)]

    def unshard(self, parts) -> torch.Tensor:
        out = torch.empty((self.N,) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype, device=parts[0].device)
        for r, p in enumerate(parts):
            out[self.own_ids(r).to(p.device)] = p
        return out


def local_sides(src: torch.Tensor, dst: torch.Tensor, part: HubPartition, rank: int):
    """(key, val, n_rows, n_cols) of side A and side B of rank ``rank`` for messages ``src -> dst``
    (call with the two swapped for the transposed sides).  Global self loops are dropped and one
    loop per local row is appended LAST, which is where add_remaining_self_loops puts it."""
    dev = src.device
    keep = src != dst
    src, dst = src[keep], dst[keep]
    hub_s, hub_d = part.hub[src], part.hub[dst]
    if bool((~hub_s & ~hub_d).any()):
        raise ValueError("HubPartition: an edge joins two light nodes; mark one endpoint as a hub (auto_hubs)")
    nL, nH = part.n_light(rank), part.n_hub(rank)
    mine_d = part.owner(dst) == rank
    # side A: rows = local order; table = [hub table ; local light rows]
    a = mine_d & hub_s                                             # (light or hub) <- hub
    key_a = torch.where(hub_d[a], nL + part.local(dst[a]), part.local(dst[a]))
    val_a = part.hub_row(src[a])
    rows = torch.arange(nL + nH, device=dev)
    loop_col = torch.cat([part.hub_rows + rows[:nL], rank * part.h_per + rows[:nH]])
    side_a = (torch.cat([key_a, rows]), torch.cat([val_a, loop_col]), nL + nH, part.hub_rows + nL)
    # side B: rows = hub table rows; sources = this rank's light rows, read from its own block
    b = hub_d & ~hub_s & (part.owner(src) == rank)
    side_b = (part.hub_row(dst[b]), part.local(src[b]), part.hub_rows, nL + nH)
    return side_a, side_b


# tests set this to push a world-size-1 run through RCCL as well (the one-GPU box's only way to
# exercise the real collectives); normally a single rank just copies
ALWAYS_COMMUNICATE = False

# bench.py sets this to a list: every wait on a collective then leaves (tag, start event, end event) recorded on the
# waiting stream -- the time that stream stood still for the exchange, i.e. the communication that was NOT hidden
# behind side B / dW / the local aggregation (exposed-comm time is reported separately).
_COMM_PROFILE = None


def _wait(work, tag: str, ref: torch.Tensor) -> None:
    if work is None:
        return
    prof = _COMM_PROFILE
    if prof is None or not ref.is_cuda:
        work.wait()
        return
    s = torch.cuda.current_stream(ref.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    work.wait()
    e1.record(s)
    prof.append((tag, e0, e1))


def _solo(world: int) -> bool:
    return world == 1 and not ALWAYS_COMMUNICATE


def _portable(group) -> bool:
    return dist.get_backend(group) != "nccl"


def all_gather_rows(block: torch.Tensor, out: torch.Tensor, world: int, group=None, async_op: bool = False):
    """block [h_per, F] of every rank -> out [W * h_per, F] (rank-major)"""
    if _solo(world):
        out.copy_(block)
        return None
    if _portable(group):                                           # gloo (CPU tests, one-GPU test rigs)
        work = dist.all_gather(list(out.view(world, block.size(0), -1).unbind(0)), block, group=group,
                               async_op=async_op)
    else:
        work = dist.all_gather_into_tensor(out, block, group=group, async_op=async_op)
    return work if async_op else None


class _Done:
    def wait(self):
        return True


def reduce_scatter_rows(part_sums: torch.Tensor, out: torch.Tensor, rank: int, world: int, group=None,
                        async_op: bool = False):
    """part_sums [W * h_per, F] of every rank -> out [h_per, F] = sum over ranks of block ``rank``"""
    if _solo(world):
        out.copy_(part_sums)
        return None
    if _portable(group):
        dist.all_reduce(part_sums, group=group)
        out.copy_(part_sums.view(world, out.size(0), -1)[rank])
        return _Done() if async_op else None
    work = dist.reduce_scatter_tensor(out, part_sums, group=group, async_op=async_op)
    return work if async_op else None


class HipBackend:
    """Local compute of one rank on its MI355X through the C ABI."""

    def make_side(self, key, val, n_rows: int, n_cols: int):
        from .graph import build_side
        return build_side(key.contiguous(), val.contiguous(), n_rows, n_cols, False, 0, False)

    def segsum(self, side, table, mean: bool = False):
        from . import functional as NF
        return NF.segsum(None, side, table, mean=mean)

    def linear_fwd(self, a, w, b):
        from . import functional as NF
        return NF.linear_fwd(a, w, b)

    def linear_bwd_data(self, dc, w, rowscale):
        from . import functional as NF
        return NF.linear_bwd_data(dc, w, rowscale)

    def linear_bwd_weight(self, a, dc, want_bias, shared=False):
        from . import functional as NF
        return NF.linear_bwd_weight(a, dc, want_bias, shared=shared)

    def side_stream(self, like):
        """second HIP stream for the weight-gradient GEMM, or None when the shard is too small to gain"""
        from . import functional as NF
        if not NF.OVERLAP_STREAMS or like.size(0) < NF.OVERLAP_MIN_ROWS:
            return None
        return NF._side_stream(like.device)


class ShardedGraph:
    """This rank's shard of the (self-loop-augmented) graph: sides A, B and their transposes."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, rank: int, world: int, device,
                 backend=None, group=None, hub_mask: Optional[torch.Tensor] = None):
        self.part = part = HubPartition(num_nodes, world, hub_mask, device)
        self.rank, self.world, self.group = rank, world, group
        self.nL, self.nH = part.n_light(rank), part.n_hub(rank)
        self.n_local = self.nL + self.nH
        self.exchange_partials = part.nL > 0                      # same answer on every rank
        self.backend = be = backend or HipBackend()
        ei = edge_index.to(device)
        src, dst = ei[0], ei[1]
        a, b = local_sides(src, dst, part, rank)
        at, bt = local_sides(dst, src, part, rank)
        self.A, self.At = be.make_side(*a), be.make_side(*at)
        self.B = be.make_side(*b) if self.exchange_partials else None
        self.Bt = be.make_side(*bt) if self.exchange_partials else None
        self.local_nnz = int(a[0].numel()) + int(b[0].numel())    # entries this rank walks per direction
        # 1 / (in-degree + 1) of the local rows; for hub rows also the share of it that side A holds
        keep = src != dst
        cnt = torch.bincount(dst[keep], minlength=num_nodes).to(torch.float32) + 1.0
        own = part.own_ids(rank)
        self.inv_cnt = (1.0 / cnt[own]).contiguous()
        self.cnt_a_hub = torch.bincount(a[0], minlength=self.n_local)[self.nL:].to(torch.float32).view(-1, 1)
        self.own = own

    def shard(self, x_full: torch.Tensor) -> torch.Tensor:
        return x_full[self.own.to(x_full.device)]


def _exchange_start(sg: ShardedGraph, rows: torch.Tensor, side_b):
    """Launch the all-gather of the hub rows of ``rows`` and the reduce-scatter of side B's partial
    hub sums.  Returns (table for side A, its pending work, reduced hub sums or None, its pending work)."""
    part, be, W = sg.part, sg.backend, sg.world
    F = rows.size(1)
    table = rows.new_empty((part.hub_rows + sg.nL, F))
    block = rows[sg.nL:]
    if sg.nH != part.h_per:                                        # short rank: one zero pad row
        block = rows.new_zeros((part.h_per, F))
        block[: sg.nH] = rows[sg.nL:]
    g_work = all_gather_rows(block.contiguous(), table[: part.hub_rows], W, sg.group, async_op=not _solo(W))
    table[part.hub_rows:] = rows[: sg.nL]
    hsum = r_work = None
    if sg.exchange_partials:
        partial = be.segsum(side_b, rows)                          # no remote input: overlaps the all-gather
        hsum = rows.new_empty((part.h_per, F))
        r_work = reduce_scatter_rows(partial, hsum, sg.rank, W, sg.group, async_op=not _solo(W))
    return table, g_work, hsum, r_work


class _ShardedSageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_own, weight, bias, sg: ShardedGraph):
        be = sg.backend
        x_own = x_own.contiguous()
        table, g_work, hsum, r_work = _exchange_start(sg, x_own, sg.B)
        _wait(g_work, "fwd_all_gather", table)
        agg = be.segsum(sg.A, table, mean=True)
        if hsum is not None:
            _wait(r_work, "fwd_reduce_scatter", hsum)
            if sg.nH:
                inv = sg.inv_cnt[sg.nL:].view(-1, 1)
                agg[sg.nL:] = (agg[sg.nL:] * sg.cnt_a_hub + hsum[: sg.nH]) * inv
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
        started = None
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.needs_input_grad[0]:
            dagg = be.linear_bwd_data(grad_out, weight, sg.inv_cnt)
            started = _exchange_start(sg, dagg, sg.Bt)
        # dW is independent of the dX chain.  On the GPU backend it is launched here, on this stream, so that it is
        # resident before the aggregation -- sent to a second stream -- fills the CUs (see functional._SageConvFn);
        # other backends run everything in line.
        side = be.side_stream(grad_out) if (want_w and started is not None and hasattr(be, "side_stream")) else None
        main = torch.cuda.current_stream(grad_out.device) if side is not None else None
        if side is not None:
            side.wait_stream(main)
        if want_w:
            dw, db = be.linear_bwd_weight(agg, grad_out, ctx.has_bias, shared=True) if side is not None else \
                be.linear_bwd_weight(agg, grad_out, ctx.has_bias)
        if started is not None:
            table, g_work, hsum, r_work = started

            def finish():
                _wait(g_work, "bwd_all_gather", table)
                out = be.segsum(sg.At, table)
                if hsum is not None:
                    _wait(r_work, "bwd_reduce_scatter", hsum)
                    if sg.nH:
                        out[sg.nL:] += hsum[: sg.nH]
                return out
            if side is not None:
                with torch.cuda.stream(side):
                    dx = finish()
                for t in (table, hsum):
                    if t is not None:
                        t.record_stream(side)
                dx.record_stream(main)
                main.wait_stream(side)
            else:
                dx = finish()
        if want_w and not _solo(sg.world):
            _wait(dist.all_reduce(dw, group=sg.group, async_op=True), "bwd_all_reduce_dw", dw)
            if db is not None:
                _wait(dist.all_reduce(db, group=sg.group, async_op=True), "bwd_all_reduce_db", db)
        return dx, dw, db, None


class ShardedSAGELayer(nn.Module):
    """SAGEConv (PyG 1.4.2 semantics, mean over in-neighbours and self, then ``@ W + b``) on a sharded
    graph.  Input and output are this rank's rows (``ShardedGraph.own``: light nodes, then hubs);
    parameters are replicated and their gradients all-reduced, as data-parallel training expects."""

    def __init__(self, sg: ShardedGraph, weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
        super().__init__()
        self.sg = sg
        self.weight = nn.Parameter(weight.clone())
        self.bias = nn.Parameter(bias.clone()) if bias is not None else None

    def forward(self, x_own: torch.Tensor) -> torch.Tensor:
        return _ShardedSageFn.apply(x_own, self.weight, self.bias, self.sg)
