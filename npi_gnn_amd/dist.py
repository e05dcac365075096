"""One conv layer across the GPUs of a node: one process per GPU, RCCL over xGMI.

The reference is single-process (no distributed code); this is the build's own
multi-GPU form of the same layers (BASELINE.json configs[3] and [4]).

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

Local row order on rank r: its light nodes first, then the hubs it owns.  Side A gathers from a
TWO-PART table ``[gathered hub rows (W * h_per, rank-major, padded) ; the rank's own rows]``: the
all-gather fills the first part, the second part IS the layer input (``npi_segsum_ex``; nothing is
copied behind the received rows).  Side B reads the rank's own rows only, so it runs while the
all-gather is in flight, and the reduce-scatter of its result overlaps side A and (backward) the
weight-gradient GEMM:

  SAGE forward :  hubs  = all_gather(x_own[hub rows])                 | part = segsum(B, x_own)
                  hsum  = reduce_scatter(part)                        | agg  = segsum_mean(A, [hubs ; x_own])
                  agg[hub rows] = (agg * cnt_A + hsum) / cnt ;  out = agg @ W + b
  SAGE backward:  dagg  = (dOut @ W^T) / cnt
                  hubs  = all_gather(dagg[hub rows])                  | part = segsum(B^T, dagg)
                  hsum  = reduce_scatter(part)                        | dW, db (+ all_reduce, 256 KiB)
                  dX    = segsum(A^T, [hubs ; dagg]) ;  dX[hub rows] += hsum

GCNConv is the same exchange at width F_out with the symmetric normalisation as per-entry weights
on both sides (the degrees are global, known to every rank).  GATConv needs the softmax of a hub
row whose sources sit on every rank: each rank computes (max, sum exp) of its part, one small
all-reduce(MAX) makes the row maxima global, the partial weighted sums and exp-sums are then taken
relative to that maximum and simply ADD (reduce-scatter) -- see ``_ShardedGatFn``.

``EdgeShardedGraph`` / ``EdgeShardedSAGELayer`` are the north-star's literal baseline: every GPU
walks a contiguous slice of the target-sorted entry stream over a full replica of ``x`` and the
partial ``[N, F]`` sums are all-reduced (2.04 GB through the ring per direction at C4, against 0.2 GB
for the hub cut) -- kept selectable (``bench.py --partition edges``) so the difference is measured,
not asserted.

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

from .schedule import DEFAULT, Schedule

NEG = -3.0e38      # "no entry" row maximum (what the softmax-statistics kernels start from)


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

    def hub_table_ids(self) -> torch.Tensor:
        """global node id of every row of the hub table (-1 for the pad rows of short ranks)"""
        ids = torch.full((self.hub_rows,), -1, dtype=torch.long, device=self.hub.device)
        h = torch.nonzero(self.hub).flatten()
        ids[self.hub_row(h)] = h
        return ids

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
    loop per local row is appended LAST, which is where add_remaining_self_loops puts it.
    Column ids of side A index the two-part table ``[hub table (hub_rows) ; own rows (nL + nH)]``."""
    dev = src.device
    keep = src != dst
    src, dst = src[keep], dst[keep]
    hub_s, hub_d = part.hub[src], part.hub[dst]
    if bool((~hub_s & ~hub_d).any()):
        raise ValueError("HubPartition: an edge joins two light nodes; mark one endpoint as a hub (auto_hubs)")
    nL, nH = part.n_light(rank), part.n_hub(rank)
    mine_d = part.owner(dst) == rank
    # side A: rows = local order; table = [hub table ; own rows]
    a = mine_d & hub_s                                             # (light or hub) <- hub
    key_a = torch.where(hub_d[a], nL + part.local(dst[a]), part.local(dst[a]))
    val_a = part.hub_row(src[a])
    rows = torch.arange(nL + nH, device=dev)
    loop_col = torch.cat([part.hub_rows + rows[:nL], rank * part.h_per + rows[:nH]])
    side_a = (torch.cat([key_a, rows]), torch.cat([val_a, loop_col]), nL + nH, part.hub_rows + nL + nH)
    # side B: rows = hub table rows; sources = this rank's light rows, read from its own block
    b = hub_d & ~hub_s & (part.owner(src) == rank)
    side_b = (part.hub_row(dst[b]), part.local(src[b]), part.hub_rows, nL + nH)
    return side_a, side_b


def sides_from_routed(l1, l2, part: HubPartition, rank: int):
    """Sides A, B, A^T, B^T of rank ``rank`` from the two routed edge lists (``route_edges``):
    ``l1 = (src, dst)`` -- the non-loop edges with a HUB source whose TARGET this rank owns (sides A and B^T),
    ``l2 = (src, dst)`` -- those with a HUB target whose SOURCE this rank owns (sides A^T and B), both in edge_index order.
    Equal, entry for entry, to ``local_sides(src, dst)`` / ``local_sides(dst, src)`` on the full edge list."""
    dev = part.hub.device
    nL, nH = part.n_light(rank), part.n_hub(rank)
    rows = torch.arange(nL + nH, device=dev)
    loop_col = torch.cat([part.hub_rows + rows[:nL], rank * part.h_per + rows[:nH]])
    n_tab = part.hub_rows + nL + nH

    def pair(keys, vals):
        """(own rows <- hub table) side with its loops, and the (hub table rows <- own light rows) partial side"""
        hub_k = part.hub[keys]
        key = torch.where(hub_k, nL + part.local(keys), part.local(keys))
        full = (torch.cat([key, rows]), torch.cat([part.hub_row(vals), loop_col]), nL + nH, n_tab)
        sel = ~hub_k
        partial = (part.hub_row(vals[sel]), part.local(keys[sel]), part.hub_rows, nL + nH)
        return full, partial

    a, bt = pair(l1[1], l1[0])             # keyed by target
    at, b = pair(l2[0], l2[1])             # keyed by source
    return a, b, at, bt


def direct_sides_from_routed(l1, l2, part: HubPartition, rank: int):
    """The sides for a cut WITHOUT hub-hub edges (every bipartite graph): a hub row then gets nothing but light sources
    and its own self loop, so the loop joins the PARTIAL side of the owner -- its source row is local -- and the
    reduce-scattered sums are the complete hub rows: they land in the output directly, no merge pass.

      A  : rows = this rank's nL light rows  <- [hub table ; own rows] (the light rows' loops read the second part)
      B  : rows = hub table                  <- own rows: light sources, plus the loops of the hubs this rank owns (LAST)
    and the same two keyed by source (A^T, B^T).  Returns (a, b, at, bt) as (key, val, n_rows, n_cols)."""
    dev = part.hub.device
    nL, nH = part.n_light(rank), part.n_hub(rank)
    light = torch.arange(nL, device=dev)
    own_hub_tbl = rank * part.h_per + torch.arange(nH, device=dev)
    own_hub_loc = nL + torch.arange(nH, device=dev)
    n_tab = part.hub_rows + nL + nH

    def pair(keys, vals):
        full = (torch.cat([part.local(keys), light]), torch.cat([part.hub_row(vals), part.hub_rows + light]), nL, n_tab)
        partial = (torch.cat([part.hub_row(vals), own_hub_tbl]), torch.cat([part.local(keys), own_hub_loc]),
                   part.hub_rows, nL + nH)
        return full, partial

    a, bt = pair(l1[1], l1[0])             # keyed by (light) target
    at, b = pair(l2[0], l2[1])             # keyed by (light) source
    return a, b, at, bt


def _stable_buckets(dest: torch.Tensor, world: int, backend):
    """(order, counts): ``order`` lists the positions of ``dest`` grouped by value 0..world-1, input order kept inside a
    group (so that chunks concatenated in rank order reproduce edge_index order); ``counts[k]`` = size of group k."""
    if hasattr(backend, "bucket"):
        return backend.bucket(dest, world)
    return torch.sort(dest, stable=True)[1], torch.bincount(dest, minlength=world)


def _all_to_all_rows(rows: torch.Tensor, counts: torch.Tensor, world: int, group) -> torch.Tensor:
    """rows [n, K] grouped by destination rank (``counts[k]`` rows for rank k) -> the rows every rank addressed to this
    one, in rank order.  Set-up traffic (the routed edge list), not part of a step."""
    recv_counts = torch.empty_like(counts)
    host = _portable(group) and rows.is_cuda                      # gloo has no all-to-all on device tensors
    if host:
        cin, cout = counts.cpu(), recv_counts.cpu()
        dist.all_to_all_single(cout, cin, group=group)
        recv = cout.tolist()
        out = torch.empty((sum(recv), rows.size(1)), dtype=rows.dtype)
        dist.all_to_all_single(out, rows.cpu(), recv, cin.tolist(), group=group)
        return out.to(rows.device)
    dist.all_to_all_single(recv_counts, counts, group=group)
    recv = recv_counts.tolist()
    out = torch.empty((sum(recv), rows.size(1)), dtype=rows.dtype, device=rows.device)
    dist.all_to_all_single(out, rows.contiguous(), recv, counts.tolist(), group=group)
    return out


def route_edges(src: torch.Tensor, dst: torch.Tensor, part: HubPartition, rank: int, world: int, group=None,
                backend=None):
    """The distributed partitioner: every rank holds only ITS SLICE ``(src, dst)`` of the edge list (columns
    ``[r E / W, (r+1) E / W)`` of edge_index, already on its device) and sends each edge to the rank(s) that compute with
    it -- the owner of the target when the source is a hub, the owner of the source when the target is a hub -- in one
    all-to-all per list.  No rank ever holds the whole edge list; what it receives is its ~1/W share, in edge_index order
    (slices travel in rank order and the bucketing is stable), so the CSRs built from it are the ones the full-list
    masks (``local_sides``) give.  Also returns the global in- and out-degree (non-loop edges) of every node: one
    all-reduce of an int64 [N] vector each -- node metadata, needed for the mean and for GCN's normalisation -- and
    whether any edge joins two hubs (none on a bipartite graph: the hub rows then need no merge pass, ``direct_sides``)."""
    keep = src != dst
    src, dst = src[keep], dst[keep]
    hub_s, hub_d = part.hub[src], part.hub[dst]
    bad = (~hub_s & ~hub_d).any().to(torch.int64).view(1)
    hub_hub = (hub_s & hub_d).any().to(torch.int64).view(1)
    in_cnt = torch.bincount(dst, minlength=part.N)
    out_cnt = torch.bincount(src, minlength=part.N)
    if world > 1:
        stats = torch.cat([bad, hub_hub, in_cnt, out_cnt])
        dist.all_reduce(stats, group=group)
        bad, hub_hub, in_cnt, out_cnt = stats[:1], stats[1:2], stats[2:2 + part.N], stats[2 + part.N:]
    if int(bad) > 0:
        raise ValueError("HubPartition: an edge joins two light nodes; mark one endpoint as a hub (auto_hubs)")
    lists = []
    for mask, owner_of in ((hub_s, dst), (hub_d, src)):
        s, d = src[mask], dst[mask]
        rows = torch.stack([s, d], dim=1)
        if world > 1:
            order, counts = _stable_buckets(part.owner(owner_of[mask]), world, backend)
            rows = _all_to_all_rows(rows[order], counts, world, group)
        lists.append((rows[:, 0].contiguous(), rows[:, 1].contiguous()))
    return lists[0], lists[1], in_cnt, out_cnt, int(hub_hub) > 0


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


class _Done:
    def wait(self):
        return True


def all_gather_rows(block: torch.Tensor, out: torch.Tensor, world: int, group=None, async_op: bool = False):
    """block [h_per, F] of every rank -> out [W * h_per, F] (rank-major)"""
    if _solo(world):
        if out.data_ptr() != block.data_ptr():
            out.copy_(block)
        return None
    if _portable(group):                                           # gloo (CPU tests, one-GPU test rigs)
        work = dist.all_gather(list(out.view(world, block.size(0), -1).unbind(0)), block, group=group,
                               async_op=async_op)
    else:
        work = dist.all_gather_into_tensor(out, block, group=group, async_op=async_op)
    return work if async_op else None


def reduce_scatter_rows(part_sums: torch.Tensor, out: torch.Tensor, rank: int, world: int, group=None,
                        async_op: bool = False):
    """part_sums [W * h_per, F] of every rank -> out [h_per, F] = sum over ranks of block ``rank``"""
    if _solo(world):
        if out.data_ptr() != part_sums.data_ptr():
            out.copy_(part_sums)
        return None
    if _portable(group):
        dist.all_reduce(part_sums, group=group)
        out.copy_(part_sums.view(world, out.size(0), -1)[rank])
        return _Done() if async_op else None
    work = dist.reduce_scatter_tensor(out, part_sums, group=group, async_op=async_op)
    return work if async_op else None


def _all_reduce(t: torch.Tensor, world: int, group=None, op=None, tag: str = "all_reduce", async_op: bool = False):
    """in place; ``async_op``: returns the pending work (None for one rank) -- the caller ``_wait``s before it reads ``t``"""
    if _solo(world):
        return None
    work = dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=group, async_op=True)
    if async_op:
        return work
    _wait(work, tag, t)
    return None


def _all_reduce_params(dw: torch.Tensor, db: Optional[torch.Tensor], sg):
    """the sums of dW and db over the ranks in ONE exchange (a 1 KB all-reduce costs a full collective's latency -- 20-50 us at
    the end of a 1.5 ms step, with nothing left to hide it behind): both travel in one buffer and come back as views of it"""
    if db is None or _solo(sg.world):
        _all_reduce(dw, sg.world, sg.small_group, tag="bwd_all_reduce_dw")
        if db is not None:
            _all_reduce(db, sg.world, sg.small_group, tag="bwd_all_reduce_db")
        return dw, db
    flat = torch.cat([dw.reshape(-1), db.reshape(-1)])
    _all_reduce(flat, sg.world, sg.small_group, tag="bwd_all_reduce_params")
    return flat[: dw.numel()].view(dw.shape), flat[dw.numel():].view(db.shape)


class HipBackend:
    """Local compute of one rank on its MI355X through the C ABI."""

    def make_side(self, key, val, n_rows: int, n_cols: int):
        from .graph import build_side
        return build_side(key.contiguous(), val.contiguous(), n_rows, n_cols, False, 0, False)

    def bucket(self, dest, world: int):
        """stable grouping of positions by destination rank: the CSR build with one row per rank (one radix pass)"""
        from .graph import build_side
        n = int(dest.numel())
        if n == 0:
            return dest.new_empty(0), torch.zeros(world, dtype=torch.long, device=dest.device)
        side = build_side(dest.contiguous(), torch.arange(n, device=dest.device), world, n, False, 0, False)
        return side.col[:n].long(), (side.rowptr[1:] - side.rowptr[:-1]).long()

    def row_lengths(self, side):
        return (side.rowptr[1:] - side.rowptr[:-1])

    def row_of_entry(self, side):
        return side.rowidx[: side.nnz_max].long()

    def col_of_entry(self, side):
        return side.col[: side.nnz_max].long()

    def segsum(self, side, table, mean: bool = False, table2=None, w=None, bias=None, out=None, scales_out=None):
        from . import functional as NF
        return NF.segsum(None, side, table, w=w, mean=mean, bias=bias, x2=table2, out=out, scales_out=scales_out)

    def linear_fwd(self, a, w, b, out=None, ws=None, reserve_cus=0, a_scales=None):
        from . import functional as NF
        return NF.linear_fwd(a, w, b, out=out, ws=ws, reserve_cus=reserve_cus, a_scales=a_scales)

    def prepare_weight(self, w, backward=True, f16=False):
        """(ws_fwd, ws_bwd) for linear_fwd(..., ws=) / linear_bwd_data(..., ws=): both re-laid copies of ``w`` in one launch"""
        from . import functional as NF
        return NF.prepare_weight(w, backward, f16=f16)

    def light_scales(self, sch: Schedule, side, rows: torch.Tensor, n_light: int, weight: torch.Tensor):
        """A ``[n_light]`` buffer for the power-of-two row scales of the light rows' aggregate, when their projection takes two fp16
        pieces per operand (``Schedule.f16x2_min_rows``; 256 features: the full side's launch writes them, ``segsum(scales_out=)``),
        else None.  The light rows are nine tenths of a rank's projection; the hub rows arrive by reduce-scatter, nobody wrote
        their scales, and their GEMM stays on three bf16 pieces."""
        from . import functional as NF
        if (rows.dtype == weight.dtype and rows.size(1) == weight.size(0)
                and NF._f16x2(sch, n_light, weight.size(0), weight.size(1), rows.dtype) and NF.segsum_scales_ok(side, rows)):
            return torch.empty(n_light, dtype=torch.float32, device=rows.device)
        return None

    bwd_data_into = True                              # linear_bwd_data takes ``out=`` (a row block of a larger buffer)

    def linear_bwd_data(self, dc, w, rowscale, out=None, ws=None, reserve_cus=0):
        from . import functional as NF
        return NF.linear_bwd_data(dc, w, rowscale, out=out, ws=ws, reserve_cus=reserve_cus)

    def linear_bwd_weight(self, a, dc, want_bias, shared=False):
        from . import functional as NF
        return NF.linear_bwd_weight(a, dc, want_bias, shared=shared)

    def colsum(self, x):
        from . import functional as NF
        return NF.colsum(x)

    def side_stream(self, like, sch: Schedule = DEFAULT):
        """second HIP stream for the weight-gradient GEMM, or None when the shard is too small to gain -- or while the step is
        being CAPTURED with a real process group up: the communicator issues its collectives on a stream of its own, forked
        from whichever stream calls it; called from the side stream that is a fork of a fork, and ending such a capture takes
        the HIP runtime down (see ``partial_stream``).  The chain then stays on the launch stream: same kernels, same numbers."""
        from . import functional as NF
        if not NF._overlaps(sch, like.size(0)):
            return None
        if torch.cuda.is_current_stream_capturing() and dist.is_available() and dist.is_initialized():
            return None
        return NF._side_stream(like.device)

    def overlap_wanted(self, like, sch: Schedule = DEFAULT) -> bool:
        """would ``side_stream`` hand out a stream if no capture were on?  (the weight-gradient GEMM keeps the grid regime -- and
        with it the slab count and the order of its sums -- of the eager step, so a captured step replays the same NUMBERS)"""
        from . import functional as NF
        return NF._overlaps(sch, like.size(0))

    def partial_stream(self, like, sch: Schedule = DEFAULT):
        """third HIP stream for the partial (side B) aggregation of a direction, or None for small shards.  None as well while
        the current stream is being CAPTURED into a HIP graph: in the backward this stream is forked from the side stream, itself
        forked from the launch stream, and ending a capture with that nested fork takes the HIP runtime down (a segmentation
        fault inside hipStreamEndCapture, ROCm 7.2: EXPERIMENTS Part B, sharded capture) -- the partial side then runs in line
        on the stream that asked, which is the ``partial_stream=False`` arrangement: the same kernels and the same numbers."""
        from . import functional as NF
        if not (sch.partial_stream and NF._overlaps(sch, like.size(0))):
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        return NF._side_stream(like.device, 1)

    # ---- GATConv pieces (functional.py wraps the C ABI; index spaces: rows of the side / its table) ----
    def linear_fwd_scores(self, a, w, att2, sch: Schedule = DEFAULT):
        """(h, a_dst, a_src) with the scores from the GEMM's store epilogue (one head), or None when the shape is not served"""
        from . import functional as NF
        if sch.gat_scores_epilogue and NF.linear_fwd_scores_ok(a, w):
            return NF.linear_fwd_scores(a, w, att2)
        return None

    def gat_scores(self, h, att2, H, C):
        from . import functional as NF
        return NF.gat_scores(h, att2, H, C)

    def gat_stats(self, side, a_row, a_col, H, slope):
        from . import functional as NF
        return NF.gat_softmax_stats(side, a_row, a_col, H, slope)

    def gat_aggregate(self, side, table, table2, H, C, a_dst, a_src, m, s, slope, by_source, bias=None,
                      g_dst=None, g_src=None, att=None, out=None):
        from . import functional as NF
        if side.nnz_max == 0:                      # a side without entries: every row is empty -- zeros, plus the bias (as gat_aggregate_scores)
            res = table.new_zeros((side.n_rows, H * C)) if bias is None else bias.view(1, -1).expand(side.n_rows, -1).clone()
            return res if out is None else out.copy_(res)
        return NF._gat_aggregate(None, side, table, H, C, a_dst.contiguous(), a_src.contiguous(), m.contiguous(), s.contiguous(),
                                 slope, by_source, bias=bias, g_dst=g_dst, g_src=g_src, att=att, x2=table2, out=out)

    def gat_rowdot(self, a, b, bias, H, C):
        from . import functional as NF
        return NF.gat_rowdot(a, b, bias, H, C)

    def gat_rowdot_colsum(self, a, b, bias, H, C, want_colsum=True):
        from . import functional as NF
        return NF.gat_rowdot_colsum(a, b, bias, H, C, want_colsum=want_colsum)

    def gat_edge_grad(self, side, col_feat, col_feat2, row_feat, H, C, a_dst, a_src, m, s, D, slope, swap):
        from . import functional as NF
        return NF.gat_edge_grad(side, col_feat, col_feat2, row_feat, H, C, a_dst, a_src, m, s, D, slope, swap)

    def seg_rowsum(self, side, vals, H, map_=None):
        from . import functional as NF
        return NF.seg_rowsum(side, vals, H, map_=map_)

    # ---- one head on the direct layout (_ShardedGatDirectFn): the round-3 kernels of the single-GPU layer
    def entry_source_index(self, side):
        """for every entry of the side, its position in the (key, val) arrays the side was built from"""
        return side.eid[: side.nnz_max].long()

    def gat_stats_scores(self, side, a_row, a_col, slope):
        from . import functional as NF
        return NF.gat_softmax_stats(side, a_row.contiguous(), a_col.contiguous(), 1, slope, want_scores=True)

    def gat_aggregate_scores(self, side, table, table2, C, scores, m, s, bias=None, out=None):
        from . import functional as NF
        return NF.gat_aggregate_scores(side, table, table2, C, scores, m.contiguous(), s.contiguous(), bias=bias, out=out)

    def gat_pack(self, a_dst, m, s, D, out=None):
        from . import functional as NF
        return NF.gat_pack_targets(a_dst, m, s, D, out=out)

    fused_rowsum = True                               # gat_backward_fused(rowsum_out=): the pass also sums dz by its own rows (one head)

    def gat_backward_fused(self, side, dout, dout2, hrow, C, tpack, a_src_rows, slope, out=None, H=1, rowsum_out=None):
        from . import functional as NF
        return NF.gat_backward_fused_packed(side, dout, dout2, hrow, C, tpack, a_src_rows, slope, out=out, H=H, rowsum_out=rowsum_out)

    def gat_rank1_add(self, dh, g_dst, g_src, att2, H, C):
        from . import functional as NF
        return NF.gat_rank1_add(dh, g_dst.contiguous(), g_src.contiguous(), att2, H, C)

    def gat_att_grad(self, h, g_dst, g_src, H, C):
        from . import functional as NF
        return NF.gat_att_grad(h, g_dst, g_src, H, C)

    def linear_bwd_data_rank2_ok(self, dc, w, sch: Schedule = DEFAULT):
        from . import functional as NF
        return sch.gat_rank2_epilogue and NF.linear_bwd_data_rank2_ok(dc, w)

    def linear_bwd_data_rank2(self, dc, w, row0, row1, col0, col1):
        from . import functional as NF
        return NF.linear_bwd_data_rank2(dc, w, row0, row1, col0, col1)

    def gat_rank2_cols(self, w, att2):
        from . import functional as NF
        return NF.gat_rank2_cols(w, att2)

    def gat_rank2_tail(self, P, w, att2, dw, want_datt):
        from . import functional as NF
        return NF.gat_rank2_tail(P, w, att2, dw, want_datt)


class ShardedGraph:
    """This rank's shard of the (self-loop-augmented) graph: sides A, B and their transposes."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, rank: int, world: int, device,
                 backend=None, group=None, hub_mask: Optional[torch.Tensor] = None, sliced: bool = False,
                 schedule: Schedule = DEFAULT, small_group=None):
        """``small_group``: a SECOND communicator over the same ranks (``dist.new_group()``) for the step's small exchanges --
        per-row scalars of the hub rows, the MAX of the softmax, parameter-gradient sums.  Collectives of one communicator
        run in issue order, so on ``group`` alone a 4-byte-per-row gather waits behind the 512-byte-per-row one issued before
        it; None: everything on ``group``.  ``schedule``: how the layers on this shard arrange their launches and collectives (``schedule.Schedule``: hub-row
        layout, streams, split projection); the same on every rank.  ``sliced=True``: ``edge_index`` is THIS RANK'S slice of the edge list (the ranks' slices, in rank order, are
        the whole list); ``sliced=False``: the complete list, the same on every rank -- the rank then keeps only columns
        ``[rank E / W, (rank + 1) E / W)`` BEFORE anything moves to its GPU.  Either way the edges reach the ranks that
        compute with them through ``route_edges`` (an all-to-all): no GPU ever holds the whole edge list.  Without an
        initialised process group (W == 1; virtual ranks in one process) the complete list is required and the rank's
        share is cut out of it with masks -- the same entries in the same order (tests/test_dist_gloo.py)."""
        self.part = part = HubPartition(num_nodes, world, hub_mask, device)
        self.rank, self.world, self.group = rank, world, group
        self.small_group = small_group if small_group is not None else group
        self.schedule = schedule
        self.nL, self.nH = part.n_light(rank), part.n_hub(rank)
        self.n_local = self.nL + self.nH
        self.exchange_partials = part.nL > 0                      # same answer on every rank
        self.backend = be = backend or HipBackend()
        routed = world > 1 and dist.is_available() and dist.is_initialized()
        if routed:
            if not sliced:
                E = edge_index.size(1)
                edge_index = edge_index[:, rank * E // world: (rank + 1) * E // world]
            ei = edge_index.to(device)
            l1, l2, in_cnt, out_cnt, hub_hub = route_edges(ei[0], ei[1], part, rank, world, group, be)
            del ei
        else:
            if sliced and world > 1:
                raise ValueError("ShardedGraph(sliced=True) needs an initialised process group to route the edges")
            ei = edge_index.to(device)
            keep = ei[0] != ei[1]
            src, dst = ei[0][keep], ei[1][keep]
            hub_s, hub_d = part.hub[src], part.hub[dst]
            if bool((~hub_s & ~hub_d).any()):
                raise ValueError("HubPartition: an edge joins two light nodes; mark one endpoint as a hub (auto_hubs)")
            m1, m2 = hub_s & (part.owner(dst) == rank), hub_d & (part.owner(src) == rank)
            l1, l2 = (src[m1], dst[m1]), (src[m2], dst[m2])
            in_cnt, out_cnt = torch.bincount(dst, minlength=num_nodes), torch.bincount(src, minlength=num_nodes)
            hub_hub = bool((hub_s & hub_d).any())
            del ei, src, dst, hub_s, hub_d, m1, m2
        self._l1, self._l2 = l1, l2                               # this rank's routed share of the edge list (~2 E / W edges)
        # no hub-hub edge anywhere (bipartite graphs): SAGE / GCN use the direct layout (direct_sides_from_routed)
        self.direct_ok = self.exchange_partials and not hub_hub and schedule.direct_hub_rows
        self._classic = self._direct = None
        n_b = int((~part.hub[l2[0]]).sum()) if self.exchange_partials else 0
        self.local_nnz = int(l1[0].numel()) + self.n_local + n_b  # entries this rank walks per direction
        # 1 / (in-degree + 1) of the local rows; for hub rows also the share of it that side A holds
        own = part.own_ids(rank)
        self._in_cnt = in_cnt
        self.inv_cnt = (1.0 / (in_cnt[own].to(torch.float32) + 1.0)).contiguous()
        hub_keys = l1[1][part.hub[l1[1]]]
        self.cnt_a_hub = (torch.bincount(part.local(hub_keys), minlength=self.nH)[: self.nH].to(torch.float32) + 1.0).view(-1, 1)
        # the hub rows' merge after the reduce-scatter, agg = agg (cnt_A / cnt) + hsum / cnt, as two per-row factors
        self.hub_scale_a = (self.cnt_a_hub * self.inv_cnt[self.nL:].view(-1, 1)).contiguous()
        self.hub_scale_b = self.inv_cnt[self.nL:].view(-1, 1).contiguous()
        self._out_deg = out_cnt                                     # global, [N] (GCNConv's degree is over the SOURCE rows)
        self.own = own
        self.own_hub = slice(rank * part.h_per, rank * part.h_per + self.nH)       # this rank's rows of the hub table
        self._gcn = None
        self._gcn_direct = None
        self._gat_maps = None
        self._b_empty = None

    # ---- classic layout: A holds every local row with its loop, B the partial hub sums from light sources (GATConv, and
    #      SAGE / GCN on cuts with hub-hub edges); built on first use
    def _classic_sides(self):
        if self._classic is None:
            be = self.backend
            a, b, at, bt = sides_from_routed(self._l1, self._l2, self.part, self.rank)
            x = self.exchange_partials
            self._classic = (be.make_side(*a), be.make_side(*b) if x else None, be.make_side(*at),
                             be.make_side(*bt) if x else None)
        return self._classic

    A = property(lambda self: self._classic_sides()[0])
    B = property(lambda self: self._classic_sides()[1])
    At = property(lambda self: self._classic_sides()[2])
    Bt = property(lambda self: self._classic_sides()[3])

    def direct(self):
        """(A, B, At, Bt) of the direct layout (``direct_sides_from_routed``); only when ``direct_ok``"""
        if self._direct is None:
            be = self.backend
            self._direct = tuple(be.make_side(*t) for t in direct_sides_from_routed(self._l1, self._l2, self.part, self.rank))
        return self._direct

    def direct_weights(self, gcn: bool):
        """Per-entry weights of the direct layout's four sides.  SAGE: the partial sides carry ``1 / (in-degree + 1)`` of
        their (hub) row -- global counts, so the partial sums of all ranks ADD to the mean; the full sides use the
        kernel's own mean (a light row is complete on its owner).  GCN: the symmetric normalisation on every side."""
        if self._gcn_direct is None:
            self._gcn_direct = {}
        if gcn not in self._gcn_direct:
            part, be = self.part, self.backend
            A, B, At, Bt = self.direct()
            tid = part.hub_table_ids()
            if gcn:
                dinv = (self._out_deg.to(torch.float32) + 1.0).pow(-0.5)
                d_hub = torch.where(tid >= 0, dinv[tid.clamp(min=0)], torch.zeros((), device=dinv.device))
                d_own = dinv[self.own]
                d_tbl = torch.cat([d_hub, d_own])
                w = {n: (d_own[be.row_of_entry(sd)] * d_tbl[be.col_of_entry(sd)]).contiguous() for n, sd in (("A", A), ("At", At))}
                w.update({n: (d_hub[be.row_of_entry(sd)] * d_own[be.col_of_entry(sd)]).contiguous() for n, sd in (("B", B), ("Bt", Bt))})
            else:
                inv_hub = torch.where(tid >= 0, 1.0 / (self._in_cnt[tid.clamp(min=0)].to(torch.float32) + 1.0),
                                      torch.zeros((), device=tid.device))
                # backward: dagg arrives pre-divided by the TARGET's count (the GEMM epilogue), so B^T adds plain sums
                w = {"A": None, "At": None, "B": inv_hub[be.row_of_entry(B)].contiguous(), "Bt": None}
            self._gcn_direct[gcn] = w
        return self._gcn_direct[gcn]

    def gat_maps(self):
        """Direct layout, GATConv backward: dz of every entry is computed ONCE, on the by-source sides (A^T: own light sources,
        B^T: hub sources); the by-target row sums need it in the order of A / B.  A and B^T hold the same routed edges (list 1),
        B and A^T likewise (list 2), the light rows' loops live in A and A^T, the own hubs' loops in B and B^T -- so the two maps
        ``entry of A / of B -> position in cat[dz of B^T, dz of A^T]`` are local index plumbing, built once."""
        if self._gat_maps is None:
            be = self.backend
            A, B, At, Bt = self.direct()
            n1, n2 = int(self._l1[0].numel()), int(self._l2[0].numel())
            dev = self.part.hub.device
            eA, eB, eAt, eBt = (be.entry_source_index(sd) for sd in (A, B, At, Bt))
            nBt, nAt = int(eBt.numel()), int(eAt.numel())
            pos_bt = torch.zeros(n1 + self.nH + 1, dtype=torch.long, device=dev)
            pos_bt[eBt] = torch.arange(nBt, device=dev)              # list-1 edge e / own hub loop n1 + k -> entry of B^T
            pos_at = torch.zeros(n2 + self.nL + 1, dtype=torch.long, device=dev)
            pos_at[eAt] = torch.arange(nAt, device=dev)              # list-2 edge e / light loop n2 + i -> entry of A^T
            map_a = torch.where(eA < n1, pos_bt[eA.clamp(max=max(n1 - 1, 0))], nBt + pos_at[(n2 + eA - n1).clamp(min=0)])
            map_b = torch.where(eB < n2, nBt + pos_at[eB.clamp(max=max(n2 - 1, 0))], pos_bt[(n1 + eB - n2).clamp(min=0)])
            self._gat_maps = (map_a.to(torch.int32).contiguous(), map_b.to(torch.int32).contiguous(), nBt, nAt)
        return self._gat_maps

    def shard(self, x_full: torch.Tensor) -> torch.Tensor:
        return x_full[self.own.to(x_full.device)]

    def gcn_norm(self):
        """Per-entry symmetric normalisation (PyG 1.4.2 ``GCNConv.norm``: deg = out-degree incl. the self loop, over
        the SOURCE rows) of sides A, B, At, Bt -- the degrees are global, every rank derives them from the edge list."""
        if self._gcn is None:
            part, be, N = self.part, self.backend, self.part.N
            deg = self._out_deg.to(torch.float32) + 1.0
            dinv = deg.pow(-0.5)
            tid = part.hub_table_ids()
            d_hub = torch.where(tid >= 0, dinv[tid.clamp(min=0)], torch.zeros((), device=dinv.device))
            d_own = dinv[self.own]
            d_tbl = torch.cat([d_hub, d_own])                     # index space of side A's columns
            w = {}
            for name, side in (("A", self.A), ("At", self.At)):
                w[name] = (d_own[be.row_of_entry(side)] * d_tbl[be.col_of_entry(side)]).contiguous()
            for name, side in (("B", self.B), ("Bt", self.Bt)):
                w[name] = None if side is None else (d_hub[be.row_of_entry(side)] * d_own[be.col_of_entry(side)]).contiguous()
            self._gcn = w
        return self._gcn

    def b_empty(self):
        """[hub_rows, 1] bool: hub-table rows to which this rank contributes no entry (side B)"""
        if self._b_empty is None and self.B is not None:
            self._b_empty = (self.backend.row_lengths(self.B) == 0).view(-1, 1)
        return self._b_empty


def _hub_block(sg: ShardedGraph, rows: torch.Tensor) -> torch.Tensor:
    """this rank's hub rows of ``rows`` [n_local, K], zero-padded to h_per"""
    block = rows[sg.nL:]
    if sg.nH != sg.part.h_per:                                     # short rank: one zero pad row
        block = rows.new_zeros((sg.part.h_per,) + tuple(rows.shape[1:]))
        block[: sg.nH] = rows[sg.nL:]
    return block.contiguous()


def gather_hub(sg: ShardedGraph, rows: torch.Tensor, async_op: bool = False, small: bool = False):
    """all-gather of the hub rows of ``rows`` [n_local, K] -> ([hub_rows, K] rank-major, pending work or None); ``small``: on the
    communicator of the small exchanges (``ShardedGraph.small_group``)"""
    part, W = sg.part, sg.world
    block = _hub_block(sg, rows)
    if _solo(W):
        return block, None                                         # one rank: the hub table IS its own hub block
    table = rows.new_empty((part.hub_rows,) + tuple(rows.shape[1:]))
    work = all_gather_rows(block, table, W, sg.small_group if small else sg.group, async_op=async_op)
    return table, work


def scatter_hub_sums(sg: ShardedGraph, partial: torch.Tensor, async_op: bool = False):
    """reduce-scatter of partial hub sums [hub_rows, K] -> ([h_per, K] for this rank's hubs, pending work or None)"""
    W = sg.world
    if _solo(W):
        return partial, None
    hsum = partial.new_empty((sg.part.h_per,) + tuple(partial.shape[1:]))
    work = reduce_scatter_rows(partial, hsum, sg.rank, W, sg.group, async_op=async_op)
    return hsum, work


# Side B (partial hub sums from the rank's own rows) needs nothing from another rank, side A needs the all-gathered hub
# table: with B on its own HIP stream (Schedule.partial_stream) the two aggregations of a direction share the CUs (W > 1: A
# starts the moment the table has arrived instead of behind B); otherwise one stream, B first.  Schedule.split_projection
# (direct layout, W > 1): the projection of the light rows (complete on the rank) is launched before the reduce-scattered hub
# rows have arrived, the hub rows' projection after them; otherwise one GEMM over all rows behind the reduce-scatter.


def _hub_aggregate(sg: ShardedGraph, rows: torch.Tensor, full, partial, w_full, w_part, mean: bool, tag: str, bias=None,
                   direct: bool = False, defer: bool = False, gathered=None, light_scales=None):
    """One direction of the hub-cut aggregation of ``rows`` [n_local, F]:

        table = all_gather(hub rows of ``rows``)                      | psum = segsum(partial side, rows)
        out   = segsum(full side, [table ; rows])  (mean / bias fused) | hsum = reduce_scatter(psum)

    Classic layout: returns (out [n_local, F], hsum or None, table) and the caller folds ``hsum[:nH]`` into ``out[nL:]``
    (the rule differs: SAGE's mean re-weights the two shares, sums just add).  ``direct`` (sides of
    ``ShardedGraph.direct()``): the full side covers the light rows only and the reduce-scatter delivers the COMPLETE hub
    rows, so both write into one ``[nL + h_per, F]`` buffer and (out[:n_local], None, table) comes back -- nothing to fold.
    ``defer`` (direct only): the light rows are returned BEFORE the hub rows have arrived -- the second item is then a
    callable that waits for the reduce-scatter (and the partial stream); the caller may work on ``out[:nL]`` first.
    ``gathered``: ``gather_hub(sg, rows, async_op=True)`` already issued by the caller (the backward starts it as soon as
    the hub rows of ``rows`` exist).  ``tag``: prefix of the exposed-communication records ("fwd" / "bwd")."""
    be, W = sg.backend, sg.world
    # needs nothing but the hub rows of ``rows``: issued first
    table, g_work = gathered if gathered is not None else gather_hub(sg, rows, async_op=True)
    hsum = b_stream = r_work = out_full = None
    if direct:
        out_full = rows.new_empty((sg.nL + sg.part.h_per, rows.size(1)))
    if sg.exchange_partials:
        # (one rank: nothing to overlap with -- the two launches of a direction share a stream, 7.30 vs 7.33 ms at C4)
        b_stream = be.partial_stream(rows, sg.schedule) if (not _solo(W) and hasattr(be, "partial_stream")) else None
        cur = None

        def partial_side():
            if direct and _solo(W):                                # one rank: the hub table IS its block of the output
                return be.segsum(partial, rows, w=w_part, out=out_full[sg.nL:]), None
            psum = be.segsum(partial, rows, w=w_part)              # no remote input: overlaps the all-gather
            if direct:
                hub_out = out_full[sg.nL:]
                return hub_out, reduce_scatter_rows(psum, hub_out, sg.rank, W, sg.group, async_op=True)
            return scatter_hub_sums(sg, psum, async_op=True)
        if b_stream is None:
            hsum, r_work = partial_side()
        else:
            cur = torch.cuda.current_stream(rows.device)
            b_stream.wait_stream(cur)                              # ``rows`` is complete for the partial side
            with torch.cuda.stream(b_stream):
                hsum, r_work = partial_side()
            for t in (rows, w_part, out_full):
                if t is not None:
                    t.record_stream(b_stream)                      # allocated on ``cur``, used on the partial stream
    _wait(g_work, tag + "_all_gather", table)
    kw_sc = {"scales_out": light_scales} if light_scales is not None else {}      # (direct layout: the full side IS the light rows)
    out = be.segsum(full, table, mean=mean, table2=rows, w=w_full, bias=bias, out=out_full[: sg.nL] if direct else None, **kw_sc)

    def hub_rows_arrived():
        if hsum is not None:
            _wait(r_work, tag + "_reduce_scatter", hsum)
            if b_stream is not None:
                cur.wait_stream(b_stream)                          # one rank: no collective to wait for
                hsum.record_stream(cur)
    if direct and defer:
        return out_full[: sg.n_local], hub_rows_arrived, table
    hub_rows_arrived()
    if direct:
        return out_full[: sg.n_local], None, table
    return out, hsum, table


class _ShardedSageFn(torch.autograd.Function):
    """Aggregate, then project.  ``gcn = False``: SAGEConv (mean over in-neighbours and self).  ``gcn = True``: GCNConv in
    the same order, ``(A_hat x) W + b`` with the symmetric normalisation as per-entry weights (see
    functional._GcnAggFirstFn: identical to PyG's ``A_hat (x W) + b`` up to rounding, dW overlaps the backward aggregation)."""

    @staticmethod
    def forward(ctx, x_own, weight, bias, sg: ShardedGraph, gcn: bool = False):
        be = sg.backend
        x_own = x_own.contiguous()
        nrm = {"A": None, "B": None} if (sg.direct_ok or not gcn) else sg.gcn_norm()
        if sg.direct_ok:
            d, w = sg.direct(), sg.direct_weights(gcn)
            split = sg.schedule.split_projection and sg.nL > 0 and sg.nH > 0 and not _solo(sg.world)
            # the light rows' projection on two fp16 pieces per operand: their scales from the aggregation launch that writes them
            lsc = be.light_scales(sg.schedule, d[0], x_own, sg.nL, weight) if (split and hasattr(be, "light_scales")) else None
            agg, arrived, _ = _hub_aggregate(sg, x_own, d[0], d[1], w["A"], w["B"], not gcn, "fwd", direct=True, defer=split,
                                             light_scales=lsc)
            # both re-laid copies of W (this direction's GEMMs and the backward's) in one launch
            wsf, ws_bwd = be.prepare_weight(weight, ctx.needs_input_grad[0]) if hasattr(be, "prepare_weight") else (None, None)
            kw = {"ws": wsf} if wsf is not None else {}
            if hasattr(be, "prepare_weight") and not _solo(sg.world) and sg.schedule.gemm_reserve_cus:
                kw["reserve_cus"] = sg.schedule.gemm_reserve_cus           # a collective may be resident beside these GEMMs
            if split:
                # the light rows are complete on this rank: their projection runs while the hub rows are still on the wire
                # (row-wise independent: the same numbers as one GEMM over all rows)
                out = agg.new_empty((sg.n_local, weight.size(1)))
                kw_light = dict(kw)
                if hasattr(be, "prepare_weight"):                  # this GEMM shares the chip with the reduce-scatter's kernel
                    kw_light["reserve_cus"] = max(kw.get("reserve_cus", 0), sg.schedule.split_projection_reserve_cus)
                if lsc is not None:                                # (its own fp16 x 2 copy of W: one more small launch)
                    kw_light.update(ws=be.prepare_weight(weight, False, f16=True)[0], a_scales=lsc)
                be.linear_fwd(agg[: sg.nL], weight, bias, out=out[: sg.nL], **kw_light)
                arrived()
                be.linear_fwd(agg[sg.nL:], weight, bias, out=out[sg.nL:], **kw)
            else:
                out = be.linear_fwd(agg, weight, bias, **kw)
        else:
            agg, hsum, _ = _hub_aggregate(sg, x_own, sg.A, sg.B, nrm["A"], nrm["B"], not gcn, "fwd")
            if hsum is not None and sg.nH:
                if gcn:
                    agg[sg.nL:] += hsum[: sg.nH]
                else:                                              # mean over both shares: (agg cnt_A + hsum) / cnt
                    agg[sg.nL:].mul_(sg.hub_scale_a).addcmul_(hsum[: sg.nH], sg.hub_scale_b)
            out = be.linear_fwd(agg, weight, bias)
            ws_bwd = None
        ctx.ws_bwd = ws_bwd
        ctx.sg = sg
        ctx.gcn = gcn
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
        dagg = None
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        want_x = ctx.needs_input_grad[0]
        nrm = {"At": None, "Bt": None} if (sg.direct_ok or not ctx.gcn) else sg.gcn_norm()
        gathered = None
        if want_x:
            rs = None if ctx.gcn else sg.inv_cnt
            nL = sg.nL
            if (sg.direct_ok and sg.schedule.split_projection and sg.schedule.early_hub_gather and nL > 0 and sg.nH > 0
                    and not _solo(sg.world) and getattr(be, "bwd_data_into", False)):
                # the hub rows of dAgg first (a one-round GEMM): their all-gather is on the wire while the light rows -- nine
                # tenths of the GEMM -- are computed (row-wise independent: the same numbers as one GEMM over all rows)
                kw = {"ws": ctx.ws_bwd} if ctx.ws_bwd is not None else {}
                if sg.schedule.gemm_reserve_cus:
                    kw["reserve_cus"] = sg.schedule.gemm_reserve_cus
                dagg = grad_out.new_empty((sg.n_local, weight.size(0)))
                be.linear_bwd_data(grad_out[nL:], weight, None if rs is None else rs[nL:], out=dagg[nL:], **kw)
                gathered = gather_hub(sg, dagg, async_op=True)
                be.linear_bwd_data(grad_out[:nL], weight, None if rs is None else rs[:nL], out=dagg[:nL], **kw)
            elif ctx.ws_bwd is not None:
                dagg = be.linear_bwd_data(grad_out, weight, rs, ws=ctx.ws_bwd,
                                          reserve_cus=0 if _solo(sg.world) else sg.schedule.gemm_reserve_cus)
            else:
                dagg = be.linear_bwd_data(grad_out, weight, rs)
        # dW is independent of the dX chain.  On the GPU backend it is launched FIRST, on this stream, so that it is resident
        # before the aggregations -- BOTH sides, sent to a second stream -- fill the CUs (see functional._SageConvFn); with
        # side B in front of dW, as in round 1, half of the aggregation ran alone and dW then outlasted the other half
        # (W = 1: 7.99 ms per step).  Other backends run everything in line.
        side = be.side_stream(grad_out, sg.schedule) if (want_w and want_x and hasattr(be, "side_stream")) else None
        main = torch.cuda.current_stream(grad_out.device) if side is not None else None
        if side is not None:
            side.wait_stream(main)
        if want_w:
            shared = side is not None or (want_x and hasattr(be, "overlap_wanted") and be.overlap_wanted(grad_out, sg.schedule))
            dw, db = be.linear_bwd_weight(agg, grad_out, ctx.has_bias, shared=True) if shared else \
                be.linear_bwd_weight(agg, grad_out, ctx.has_bias)
        if want_x:
            def chain():
                if sg.direct_ok:
                    d, w = sg.direct(), sg.direct_weights(ctx.gcn)
                    out, hsum, table = _hub_aggregate(sg, dagg, d[2], d[3], w["At"], w["Bt"], False, "bwd", direct=True,
                                                      gathered=gathered)
                else:
                    out, hsum, table = _hub_aggregate(sg, dagg, sg.At, sg.Bt, nrm["At"], nrm["Bt"], False, "bwd")
                if hsum is not None and sg.nH:
                    out[sg.nL:] += hsum[: sg.nH]
                return out, (table, hsum)
            if side is not None:
                with torch.cuda.stream(side):
                    dx, keep = chain()
                for t in keep + (dagg,):
                    if t is not None:
                        t.record_stream(side)
                dx.record_stream(main)
                main.wait_stream(side)
            else:
                dx, _ = chain()
        if want_w:
            dw, db = _all_reduce_params(dw, db, sg)
        return dx, dw, db, None, None


class _ShardedGcnFn(torch.autograd.Function):
    """GCNConv (PyG 1.4.2, normalize=True, unweighted): project first, then the hub exchange at width F_out with
    ``norm_e = deg^-1/2[src] deg^-1/2[dst]`` as per-entry weights of every side."""

    @staticmethod
    def forward(ctx, x_own, weight, bias, sg: ShardedGraph):
        be = sg.backend
        nrm = sg.gcn_norm()
        x_own = x_own.contiguous()
        xw = be.linear_fwd(x_own, weight, None)
        out, hsum, _ = _hub_aggregate(sg, xw, sg.A, sg.B, nrm["A"], nrm["B"], False, "fwd", bias=bias)
        if hsum is not None and sg.nH:
            out[sg.nL:] += hsum[: sg.nH]
        ctx.sg = sg
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x_own, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x_own, weight = ctx.saved_tensors
        sg: ShardedGraph = ctx.sg
        be = sg.backend
        nrm = sg.gcn_norm()
        grad_out = grad_out.contiguous()
        dx = dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = be.colsum(grad_out)
            _all_reduce(db, sg.world, sg.small_group, tag="bwd_all_reduce_db")
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dxw, hsum, _ = _hub_aggregate(sg, grad_out, sg.At, sg.Bt, nrm["At"], nrm["Bt"], False, "bwd")
            if hsum is not None and sg.nH:
                dxw[sg.nL:] += hsum[: sg.nH]
            if ctx.needs_input_grad[1]:
                dw, _ = be.linear_bwd_weight(x_own, dxw, False)
                _all_reduce(dw, sg.world, sg.small_group, tag="bwd_all_reduce_dw")
            if ctx.needs_input_grad[0]:
                dx = be.linear_bwd_data(dxw, weight, None)
        return dx, dw, db, None


class _ShardedGatFn(torch.autograd.Function):
    """GATConv (PyG 1.4.2, dropout 0, concat) on the hub cut.  The softmax of a LIGHT target row is local (all its
    sources are hub rows of the gathered table, plus itself).  A HUB target row has sources on every rank:

      forward   rank r: (m_r, s_r) = (max, sum exp(. - m_r)) of ITS entries of the row      [side B; owner: side A]
                M = all_reduce(MAX) of m_r                       one small collective, [hub_rows, H]
                U_r = sum_p exp(e_p - M) h_j,  S_r = s_r exp(m_r - M)                       relative to the GLOBAL max
                owner: out = (sum_r U_r) / (sum_r S_r + 1e-16)   reduce-scatter of U [hub_rows, F] and S [hub_rows, H]
      backward  every rank needs (a_dst, M, S, D, dOut) of the hub TARGETS it holds entries of: all-gather of the hub
                rows of dOut and of three scalars; dz is recomputed in the orientation it is summed in (by target for
                g_dst, by source for g_src; npi_gat_edge_grad_ex swap), partial hub sums reduce-scattered.
    """

    @staticmethod
    def forward(ctx, x_own, weight, att, bias, sg: ShardedGraph, heads: int, slope: float):
        be, part, W = sg.backend, sg.part, sg.world
        nL, nH = sg.nL, sg.nH
        H = int(heads)
        C = weight.size(1) // H
        x_own = x_own.contiguous()
        att2 = att.reshape(H, 2 * C).contiguous()
        h = be.linear_fwd(x_own, weight, None)
        a_dst, a_src = be.gat_scores(h, att2, H, C)                              # [n_local, H] each
        tbl_h, g_work = gather_hub(sg, h, async_op=True)                          # big: hub rows of h
        hub_sc, _ = gather_hub(sg, torch.cat([a_dst, a_src], dim=1), small=True)              # small
        hub_a_dst, hub_a_src = hub_sc[:, :H].contiguous(), hub_sc[:, H:].contiguous()
        tbl_a_src = torch.cat([hub_a_src, a_src])                                 # index space of side A's columns
        mA, sA = be.gat_stats(sg.A, a_dst, tbl_a_src, H, slope)                   # every row holds its self loop
        own = sg.own_hub
        if sg.exchange_partials:
            mB, sB = be.gat_stats(sg.B, hub_a_dst, a_src, H, slope)
            empty = sg.b_empty()
            mB = torch.where(empty, torch.full_like(mB, NEG), mB)
            M = mB.clone()
        else:
            M = torch.full((part.hub_rows, H), NEG, dtype=h.dtype, device=h.device)
        M[own] = torch.maximum(M[own], mA[nL:])
        _all_reduce(M, W, sg.small_group, op=dist.ReduceOp.MAX, tag="fwd_all_reduce_max")
        ones_h = torch.ones((nH, H), dtype=h.dtype, device=h.device)
        m_own = torch.cat([mA[:nL], M[own]])
        _wait(g_work, "fwd_all_gather", tbl_h)
        # light rows: finished softmax; hub rows: weighted sum relative to M, not yet normalised
        out = be.gat_aggregate(sg.A, tbl_h, h, H, C, a_dst, tbl_a_src, m_own, torch.cat([sA[:nL], ones_h]), slope,
                               False, bias=bias)
        S_tot = sA[nL:] * torch.exp(mA[nL:] - M[own])
        U = out[nL:] - bias if bias is not None else out[nL:]
        if sg.exchange_partials:
            U_B = be.gat_aggregate(sg.B, h, None, H, C, hub_a_dst, a_src, M, torch.ones_like(M), slope, False)
            S_B = torch.where(empty, torch.zeros_like(sB), sB * torch.exp(mB - M))
            hU, wU = scatter_hub_sums(sg, U_B, async_op=True)
            hS, wS = scatter_hub_sums(sg, S_B.contiguous(), async_op=True)
            _wait(wU, "fwd_reduce_scatter", hU)
            _wait(wS, "fwd_reduce_scatter_s", hS)
            U = U + hU[:nH]
            S_tot = S_tot + hS[:nH]
        if nH:
            res = (U.view(nH, H, C) / (S_tot.view(nH, H, 1) + 1e-16)).reshape(nH, H * C)
            out[nL:] = res + bias if bias is not None else res
        s_own = torch.cat([sA[:nL], S_tot])
        hubS, _ = gather_hub(sg, s_own, small=True)                                           # the backward's per-target sums
        ctx.sg, ctx.H, ctx.C, ctx.slope = sg, H, C, float(slope)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x_own, weight, att2, h, a_dst, a_src, tbl_h, hub_a_dst, hub_a_src, M, hubS, m_own, s_own,
                              out, bias if bias is not None else h.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (x_own, weight, att2, h, a_dst, a_src, tbl_h, hub_a_dst, hub_a_src, hubM, hubS, m_own, s_own, out,
         bias) = ctx.saved_tensors
        sg: ShardedGraph = ctx.sg
        be, W = sg.backend, sg.world
        nL, nH = sg.nL, sg.nH
        H, C, slope = ctx.H, ctx.C, ctx.slope
        dO = grad_out.contiguous()
        db = None
        if ctx.has_bias and ctx.needs_input_grad[3]:
            db = be.colsum(dO)
            _all_reduce(db, W, sg.small_group, tag="bwd_all_reduce_db")
        D = be.gat_rowdot(dO, out, bias if ctx.has_bias else None, H, C)          # [n_local, H]
        tbl_dO, g_work = gather_hub(sg, dO, async_op=True)                        # big: hub rows of dOut
        hubD, _ = gather_hub(sg, D, small=True)
        # per-TARGET scalars in the index space of the two-part table [hub table ; own rows]
        tbl_a_dst = torch.cat([hub_a_dst, a_dst])
        tbl_m, tbl_s, tbl_D = torch.cat([hubM, m_own]), torch.cat([hubS, s_own]), torch.cat([hubD, D])
        tbl_a_src = torch.cat([hub_a_src, a_src])
        # g_dst[i] = sum over the entries whose TARGET is i
        dz = be.gat_edge_grad(sg.A, tbl_h, h, dO, H, C, a_dst, tbl_a_src, m_own, s_own, D, slope, 0)
        g_dst = be.seg_rowsum(sg.A, dz, H)
        _wait(g_work, "bwd_all_gather", tbl_dO)
        # g_src[j] = sum over the entries whose SOURCE is j: the same dz, recomputed on the by-source sides
        dz = be.gat_edge_grad(sg.At, tbl_dO, dO, h, H, C, tbl_a_dst, a_src, tbl_m, tbl_s, tbl_D, slope, 1)
        g_src = be.seg_rowsum(sg.At, dz, H)
        if sg.exchange_partials:
            dz = be.gat_edge_grad(sg.B, h, None, tbl_dO, H, C, hub_a_dst, a_src, hubM, hubS, hubD, slope, 0)
            pg_dst = be.seg_rowsum(sg.B, dz, H)
            dz = be.gat_edge_grad(sg.Bt, dO, None, tbl_h, H, C, a_dst, hub_a_src, m_own, s_own, D, slope, 1)
            pg_src = be.seg_rowsum(sg.Bt, dz, H)
            pg, wg = scatter_hub_sums(sg, torch.cat([pg_dst, pg_src], dim=1).contiguous(), async_op=True)
            # partial d h of the hub rows from this rank's light targets, under the small reduce-scatter
            pdh = be.gat_aggregate(sg.Bt, dO, None, H, C, a_dst, hub_a_src, m_own, s_own, slope, True)
            hdh, wh = scatter_hub_sums(sg, pdh, async_op=True)
            _wait(wg, "bwd_reduce_scatter_g", pg)
            if nH:
                g_dst[nL:] += pg[:nH, :H]
                g_src[nL:] += pg[:nH, H:]
        # d h_j = sum_i alpha_ij dOut_i + g_dst[j] att[:C] + g_src[j] att[C:]
        dh = be.gat_aggregate(sg.At, tbl_dO, dO, H, C, tbl_a_dst, a_src, tbl_m, tbl_s, slope, True,
                              g_dst=g_dst, g_src=g_src, att=att2)
        if sg.exchange_partials:
            _wait(wh, "bwd_reduce_scatter", hdh)
            if nH:
                dh[nL:] += hdh[:nH]
        datt = dw = dx = None
        if ctx.needs_input_grad[2]:
            datt = be.gat_att_grad(h, g_dst, g_src, H, C)
            _all_reduce(datt, W, sg.small_group, tag="bwd_all_reduce_datt")
            datt = datt.view(1, H, 2 * C)
        if ctx.needs_input_grad[1]:
            dw, _ = be.linear_bwd_weight(x_own, dh, False)
            _all_reduce(dw, W, sg.small_group, tag="bwd_all_reduce_dw")
        if ctx.needs_input_grad[0]:
            dx = be.linear_bwd_data(dh, weight, None)
        return dx, dw, datt, db, None, None, None


class _fork:
    """``with _fork(stream, inputs) as f: ...`` runs the block on ``stream`` (inline when it is None): the stream first waits
    for the current one, ``inputs`` (allocated on the current stream) are recorded on it; ``f.join(*outputs)`` makes the
    current stream wait for the block and records the outputs (allocated on ``stream``) on the current stream."""

    def __init__(self, stream, inputs=()):
        self.s, self.inputs = stream, [t for t in inputs if t is not None and t.is_cuda]
        self.ctx = None

    def __enter__(self):
        if self.s is not None:
            self.cur = torch.cuda.current_stream(self.s.device)
            self.s.wait_stream(self.cur)
            self.ctx = torch.cuda.stream(self.s)
            self.ctx.__enter__()
            for t in self.inputs:
                t.record_stream(self.s)
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *outputs):
        if self.s is not None:
            self.cur.wait_stream(self.s)
            for t in outputs:
                if t is not None and t.is_cuda:
                    t.record_stream(self.cur)


class _ShardedGatDirectFn(torch.autograd.Function):
    """GATConv, 1 / 2 / 4 / 8 heads, on the direct layout (cuts without hub-hub edges) with the single-GPU layer's round-3 kernels: the
    statistics pass leaves the per-entry scores, the aggregation reads them back, and the backward is ONE fused gather pass per
    by-source side on packed per-target scalars -- every entry's SDDMM dot is computed once (the classic path computes it on
    both orientations of both sides).  Per rank and direction: two aggregation-sized launches over its ~E/W entries.

      forward   light rows : complete softmax on the owner (sources: the gathered hub table + the own self loop)
                hub rows   : (m_r, s_r) and U_r = sum exp(e - M) h_j over the rank's light sources and, on the owner, the
                             hub's own loop; M = all_reduce(MAX) of m_r; (U, S) reduce-scattered; out = U / S + b
      backward  dz of an entry lives on the by-source side that holds it (A^T: own light sources, B^T: hub sources);
                by-target row sums read it through ``ShardedGraph.gat_maps``; partial d h / g_src / g_dst of the hub rows are
                reduce-scattered; everything else is the single-GPU layer's arithmetic."""

    @staticmethod
    def forward(ctx, x_own, weight, att, bias, sg: ShardedGraph, heads: int, slope: float):
        be, part, W = sg.backend, sg.part, sg.world
        nL, nH, hp = sg.nL, sg.nH, sg.part.h_per
        H = int(heads)
        C = weight.size(1) // H
        F = H * C
        x_own = x_own.contiguous()
        att2 = att.reshape(H, 2 * C).contiguous()
        A, B, _, _ = sg.direct()
        hs = be.linear_fwd_scores(x_own, weight, att2, sg.schedule) if (H == 1 and hasattr(be, "linear_fwd_scores")) else None
        if hs is not None:
            h, a_dst, a_src = hs                                                   # both scores in the GEMM's store epilogue
        else:
            h = be.linear_fwd(x_own, weight, None)
            a_dst, a_src = be.gat_scores(h, att2, H, C)                            # [n_local, H] each
        # the small exchange FIRST: collectives of one communicator run in issue order, and behind the 0.4 GB table of hub rows
        # this one -- which the partial chain below starts from -- would wait for all of it
        hub_sc, _ = gather_hub(sg, torch.cat([a_dst, a_src], dim=1), small=True)               # small
        tbl_h, g_work = gather_hub(sg, h, async_op=True)                           # big: hub rows of h
        tbl_a_dst, tbl_a_src = hub_sc[:, :H].contiguous(), hub_sc[:, H:].contiguous()
        # hub rows: this rank's share -- its light sources and the loops of the hubs it owns; needs nothing of tbl_h, so the
        # whole chain (statistics, MAX all-reduce, aggregation, reduce-scatters) runs on the partial stream beside the light rows
        # (one head: the statistics pass leaves the per-entry scores and the aggregation reads them back)
        out_full = h.new_empty((nL + hp, F))
        hU, wU = out_full[nL:], None                                               # the reduce-scatter lands in the output
        bs = be.partial_stream(h, sg.schedule) if (not _solo(W) and hasattr(be, "partial_stream")) else None
        with _fork(bs, (h, a_src, tbl_a_dst, out_full)) as fk:
            if H == 1:
                mB, sB, eB = be.gat_stats_scores(B, tbl_a_dst, a_src, slope)
            else:
                mB, sB = be.gat_stats(B, tbl_a_dst, a_src, H, slope)
            # constants of the graph, built once: which hub rows have no entry on this rank, and the fill values they take
            const = sg._gat_const.get(H) if hasattr(sg, "_gat_const") else None
            if const is None:
                empty = be.row_lengths(B).view(-1, 1) == 0
                const = (empty, torch.full_like(mB, NEG), torch.ones_like(mB), torch.zeros_like(sB))
                if not hasattr(sg, "_gat_const"):
                    sg._gat_const = {}
                sg._gat_const[H] = const
            empty, neg, ones, zeros = const
            M = torch.where(empty, neg, mB)
            _all_reduce(M, W, sg.small_group, op=dist.ReduceOp.MAX, tag="fwd_all_reduce_max")
            # the partial denominators go on the wire BEFORE the aggregation is even launched: small, and not behind the 0.4 GB of U
            S = torch.where(empty, zeros, sB * torch.exp(mB - M))
            wS = None
            if not _solo(W):
                s_own = S.new_empty((hp, H))
                wS = reduce_scatter_rows(S.contiguous(), s_own, sg.rank, W, sg.small_group, async_op=True)
            if H == 1:
                U = be.gat_aggregate_scores(B, h, None, C, eB, M, ones)            # sum exp(e - M) h_j, not normalised
            else:
                U = be.gat_aggregate(B, h, None, H, C, tbl_a_dst, a_src, M, ones, slope, False)
            if _solo(W):
                hU.copy_(U)
                s_own = S
            else:
                wU = reduce_scatter_rows(U, hU, sg.rank, W, sg.group, async_op=True)
                _wait(wS, "fwd_reduce_scatter_s", s_own)
        # light rows: the whole softmax is local once the hub table is here
        tbl_a_src_full = torch.cat([tbl_a_src, a_src])                             # index space of A's columns
        if H == 1:
            mA, sA, eA = be.gat_stats_scores(A, a_dst[:nL], tbl_a_src_full, slope)
            _wait(g_work, "fwd_all_gather", tbl_h)
            be.gat_aggregate_scores(A, tbl_h, h, C, eA, mA, sA, bias=bias, out=out_full[:nL])
        else:
            mA, sA = be.gat_stats(A, a_dst[:nL].contiguous(), tbl_a_src_full, H, slope)
            _wait(g_work, "fwd_all_gather", tbl_h)
            be.gat_aggregate(A, tbl_h, h, H, C, a_dst[:nL].contiguous(), tbl_a_src_full, mA, sA, slope, False, bias=bias,
                             out=out_full[:nL])
        fk.join(M, s_own)
        _wait(wU, "fwd_reduce_scatter", hU)
        if nH:
            hub = (hU[:nH].view(nH, H, C) / (s_own[:nH].view(nH, H, 1) + 1e-16)).reshape(nH, F)
            hU[:nH] = hub + bias if bias is not None else hub
        out = out_full[: nL + nH]
        m_own = torch.cat([mA, M[sg.own_hub]])
        s_all = torch.cat([sA, s_own[:nH]])
        tbl_S, _ = gather_hub(sg, s_all, small=True)                                           # every rank needs S of the hub targets it holds
        ctx.sg, ctx.H, ctx.C, ctx.slope = sg, H, C, float(slope)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x_own, weight, att2, h, a_dst, a_src, tbl_h, tbl_a_dst, tbl_a_src, M, tbl_S, m_own, s_all, out,
                              bias if bias is not None else h.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (x_own, weight, att2, h, a_dst, a_src, tbl_h, tbl_a_dst, tbl_a_src, M, tbl_S, m_own, s_all, out, bias) = ctx.saved_tensors
        sg: ShardedGraph = ctx.sg
        be, W = sg.backend, sg.world
        nL, nH, hp = sg.nL, sg.nH, sg.part.h_per
        H, C, slope = ctx.H, ctx.C, ctx.slope
        F = H * C
        A, B, At, Bt = sg.direct()
        map_a, map_b, n_bt, n_at = sg.gat_maps()
        dO = grad_out.contiguous()
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        if hasattr(be, "gat_rowdot_colsum"):                                       # D and the bias gradient in one pass over dOut, out
            D, db = be.gat_rowdot_colsum(dO, out, bias if ctx.has_bias else None, H, C, want_colsum=want_db)
        else:
            D, db = be.gat_rowdot(dO, out, bias if ctx.has_bias else None, H, C), (be.colsum(dO) if want_db else None)
        w_db = None
        if db is not None:
            w_db = _all_reduce(db, W, sg.small_group, tag="bwd_all_reduce_db", async_op=True)   # waited for at the end
        tbl_D, _ = gather_hub(sg, D, small=True)                                               # small, first (see forward)
        tbl_dO, g_work = gather_hub(sg, dO, async_op=True)                         # big: hub rows of dOut
        # packed per-TARGET scalars (a_dst, m, 1 / s, D): the own rows, and the hub table (targets of the light sources)
        # (one table [hub table ; own rows], the index space of A^T's columns; B^T's columns are the own rows)
        n_tbl = tbl_a_dst.numel()
        t_all = h.new_empty((n_tbl + a_dst.numel(), 4))
        t_own = be.gat_pack(a_dst, m_own, s_all, D, out=t_all[n_tbl:])
        be.gat_pack(tbl_a_dst, M, tbl_S, tbl_D, out=t_all[:n_tbl])
        # hub SOURCES (rows of the hub table): targets = this rank's rows (its light rows; the own hubs' loops): nothing remote,
        # so this pass and its row sums run on the partial stream beside the light sources' pass
        bs = be.partial_stream(h, sg.schedule) if (not _solo(W) and hasattr(be, "partial_stream")) else None
        # one head on the GPU backend: the by-source row sums of dz come out of the fused passes themselves (Schedule.gat_src_rowsum_fused)
        in_pass = H == 1 and getattr(be, "fused_rowsum", False) and sg.schedule.gat_src_rowsum_fused
        with _fork(bs, (dO, tbl_h, t_own, tbl_a_src)) as fk:
            pg_src = h.new_empty((Bt.n_rows, 1)) if in_pass else None
            kw_rs = {"rowsum_out": pg_src.view(-1)} if in_pass else {}
            pdh, dz_bt = be.gat_backward_fused(Bt, dO, None, tbl_h, C, t_own, tbl_a_src, slope, H=H, **kw_rs)
            dz_bt = dz_bt.view(-1, H)
            if not in_pass:
                pg_src = be.seg_rowsum(Bt, dz_bt, H)
        # own light SOURCES: targets = the hub table (+ the own loop)
        dh_full = h.new_empty((nL + hp, F))
        _wait(g_work, "bwd_all_gather", tbl_dO)
        g_src_l = h.new_empty((nL, 1)) if in_pass else None
        kw_rs = {"rowsum_out": g_src_l.view(-1)} if in_pass else {}
        _, dz_at = be.gat_backward_fused(At, tbl_dO, dO, h[:nL], C, t_all, a_src[:nL].contiguous(), slope,
                                         out=dh_full[:nL], H=H, **kw_rs)
        dz_at = dz_at.view(-1, H)
        if not in_pass:
            g_src_l = be.seg_rowsum(At, dz_at, H)
        fk.join(pdh, dz_bt, pg_src)
        dh = dh_full[: nL + nH]
        # One head on a shape the split GEMM covers: the attention terms g_dst (x) a1 + g_src (x) a2 are never added to d h --
        # dX takes them in its GEMM's store epilogue, dW as the outer-product correction P^T [a1; a2] with
        # P = x_own^T [g_dst g_src] (this rank's share: dW and d att are all-reduced anyway), d att = P W
        # (functional._GatConvFn._backward_rank2).  d h is then final as soon as its hub rows have been reduce-scattered, so
        # that exchange is issued FIRST and the weight-gradient GEMM runs on the side stream under the row sums of dz below.
        rank2 = (H == 1 and ctx.needs_input_grad[0] and weight.size(0) % 4 == 0 and hasattr(be, "linear_bwd_data_rank2")
                 and be.linear_bwd_data_rank2_ok(dh, weight, sg.schedule))
        wh = dws = None
        if not _solo(W):
            wh = reduce_scatter_rows(pdh, dh_full[nL:], sg.rank, W, sg.group, async_op=True)
        else:
            dh_full[nL:].copy_(pdh)
        dw = None
        if rank2 and ctx.needs_input_grad[1]:
            dws = be.side_stream(dh, sg.schedule) if hasattr(be, "side_stream") else None
            with _fork(dws, (x_own, dh_full, pdh)) as fw:
                _wait(wh, "bwd_reduce_scatter", dh_full)
                wh = None
                dw, _ = be.linear_bwd_weight(x_own, dh, False)
        dz_cat = torch.cat([dz_bt[:n_bt], dz_at[:n_at]])
        g_dst_l = be.seg_rowsum(A, dz_cat, H, map_=map_a)                          # light targets: complete
        pg_dst = be.seg_rowsum(B, dz_cat, H, map_=map_b)                           # hub targets: this rank's share
        pg = torch.cat([pg_dst, pg_src], dim=1).contiguous()                       # [hub_rows, 2 H]
        if _solo(W):
            g_hub = pg
        else:
            g_hub = pg.new_empty((hp, 2 * H))
            _wait(reduce_scatter_rows(pg, g_hub, sg.rank, W, sg.small_group, async_op=True), "bwd_reduce_scatter_g", g_hub)
            _wait(wh, "bwd_reduce_scatter", dh_full)
        g_dst = torch.cat([g_dst_l, g_hub[:nH, :H]])
        g_src = torch.cat([g_src_l, g_hub[:nH, H:]])
        datt = dx = None
        if rank2:
            K = weight.size(0)
            A2 = att2.view(2, C)
            U = be.gat_rank2_cols(weight, A2)                                      # [2, K]: W a1, W a2
            g_dst, g_src = g_dst.contiguous(), g_src.contiguous()
            P = be.gat_att_grad(x_own, g_dst, g_src, 1, K).view(2, K)
            if ctx.needs_input_grad[1]:
                fw.join(dw)
            datt = be.gat_rank2_tail(P, weight, A2, dw, ctx.needs_input_grad[2])   # dW += P^T [a1; a2], d att = P W
            # the parameter-gradient sums are on the wire while the dX GEMM runs
            w_att = w_dw = None
            if datt is not None:
                datt = datt.reshape(1, 2 * C)
                w_att = _all_reduce(datt, W, sg.small_group, tag="bwd_all_reduce_datt", async_op=True)
            if dw is not None:
                w_dw = _all_reduce(dw, W, sg.small_group, tag="bwd_all_reduce_dw", async_op=True)
            dx = be.linear_bwd_data_rank2(dh, weight, g_dst, g_src, U[0], U[1])
            _wait(w_att, "bwd_all_reduce_datt", datt)
            _wait(w_dw, "bwd_all_reduce_dw", dw)
            _wait(w_db, "bwd_all_reduce_db", db)
            if datt is not None:
                datt = datt.view(1, 1, 2 * C)
            return dx, dw, datt, db, None, None, None
        be.gat_rank1_add(dh, g_dst, g_src, att2, H, C)                             # d h_j += g_dst[j] att[:C] + g_src[j] att[C:]
        if ctx.needs_input_grad[2]:
            datt = be.gat_att_grad(h, g_dst.contiguous(), g_src.contiguous(), H, C)
            _all_reduce(datt, W, sg.small_group, tag="bwd_all_reduce_datt")
            datt = datt.view(1, H, 2 * C)
        if ctx.needs_input_grad[1]:
            dw, _ = be.linear_bwd_weight(x_own, dh, False)
            _all_reduce(dw, W, sg.small_group, tag="bwd_all_reduce_dw")
        if ctx.needs_input_grad[0]:
            dx = be.linear_bwd_data(dh, weight, None)
        _wait(w_db, "bwd_all_reduce_db", db)
        return dx, dw, datt, db, None, None, None


class _ShardedLayer(nn.Module):
    def __init__(self, sg: ShardedGraph, weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
        super().__init__()
        self.sg = sg
        self.weight = nn.Parameter(weight.clone())
        self.bias = nn.Parameter(bias.clone()) if bias is not None else None


class ShardedSAGELayer(_ShardedLayer):
    """SAGEConv (PyG 1.4.2 semantics, mean over in-neighbours and self, then ``@ W + b``) on a sharded
    graph.  Input and output are this rank's rows (``ShardedGraph.own``: light nodes, then hubs);
    parameters are replicated and their gradients all-reduced, as data-parallel training expects."""

    def forward(self, x_own: torch.Tensor) -> torch.Tensor:
        return _ShardedSageFn.apply(x_own, self.weight, self.bias, self.sg)


class ShardedGCNLayer(_ShardedLayer):
    """GCNConv (PyG 1.4.2: ``improved=False, normalize=True``, no edge weights) on a sharded graph."""

    def forward(self, x_own: torch.Tensor) -> torch.Tensor:
        if self.weight.size(0) <= self.weight.size(1):          # aggregate first: dW overlaps the backward aggregation
            return _ShardedSageFn.apply(x_own, self.weight, self.bias, self.sg, True)
        return _ShardedGcnFn.apply(x_own, self.weight, self.bias, self.sg)


class ShardedGATLayer(_ShardedLayer):
    """GATConv (PyG 1.4.2: ``concat=True, dropout=0``) on a sharded graph; ``att`` is ``[1, heads, 2 * out]``."""

    def __init__(self, sg: ShardedGraph, weight: torch.Tensor, att: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 heads: int = 1, negative_slope: float = 0.2):
        super().__init__(sg, weight, bias)
        self.att = nn.Parameter(att.clone())
        self.heads, self.negative_slope = int(heads), float(negative_slope)

    def forward(self, x_own: torch.Tensor) -> torch.Tensor:
        H = self.heads
        C = self.weight.size(1) // H
        fused_ok = C % 4 == 0 and H * C <= 256 and (H == 1 or (H in (2, 4, 8) and C in (32, 64, 128)))
        if self.sg.schedule.gat_direct and fused_ok and self.sg.direct_ok and hasattr(self.sg.backend, "gat_backward_fused"):
            return _ShardedGatDirectFn.apply(x_own, self.weight, self.att, self.bias, self.sg, H, self.negative_slope)
        return _ShardedGatFn.apply(x_own, self.weight, self.att, self.bias, self.sg, self.heads, self.negative_slope)


# -------------------------------------------------------------------------------------------------------------
# the north-star's baseline split: edge shards over a replicated x, all-reduce of the partial [N, F] sums
# -------------------------------------------------------------------------------------------------------------
class EdgeShardedGraph:
    """"Shard the edge list": rank r walks ITS SLICE of edge_index -- columns [r E / W, (r + 1) E / W), as they come --
    over a full replica of ``x``, plus the self loops of its block of output rows; the partial sums of ALL rows are
    all-reduced.  Outputs are the contiguous row block ``[lo, hi)`` of this rank.  ``sliced=True``: ``edge_index``
    already is the slice (no rank holds the whole list); the global in-degree comes from an all-reduce of the slices'
    counts (``in_count``: supplied by the caller when there is no process group -- virtual ranks)."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, rank: int, world: int, device, backend=None, group=None,
                 sliced: bool = False, in_count: Optional[torch.Tensor] = None):
        self.N, self.rank, self.world, self.group = int(num_nodes), rank, world, group
        self.backend = be = backend or HipBackend()
        if not sliced:
            E = edge_index.size(1)
            edge_index = edge_index[:, rank * E // world: (rank + 1) * E // world]
        ei = edge_index.to(device)
        keep = ei[0] != ei[1]
        src, dst = ei[0][keep], ei[1][keep]
        self.per = (self.N + world - 1) // world                      # padded rows per rank
        self.lo, self.hi = min(rank * self.per, self.N), min((rank + 1) * self.per, self.N)
        blk = torch.arange(self.lo, self.hi, device=device)
        self.fwd = be.make_side(torch.cat([dst, blk]), torch.cat([src, blk]), self.N, self.N)
        self.bwd = be.make_side(torch.cat([src, blk]), torch.cat([dst, blk]), self.N, self.N)
        self.local_nnz = int(src.numel())
        if in_count is None:
            in_count = torch.bincount(dst, minlength=self.N)
            if world > 1:
                if not (dist.is_available() and dist.is_initialized()):
                    raise ValueError("EdgeShardedGraph: world > 1 needs a process group or the global in_count")
                dist.all_reduce(in_count, group=group)
        cnt = in_count.to(device=device, dtype=torch.float32) + 1.0
        self.inv_cnt = (1.0 / cnt[self.lo:self.hi]).contiguous()

    def shard(self, x_full: torch.Tensor) -> torch.Tensor:
        return x_full[self.lo:self.hi]


class _EdgeShardedSageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_full, weight, bias, sg: EdgeShardedGraph):
        be = sg.backend
        part = be.segsum(sg.fwd, x_full.contiguous())                  # partial sums of ALL rows over this rank's entries
        _all_reduce(part, sg.world, sg.group, tag="fwd_all_reduce")
        agg = part[sg.lo:sg.hi] * sg.inv_cnt.view(-1, 1)
        out = be.linear_fwd(agg.contiguous(), weight, bias)
        ctx.sg = sg
        ctx.has_bias = bias is not None
        ctx.save_for_backward(agg, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        agg, weight = ctx.saved_tensors
        sg: EdgeShardedGraph = ctx.sg
        be, W = sg.backend, sg.world
        grad_out = grad_out.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = be.linear_bwd_weight(agg, grad_out, ctx.has_bias)
            _all_reduce(dw, W, sg.group, tag="bwd_all_reduce_dw")
            if db is not None:
                _all_reduce(db, W, sg.group, tag="bwd_all_reduce_db")
        if ctx.needs_input_grad[0]:
            dagg = be.linear_bwd_data(grad_out, weight, sg.inv_cnt)     # this rank's row block
            full = dagg.new_empty((W * sg.per, dagg.size(1)))
            if _solo(W):
                full[: dagg.size(0)] = dagg
            else:
                block = dagg
                if dagg.size(0) != sg.per:                              # last rank(s): pad the block
                    block = dagg.new_zeros((sg.per, dagg.size(1)))
                    block[: dagg.size(0)] = dagg
                _wait(all_gather_rows(block.contiguous(), full, W, sg.group, async_op=True), "bwd_all_gather", full)
            dx = be.segsum(sg.bwd, full[: sg.N])                        # partial dX of ALL rows
            _all_reduce(dx, W, sg.group, tag="bwd_all_reduce")          # x is replicated: its gradient is the sum
        return dx, dw, db, None


class EdgeShardedSAGELayer(_ShardedLayer):
    """SAGEConv with the north-star's edge split: input = the full (replicated) ``x``, output = this rank's block of
    rows; the gradient of ``x`` comes back complete on every rank."""

    def forward(self, x_full: torch.Tensor) -> torch.Tensor:
        return _EdgeShardedSageFn.apply(x_full, self.weight, self.bias, self.sg)
