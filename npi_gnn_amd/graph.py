"""Device-resident adjacency for the conv hot path: destination-sorted CSR (+ its transpose).

Takes the ``edge_index`` LongTensor ``[2, E]`` exactly as PyG's ``Batch`` collate hands it to the
conv (row 0 = source j, row 1 = target i; reference ``src/classes.py:701-704`` emits both
directions) and builds, on the GPU, what PyG 1.4.2's ``add_remaining_self_loops`` + scatter would
produce implicitly.  A ``CSRGraph`` may be passed to the convs in place of ``edge_index`` so that a
static full-batch graph (BASELINE.json configs 3-5) is sorted once instead of once per layer call.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import check, load, ptr, require_gpu, stream_ptr

_DEBUG = False

# Status words of graph builds that nobody has looked at yet.  A build never synchronises, so an out-of-range node id
# (which the build DROPS, where PyG's index_select / scatter raise) is reported at the next point that reads from the
# device anyway: TopKPooling's size read, InteractionGraph.batch, CSRGraph.nnz(), the metrics read -- or after
# _PENDING_MAX unchecked builds (one extra device read per that many builds).  set_debug(True) checks every build at once.
_PENDING = []
_PENDING_MAX = 256


def set_debug(flag: bool) -> None:
    """When on, every graph build synchronises and raises on out-of-range node ids
    (PyG/torch raise IndexError/RuntimeError there)."""
    global _DEBUG
    _DEBUG = bool(flag)


def pending_status(device=None) -> list:
    """Device status words not yet checked (callers that are about to read from the device append them to their read and
    hand the values to ``raise_on_status``).  ``device``: only that GPU's words are taken; the others stay pending."""
    if device is None:
        out = list(_PENDING)
        _PENDING.clear()
        return out
    out = [t for t in _PENDING if t.device == device]
    _PENDING[:] = [t for t in _PENDING if t.device != device]
    return out


def raise_on_status(values) -> None:
    if any(int(v) & 1 for v in values):
        raise IndexError("edge_index holds a node id outside [0, num_nodes): the edges were dropped by the graph build "
                         "(PyG's index_select / scatter raise here)")


def check_pending() -> None:
    """Read and check every outstanding status word now (synchronises)."""
    st = pending_status()
    if st:
        by_dev = {}
        for t in st:
            by_dev.setdefault(t.device, []).append(t)
        for ts in by_dev.values():
            raise_on_status(torch.cat(ts).tolist())


@dataclass
class CSRSide:
    """One orientation.  ``rowptr[N+1]``, ``col/eid/rowidx[nnz_max]`` int32 in entry order;
    ``item_row`` maps every 256-entry item to its first row; ``rowptr[N]`` is nnz on device."""
    rowptr: torch.Tensor
    col: torch.Tensor
    eid: torch.Tensor
    rowidx: torch.Tensor
    item_row: torch.Tensor
    status: torch.Tensor
    nnz_max: int
    n_items: int
    _inv_cnt: Optional[torch.Tensor] = None
    _carry: Dict[int, torch.Tensor] = field(default_factory=dict)
    n_rows: int = -1          # output rows (key id space)
    n_cols: int = -1          # rows of the feature table the entries index (== n_rows unless sharded)

    def inv_count(self) -> torch.Tensor:
        """1 / max(row length, 1): the scatter_mean divisor (count includes the self loop)."""
        if self._inv_cnt is None:
            dev = self.rowptr.device
            out = torch.empty(self.n_rows, dtype=torch.float32, device=dev)
            check(load().npi_row_inv_count(ptr(self.rowptr), self.n_rows, ptr(out), stream_ptr(dev)),
                  "npi_row_inv_count")
            self._inv_cnt = out
        return self._inv_cnt

    def carry(self, F: int) -> torch.Tensor:
        """f32 scratch for rows cut by an item boundary; reused across calls of the same width."""
        buf = self._carry.get(F)
        if buf is None:
            n = int(load().npi_segsum_carry_elems(self.nnz_max, F))
            buf = torch.empty(n, dtype=torch.float32, device=self.rowptr.device)
            self._carry[F] = buf
        return buf


def build_side(key: torch.Tensor, val: torch.Tensor, n_rows: int, n_cols: int, self_loops: bool = True,
               loop_col_offset: int = 0, drop_equal: bool = True) -> CSRSide:
    """CSR over ``n_rows`` key rows whose entries index a table of ``n_cols`` rows
    (``npi_csr_build_ex``); with ``n_cols == n_rows`` and the defaults this is the plain build."""
    lib = load()
    dev = require_gpu(key, val)
    E, N = int(key.numel()), int(n_rows)
    nnz_max = E + (N if self_loops else 0)
    n_items = int(lib.npi_num_items(nnz_max))
    i32 = dict(dtype=torch.int32, device=dev)
    rowptr = torch.empty(N + 1, **i32)
    col = torch.empty(max(nnz_max, 1), **i32)
    eid = torch.empty(max(nnz_max, 1), **i32)
    rowidx = torch.empty(max(nnz_max, 1), **i32)
    item_row = torch.empty(n_items + 1, **i32)
    status = torch.empty(1, **i32)
    ws_bytes = int(lib.npi_csr_workspace_bytes(E, N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.npi_csr_build_ex(ptr(key), ptr(val), E, N, int(n_cols), 1 if self_loops else 0,
                               int(loop_col_offset), 1 if drop_equal else 0, ptr(rowptr), ptr(col), ptr(eid),
                               ptr(rowidx), ptr(item_row), ptr(status), ptr(ws), ws_bytes, stream_ptr(dev)),
          "npi_csr_build_ex")
    if torch.cuda.is_current_stream_capturing():
        pass                                  # inside a HIP-graph capture: no reads, and the word lives in the graph's pool
    elif _DEBUG:
        raise_on_status([status.item()])
    else:
        _PENDING.append(status)
        if len(_PENDING) > _PENDING_MAX:
            check_pending()
    side = CSRSide(rowptr, col, eid, rowidx, item_row, status, nnz_max, n_items)
    side.n_rows, side.n_cols = N, int(n_cols)
    return side


def _build_side(key: torch.Tensor, val: torch.Tensor, E: int, N: int, self_loops: bool) -> CSRSide:
    return build_side(key, val, N, N, self_loops)


class CSRGraph:
    """Self-loop-augmented adjacency of one (batched) graph on one GPU.

    ``by_dst`` groups entries by target node (forward aggregation, PyG flow ``source_to_target``);
    ``by_src`` groups them by source node (backward: dX = A^T ...), built lazily."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, self_loops: bool = True, by_dst: Optional[CSRSide] = None):
        """``by_dst``: a by-target side somebody already derived for this very edge list (``filtered_side``): not rebuilt."""
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("edge_index must be a LongTensor of shape [2, E]")
        require_gpu(edge_index)
        self.num_nodes = int(num_nodes)
        self.num_edges = int(edge_index.size(1))
        self.self_loops = bool(self_loops)
        self.device = edge_index.device
        # what this CSR was built from: a cached graph is reused only for THIS tensor in THIS state (an in-place edit of the
        # edge list bumps ``_version`` and the next conv call sorts again -- ``cached_graph``)
        self._ei_version = edge_index._version
        self._ei_ptr = edge_index.data_ptr()
        # The producer of the edge list may vouch that it holds every edge in both directions (edge_index._npi_symmetric:
        # the device-side subgraph extraction emits both, src/classes.py:701-704, and filter_adj keeps the property).
        # Then row j of the by-source CSR holds the same neighbours as row j of the by-target CSR -- in another order --
        # and an UNWEIGHTED transposed aggregation can walk the by-target side: no second sort (functional._SageConvFn).
        self.symmetric = bool(getattr(edge_index, "_npi_symmetric", False))
        # rows of a [2,E] tensor are contiguous when the tensor is; otherwise copy (index plumbing)
        self._src = edge_index[0].contiguous()
        self._dst = edge_index[1].contiguous()
        self.by_dst = by_dst if by_dst is not None else \
            _build_side(self._dst, self._src, self.num_edges, self.num_nodes, self.self_loops)
        self._by_src: Optional[CSRSide] = None

    @property
    def by_src(self) -> CSRSide:
        if self._by_src is None:
            self._by_src = _build_side(self._src, self._dst, self.num_edges, self.num_nodes, self.self_loops)
        return self._by_src

    def inv_count(self, side: CSRSide) -> torch.Tensor:
        return side.inv_count()

    def carry(self, side: CSRSide, F: int) -> torch.Tensor:
        return side.carry(F)

    def nnz(self) -> int:
        """Entries incl. self loops (device read; synchronises -- and reports dropped out-of-range ids)."""
        n = int(self.by_dst.rowptr[-1].item())
        check_pending()
        return n


def cached_graph(edge_index: torch.Tensor, num_nodes: Optional[int] = None) -> Optional[CSRGraph]:
    """The CSR an earlier call left on this edge list -- only if the tensor is still the one it was built from: same
    storage, same shape and, above all, the same ``_version`` (every in-place write bumps it), so an edit of a cached edge
    list is never answered from the stale CSR."""
    g = getattr(edge_index, "_npi_graph", None)
    if (isinstance(g, CSRGraph) and g.num_edges == edge_index.size(1) and g._ei_version == edge_index._version
            and g._ei_ptr == edge_index.data_ptr() and (num_nodes is None or g.num_nodes == num_nodes)):
        return g
    return None


def as_graph(edge_index_or_graph, num_nodes: int) -> CSRGraph:
    if isinstance(edge_index_or_graph, CSRGraph):
        g = edge_index_or_graph
        if g.num_nodes != num_nodes:
            raise ValueError(f"CSRGraph was built for {g.num_nodes} nodes, x has {num_nodes}")
        return g
    # an edge list that is used again and again (the static batches of net1.GraphedEpoch) may carry its CSR
    g = cached_graph(edge_index_or_graph, num_nodes)
    if g is not None:
        return g
    # the edge list TopKPooling returned: its CSR is the parent's, filtered (no sort) -- unless somebody wrote to it since
    src = getattr(edge_index_or_graph, "_npi_graph_from", None)
    g = None
    if src is not None and len(src) == 7 and src[6] != edge_index_or_graph._version:
        src = None
    if src is not None and src[4] == num_nodes and src[5] == edge_index_or_graph.size(1):
        side = filtered_side(*src[:6])
        if side is not None:
            g = CSRGraph(edge_index_or_graph, num_nodes, by_dst=side)
        edge_index_or_graph._npi_graph_from = None           # the parent's arrays are not kept alive any longer
    if g is None:
        g = CSRGraph(edge_index_or_graph, num_nodes)
    if getattr(edge_index_or_graph, "_npi_symmetric", False):
        # an edge list this package produced itself (device-side extraction, filter_adj) and nobody modifies in place: its
        # CSR stays on it -- the pooling layer behind the conv derives the pooled graph's CSR from it (filtered_side)
        edge_index_or_graph._npi_graph = g
    return g


def filtered_side(parent: CSRSide, perm: torch.Tensor, remap: torch.Tensor, newpos: torch.Tensor, n_out: int,
                  num_edges_out: int) -> Optional[CSRSide]:
    """By-target side of the graph TopKPooling leaves -- nodes ``perm`` (int32, new -> old) of the parent, renumbered by
    ``remap``, edges filtered in order (``newpos``: where ``npi_filter_adj_ex`` put every input edge) -- derived from the
    parent's side without a sort (``npi_csr_filter``: three launches).  Identical to ``build_side`` on the filtered edge list of length
    ``num_edges_out`` (the padded length when the list is padded).  None when the shape is outside the kernel's range."""
    lib = load()
    dev = parent.rowptr.device
    n_out = int(n_out)
    if n_out > int(lib.npi_csr_filter_max_rows()) or parent.n_rows != parent.n_cols:
        return None
    nnz_max = int(num_edges_out) + n_out
    n_items = int(lib.npi_num_items(nnz_max))
    i32 = dict(dtype=torch.int32, device=dev)
    rowptr = torch.empty(n_out + 1, **i32)
    col = torch.empty(max(nnz_max, 1), **i32)
    eid = torch.empty(max(nnz_max, 1), **i32)
    rowidx = torch.empty(max(nnz_max, 1), **i32)
    item_row = torch.empty(n_items + 1, **i32)
    status = torch.empty(1, **i32)
    ws = torch.empty(int(lib.npi_csr_filter_workspace_elems(n_out)), **i32)
    check(lib.npi_csr_filter(ptr(parent.rowptr), ptr(parent.col), ptr(parent.eid), ptr(perm), ptr(remap), ptr(newpos), n_out,
                             nnz_max, ptr(rowptr), ptr(col), ptr(eid), ptr(rowidx), ptr(item_row), ptr(status), ptr(ws),
                             stream_ptr(dev)), "npi_csr_filter")
    side = CSRSide(rowptr, col, eid, rowidx, item_row, status, nnz_max, n_items)
    side.n_rows = side.n_cols = n_out
    return side


def attach_graph(edge_index: torch.Tensor, num_nodes: int) -> CSRGraph:
    """Build the CSR of ``edge_index`` once and leave it on the tensor: every later conv call on this very tensor reuses
    it instead of sorting again.  For edge lists that do not change (a loader that replays the same batches); the caller
    vouches that the tensor is not modified in place afterwards."""
    g = CSRGraph(edge_index, num_nodes)
    edge_index._npi_graph = g
    return g
