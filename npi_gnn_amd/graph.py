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
    if any(int(v) & 4 for v in values):
        raise ValueError("InteractionGraph.batch: the n_nodes / n_pairs it was given do not belong to its keys (the device counted "
                         "other totals); that batch holds placeholders only (node 0 / graph 0 rows, padding edges, zero features).  "
                         "Totals must come from sizes() of the same keys")


def note_status(status: torch.Tensor) -> None:
    """File a device status word: checked at once under ``set_debug(True)``, otherwise at the next device read of this package
    (or after ``_PENDING_MAX`` unchecked words); never during a HIP-graph capture."""
    if torch.cuda.is_current_stream_capturing():
        return                                # no reads inside a capture, and the word lives in the graph's pool
    if _DEBUG:
        raise_on_status([status.item()])
        return
    _PENDING.append(status)
    if len(_PENDING) > _PENDING_MAX:
        check_pending()


def check_pending() -> None:
    """Read and check every outstanding status word now (synchronises)."""
    st = pending_status()
    if st:
        by_dev = {}
        for t in st:
            by_dev.setdefault(t.device, []).append(t)
        for ts in by_dev.values():
            raise_on_status(torch.cat(ts).tolist())


#: aggregation scratch buffers kept per side (one per (width, stream) pair in use: launch stream, side stream, partial stream, a capture's)
CARRY_SLOTS = 6


@dataclass
class CSRSide:
    """One orientation.  ``rowptr[N+1]``, ``col/eid/rowidx[nnz_max]`` int32 in entry order;
    ``item_row`` maps every item of ``item`` entries (64 or 256: fixed when the side is built, and handed to every kernel that
    walks ``item_row`` -- no process-wide setting can change the meaning of an existing side) to its first row;
    ``rowptr[N]`` is nnz on device."""
    rowptr: torch.Tensor
    col: torch.Tensor
    eid: torch.Tensor
    rowidx: torch.Tensor
    item_row: torch.Tensor
    status: torch.Tensor
    nnz_max: int
    n_items: int
    _inv_cnt: Optional[torch.Tensor] = None
    _carry: Dict[tuple, torch.Tensor] = field(default_factory=dict)
    n_rows: int = -1          # output rows (key id space)
    n_cols: int = -1          # rows of the feature table the entries index (== n_rows unless sharded)
    item: int = 0             # entries per item this side was cut with

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
        """f32 scratch of the aggregation launches (partial sums of rows cut by a workgroup boundary + their arrival
        counters); reused across calls of the same width ON THE SAME STREAM.  ZEROED once, here: every launch leaves its
        counters at zero again (``npi_segsum_carry_elems``).  One buffer per (side, width, current stream): launches of one
        stream are ordered, so they may share it; two modules that walk the same side at the same width on two streams
        (possibly concurrently) get a buffer each.  ``reset_carry()`` after a launch that did not run to its end."""
        dev = self.rowptr.device
        key = (F, _lib.stream_ptr(dev))
        buf = self._carry.pop(key, None)
        if buf is None:
            n = int(load().npi_segsum_carry_elems(self.nnz_max, self.item, F))
            buf = torch.zeros(n, dtype=torch.float32, device=dev)
            # at most CARRY_SLOTS buffers per side (at the C5 size one is ~200 MB): the least recently used one goes when a new
            # (width, stream) pair appears -- every fresh stream of a capture or a bench run used to leave its own behind for the
            # side's lifetime.  A dropped buffer was allocated and used on ITS stream only, so the caching allocator's stream-
            # ordered reuse keeps a launch that is still in flight on it safe; never during a capture (the graph owns its memory).
            if len(self._carry) >= CARRY_SLOTS and not torch.cuda.is_current_stream_capturing():
                self._carry.pop(next(iter(self._carry)))
        elif _DEBUG and not torch.cuda.is_current_stream_capturing():
            self._check_counters(buf)
        self._carry[key] = buf                              # (re-inserted: dicts keep insertion order, the first key is the LRU)
        return buf

    def _check_counters(self, buf: torch.Tensor) -> None:
        """``set_debug(True)``: the arrival counters at the head of ``buf`` must be zero between launches (synchronises)"""
        n_wg = -(-self.n_items // 4)
        n = n_wg + 2 * (-(-n_wg // 64))
        if n and int(buf[:n].view(torch.int32).abs().max()) != 0:
            raise _lib.NpiError("aggregation scratch: an arrival counter is not zero between launches -- two launches shared the "
                                "buffer concurrently or one was aborted; call reset_carry() on this side")

    def reset_carry(self) -> None:
        """Drop every scratch buffer of this side (after an aborted launch or a failed ``check()``): the next launch of each
        (width, stream) gets a freshly zeroed one."""
        self._carry.clear()


def item_hint(nnz_max: int) -> int:
    """Recommended item size for a new CSR of this capacity (``npi_item_edges``, a pure function: 64 below 2^22 entries, else
    256).  Asked ONCE, when a side is built without ``item=``; the side then carries its own value.  Neither the library nor this
    module keeps a threshold that could move (``CSRGraph(item=)`` / ``build_side(item=)`` name another size explicitly)."""
    return int(load().npi_item_edges(int(nnz_max)))


def build_side(key: torch.Tensor, val: torch.Tensor, n_rows: int, n_cols: int, self_loops: bool = True,
               loop_col_offset: int = 0, drop_equal: bool = True, item: Optional[int] = None,
               sort_columns: bool = False) -> CSRSide:
    """CSR over ``n_rows`` key rows whose entries index a table of ``n_cols`` rows
    (``npi_csr_build_ex``); with ``n_cols == n_rows`` and the defaults this is the plain build.  ``item``: entries per
    item (64 or 256; default: ``item_hint``).  ``sort_columns``: a row's entries ordered by column instead of list order
    (``NPI_CSR_SORT_COLUMNS``; the appended self loop stays last)."""
    lib = load()
    dev = require_gpu(key, val)
    E, N = int(key.numel()), int(n_rows)
    nnz_max = E + (N if self_loops else 0)
    item = item_hint(nnz_max) if item is None else int(item)
    n_items = int(lib.npi_num_items(nnz_max, item))
    if n_items < 0:
        raise ValueError(f"build_side: item must be 64 or 256 (got {item})")
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
                               int(loop_col_offset), (1 if drop_equal else 0) | (2 if sort_columns else 0), ptr(rowptr), ptr(col), ptr(eid),
                               ptr(rowidx), ptr(item_row), item, ptr(status), ptr(ws), ws_bytes, stream_ptr(dev)),
          "npi_csr_build_ex")
    note_status(status)
    side = CSRSide(rowptr, col, eid, rowidx, item_row, status, nnz_max, n_items)
    side.n_rows, side.n_cols, side.item = N, int(n_cols), item
    return side


class CSRGraph:
    """Self-loop-augmented adjacency of one (batched) graph on one GPU.

    ``by_dst`` groups entries by target node (forward aggregation, PyG flow ``source_to_target``);
    ``by_src`` groups them by source node (backward: dX = A^T ...), built lazily."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, self_loops: bool = True, by_dst: Optional[CSRSide] = None,
                 symmetric: bool = False, item: Optional[int] = None, sort_columns: bool = False, keep_equal: bool = False):
        """``item``: entries per item of both sides (64 or 256; default: the hint for this capacity, ``item_hint``).
        ``sort_columns``: both sides with every row's entries in column order (a second key for the build's sort: at 4M nodes /
        100M edges 3 ms more per side, once, for 1.2 % of every GATConv layer step and 0.6 % of every SAGEConv one -- the
        4-byte gathers of per-node scalars walk ascending addresses; EXPERIMENTS A22).  The sums are the same in another order.
        ``self_loops=False, keep_equal=True``: the edge list exactly as it is -- no loop appended, existing ``(i, i)`` columns kept
        as ordinary entries (what PyG's ``SAGEConv(concat=True)`` aggregates over; the default drops them first, as
        ``add_remaining_self_loops`` does).
        ``by_dst``: a by-target side somebody already derived for this very edge list (``filtered_side``): not rebuilt.
        ``symmetric``: the producer of the edge list vouches that it holds every edge in both directions (the device-side
        subgraph extraction emits both, src/classes.py:701-704, and filter_adj keeps the property; ``GraphBatch.symmetric``).
        Then row j of the by-source CSR holds the same neighbours as row j of the by-target CSR -- in another order -- and an
        UNWEIGHTED transposed aggregation can walk the by-target side: no second sort (functional._SageConvFn)."""
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("edge_index must be a LongTensor of shape [2, E]")
        require_gpu(edge_index)
        self.num_nodes = int(num_nodes)
        self.num_edges = int(edge_index.size(1))
        self.self_loops = bool(self_loops)
        self.device = edge_index.device
        # what this CSR was built from: a GraphBatch reuses it only for THIS tensor in THIS state (an in-place edit of the
        # edge list bumps ``_version`` and the next conv call sorts again -- ``built_from``)
        self._ei_version = edge_index._version
        self._ei_ptr = edge_index.data_ptr()
        self.symmetric = bool(symmetric)
        # rows of a [2,E] tensor are contiguous when the tensor is; otherwise copy (index plumbing)
        self._src = edge_index[0].contiguous()
        self._dst = edge_index[1].contiguous()
        self._item = item if by_dst is None else by_dst.item
        self.sort_columns = bool(sort_columns)
        self.keep_equal = bool(keep_equal)
        if self.keep_equal and self.self_loops:
            raise ValueError("CSRGraph: keep_equal=True (existing self loops stay ordinary entries) needs self_loops=False")
        self.by_dst = by_dst if by_dst is not None else \
            build_side(self._dst, self._src, self.num_nodes, self.num_nodes, self.self_loops, drop_equal=not self.keep_equal,
                       item=self._item, sort_columns=self.sort_columns)
        self._by_src: Optional[CSRSide] = None

    @property
    def by_src(self) -> CSRSide:
        if self._by_src is None:
            self._by_src = build_side(self._src, self._dst, self.num_nodes, self.num_nodes, self.self_loops,
                                      drop_equal=not self.keep_equal, item=self.by_dst.item, sort_columns=self.sort_columns)
        return self._by_src

    def built_from(self, edge_index: torch.Tensor, num_nodes: Optional[int] = None) -> bool:
        """True while ``edge_index`` is still the tensor this CSR was built from: same storage, same shape and, above all,
        the same ``_version`` (every in-place write bumps it)."""
        return (self.num_edges == edge_index.size(1) and self._ei_version == edge_index._version
                and self._ei_ptr == edge_index.data_ptr() and (num_nodes is None or self.num_nodes == int(num_nodes)))

    def inv_count(self, side: CSRSide) -> torch.Tensor:
        return side.inv_count()

    def carry(self, side: CSRSide, F: int) -> torch.Tensor:
        return side.carry(F)

    def nnz(self) -> int:
        """Entries incl. self loops (device read; synchronises -- and reports dropped out-of-range ids)."""
        n = int(self.by_dst.rowptr[-1].item())
        check_pending()
        return n


@dataclass
class CSRRecipe:
    """How to obtain the by-target CSR of the graph a pooling layer leaves WITHOUT sorting: the parent's side filtered by the
    kept nodes (``filtered_side``).  Made by ``pool.topk_pool``, consumed -- at most once, and only if a conv asks -- by
    ``GraphBatch.graph``."""
    parent: CSRSide
    perm: torch.Tensor            # int32 [n_out]  new -> old node id
    remap: torch.Tensor           # int32 [N]      old -> new node id or -1
    newpos: torch.Tensor          # int32 [E]      where filter_adj put every input edge
    n_out: int
    num_edges_out: int
    ei_version: int               # ``edge_index._version`` of the pooled edge list when the recipe was written


class GraphBatch:
    """One batch of graphs as the layers of ``Net_1`` hand it to one another: the node features
    and everything this package knows about the structure they live on, in ONE explicit object -- nothing rides on tensor
    attributes.  ``SAGEConv`` / ``GCNConv`` / ``GATConv`` map a ``GraphBatch`` to a ``GraphBatch`` (new ``x``, same structure),
    ``TopKPooling`` to the pooled one, ``global_max_mean_pool`` reads its segments.  The layers still take PyG's plain
    ``(x, edge_index[, batch])`` tensors; what they then cannot know is exactly the optional part below.

    x           [n, F] float32 node features
    edge_index  [2, E] int64, PyG layout (row 0 = source, row 1 = target); may end in (-1, -1) padding columns when it comes
                from ``TopKPooling(padded_edges=True)`` -- every consumer in this package drops them
    batch       [n] int64 graph id per node, non-decreasing (None: one graph)
    num_graphs  B (None: read from ``batch`` when needed)
    sizes       host LongTensor [B], nodes per graph, when known WITHOUT a device read (``net1.KeyLoader``: a sample's
                size depends on its key only; a pooling layer: ceil(ratio n_g)) -- lets TopKPooling run without a read-back
    graph_ptr   device int32 [B+1] segment starts of ``batch`` (``segment_ptr()`` builds them on first use)
    symmetric   the edge list holds every edge in both directions (``CSRGraph.symmetric``)
    pad_base    ``x`` is the leading columns of this wider buffer whose other columns are zero (``InteractionGraph.batch``:
                178 -> 256): the first conv runs its GEMMs on the padded width
    The CSR of ``edge_index`` is built by ``graph()`` on first use and kept for as long as the tensor is unchanged
    (``CSRGraph.built_from``: storage, shape and ``_version``); a pooled batch may carry a ``CSRRecipe`` instead."""

    __slots__ = ("x", "edge_index", "batch", "num_graphs", "sizes", "graph_ptr", "symmetric", "pad_base", "_csr", "_recipe")

    def __init__(self, x: torch.Tensor, edge_index: torch.Tensor, batch: Optional[torch.Tensor] = None,
                 num_graphs: Optional[int] = None, *, sizes: Optional[torch.Tensor] = None,
                 graph_ptr: Optional[torch.Tensor] = None, symmetric: bool = False, pad_base: Optional[torch.Tensor] = None,
                 csr: Optional[CSRGraph] = None, recipe: Optional[CSRRecipe] = None):
        if isinstance(edge_index, CSRGraph):
            raise TypeError("GraphBatch: edge_index is the [2, E] tensor; pass a prebuilt CSRGraph as csr=")
        self.x, self.edge_index, self.batch = x, edge_index, batch
        if num_graphs is None and sizes is not None:
            num_graphs = int(sizes.numel())
        if num_graphs is None and graph_ptr is not None:
            num_graphs = int(graph_ptr.numel()) - 1
        if num_graphs is None and batch is None:
            num_graphs = 1
        self.num_graphs = None if num_graphs is None else int(num_graphs)
        if sizes is not None and self.num_graphs is not None and int(sizes.numel()) != self.num_graphs:
            raise ValueError(f"GraphBatch: sizes has {int(sizes.numel())} entries for {self.num_graphs} graphs")
        self.sizes, self.graph_ptr = sizes, graph_ptr
        self.symmetric, self.pad_base = bool(symmetric), pad_base
        self._csr, self._recipe = csr, recipe

    # ---- unpacking as PyG's (x, edge_index, batch) ----------------------------------------------------------------------------
    def __iter__(self):
        return iter((self.x, self.edge_index, self.batch))

    @property
    def num_nodes(self) -> int:
        return int(self.x.size(0))

    def to(self, device):
        """The reference's ``data = data.to(device)`` (src/train_with_twoDataset.PY:50).  Batches of this package are born on
        the device (``InteractionGraph.batch``): then this is the object itself.  Otherwise the tensors are moved; what was
        built for the old device (CSR, recipe, padded buffer) is dropped, the host-side knowledge (``sizes``) stays."""
        device = torch.device(device) if device is not None else self.x.device
        if device.type == "cuda" and device.index is None and self.x.is_cuda:
            device = self.x.device
        if device == self.x.device:
            return self
        mv = lambda t: None if t is None else t.to(device)                      # noqa: E731
        other = object.__new__(type(self))
        GraphBatch.__init__(other, mv(self.x), mv(self.edge_index), mv(self.batch), self.num_graphs, sizes=self.sizes,
                            graph_ptr=mv(self.graph_ptr), symmetric=self.symmetric)
        for klass in type(self).__mro__:                                       # subclasses' own fields (net1.Batch.y)
            for name in getattr(klass, "__slots__", ()):
                if name not in GraphBatch.__slots__ and hasattr(self, name):
                    v = getattr(self, name)
                    setattr(other, name, v.to(device) if isinstance(v, torch.Tensor) else v)
        return other

    def with_x(self, x: torch.Tensor) -> "GraphBatch":
        """The same graphs with other node features (what a conv returns); the structure, incl. a CSR already built, is shared."""
        return GraphBatch(x, self.edge_index, self.batch, self.num_graphs, sizes=self.sizes, graph_ptr=self.graph_ptr,
                          symmetric=self.symmetric, pad_base=None, csr=self._csr, recipe=self._recipe)

    # ---- structure, built on demand ------------------------------------------------------------------------------------------------
    def peek_graph(self) -> Optional[CSRGraph]:
        """The CSR if one was built AND ``edge_index`` has not been written to since; never builds."""
        g = self._csr
        if g is not None and g.built_from(self.edge_index, self.num_nodes):
            return g
        return None

    def graph(self) -> CSRGraph:
        """CSR of ``edge_index`` (+ self loops).  Reused while the edge list is untouched; an in-place edit (``_version``)
        makes the next call sort again, so a stale adjacency is never served."""
        g = self.peek_graph()
        if g is not None:
            return g
        n, ei = self.num_nodes, self.edge_index
        r, self._recipe = self._recipe, None                   # consumed either way: the parent's arrays are not kept alive
        g = None
        if r is not None and r.ei_version == ei._version and r.n_out == n and r.num_edges_out == ei.size(1):
            side = filtered_side(r.parent, r.perm, r.remap, r.newpos, r.n_out, r.num_edges_out)
            if side is not None:
                g = CSRGraph(ei, n, by_dst=side, symmetric=self.symmetric)
        if g is None:
            g = CSRGraph(ei, n, symmetric=self.symmetric)
        self._csr = g
        return g

    def segment_ptr(self) -> torch.Tensor:
        """int32 ``[B+1]`` segment starts of ``batch`` on the device (kept)."""
        if self.graph_ptr is None:
            dev = self.x.device
            if self.batch is None:
                self.graph_ptr = torch.tensor([0, self.num_nodes], dtype=torch.int32, device=dev)
            else:
                from .pool import graph_ptr as _graph_ptr
                self.graph_ptr = _graph_ptr(self.batch, self.num_graphs)
            if self.num_graphs is None:
                self.num_graphs = int(self.graph_ptr.numel()) - 1
        return self.graph_ptr

    def batch_vector(self) -> torch.Tensor:
        if self.batch is None:
            self.batch = torch.zeros(self.num_nodes, dtype=torch.int64, device=self.x.device)
        return self.batch

    def __repr__(self):
        return (f"GraphBatch(nodes={self.num_nodes}, edges={int(self.edge_index.size(1))}, graphs={self.num_graphs}, "
                f"F={int(self.x.size(1)) if self.x.dim() == 2 else '?'})")


def as_graph(edge_index_or_graph, num_nodes: int) -> CSRGraph:
    """What the convs aggregate over: a prebuilt ``CSRGraph``, a ``GraphBatch`` (its cached / derived CSR) or a plain
    ``edge_index`` tensor (sorted here, every call: a tensor carries no state)."""
    if isinstance(edge_index_or_graph, GraphBatch):
        edge_index_or_graph = edge_index_or_graph.graph()
    if isinstance(edge_index_or_graph, CSRGraph):
        g = edge_index_or_graph
        if g.num_nodes != num_nodes:
            raise ValueError(f"CSRGraph was built for {g.num_nodes} nodes, x has {num_nodes}")
        return g
    return CSRGraph(edge_index_or_graph, num_nodes)


def filtered_side(parent: CSRSide, perm: torch.Tensor, remap: torch.Tensor, newpos: torch.Tensor, n_out: int,
                  num_edges_out: int) -> Optional[CSRSide]:
    """By-target side of the graph TopKPooling leaves -- nodes ``perm`` (int32, new -> old) of the parent, renumbered by
    ``remap``, edges filtered in order (``newpos``: where ``npi_filter_adj`` put every input edge) -- derived from the
    parent's side without a sort (``npi_csr_filter``: three launches).  Identical to ``build_side`` on the filtered edge list of length
    ``num_edges_out`` (the padded length when the list is padded).  None when the shape is outside the kernel's range."""
    lib = load()
    dev = parent.rowptr.device
    n_out = int(n_out)
    if n_out > int(lib.npi_csr_filter_max_rows()) or parent.n_rows != parent.n_cols:
        return None
    nnz_max = int(num_edges_out) + n_out
    item = item_hint(nnz_max)
    n_items = int(lib.npi_num_items(nnz_max, item))
    i32 = dict(dtype=torch.int32, device=dev)
    rowptr = torch.empty(n_out + 1, **i32)
    col = torch.empty(max(nnz_max, 1), **i32)
    eid = torch.empty(max(nnz_max, 1), **i32)
    rowidx = torch.empty(max(nnz_max, 1), **i32)
    item_row = torch.empty(n_items + 1, **i32)
    status = torch.empty(1, **i32)
    ws = torch.empty(int(lib.npi_csr_filter_workspace_elems(n_out)), **i32)
    check(lib.npi_csr_filter(ptr(parent.rowptr), ptr(parent.col), ptr(parent.eid), ptr(perm), ptr(remap), ptr(newpos), n_out,
                             nnz_max, ptr(rowptr), ptr(col), ptr(eid), ptr(rowidx), ptr(item_row), item, ptr(status), ptr(ws),
                             stream_ptr(dev)), "npi_csr_filter")
    side = CSRSide(rowptr, col, eid, rowidx, item_row, status, nnz_max, n_items)
    side.n_rows = side.n_cols = n_out
    side.item = item
    return side
