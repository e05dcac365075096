"""One-hop enclosing subgraphs, extracted and collated on the MI355X.

The reference builds every sample with per-node Python dict / set loops at dataset-build time
(``local_subgraph_generation``, reference ``src/classes.py:652-733``) and collates 200 of them per
step through the PyG ``DataLoader``.  ``InteractionGraph.batch(keys)`` produces the same batch --
``x [n, 1 + F]`` (structural label | node2vec | k-mer), ``edge_index [2, e]`` int64, ``batch [n]`` --
for any list of target pairs directly in HBM, ready for ``Net_1`` (wire format of SURVEY.md 8(f)-3).

Node order and features are bit-identical to the reference's; the edges are the same set, emitted
in list order (target pair, pairs of the RNA, pairs of the protein) instead of the reference's
Python-``set`` iteration order, which is a CPython hashing detail.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from ._lib import check, load, ptr, require_gpu, stream_ptr
from .graph import GraphBatch, build_side


class InteractionGraph:
    """The whole ncRNA-protein interaction graph of one project/fold, resident on the GPU.

    ``pairs [P, 2]``: (rna_serial, protein_serial) in ``interaction_list`` order (positives, then
    negatives; reference ``src/generate_dataset.py:224-305``).  ``usable [P]``: False for the pairs in
    ``set_allInteractionKey_cannotUse`` (the fold's test keys, ``:296-299``).  ``feat [N, F]``: node
    features by serial number.  Pairs must be unique and rna / protein serials disjoint.
    """

    def __init__(self, pairs: torch.Tensor, usable: torch.Tensor, feat: torch.Tensor, num_nodes: Optional[int] = None,
                 check_input: bool = True):
        dev = require_gpu(pairs, usable, feat)
        if feat.dtype != torch.float32:
            raise TypeError("feat must be float32")
        pairs = pairs.to(torch.int64)
        P = pairs.size(0)
        self.num_nodes = N = int(num_nodes) if num_nodes is not None else feat.size(0)
        if feat.size(0) < N:
            raise ValueError("feat has fewer rows than num_nodes")
        if check_input and P:
            code = pairs[:, 0] * N + pairs[:, 1]
            if torch.unique(code).numel() != P:
                raise ValueError("InteractionGraph: duplicate (rna, protein) pairs")
            is_rna = torch.zeros(N, dtype=torch.bool, device=dev)
            is_rna[pairs[:, 0]] = True
            if bool(is_rna[pairs[:, 1]].any()):
                raise ValueError("InteractionGraph: a serial number is used both as rna and as protein")
        # CSR over serial numbers; the builder's stable sort keeps interaction_list order inside a row
        key = torch.cat([pairs[:, 0], pairs[:, 1]]).contiguous()
        val = torch.cat([pairs[:, 1], pairs[:, 0]]).contiguous()
        side = build_side(key, val, N, N, False, 0, False)
        self.ptr, self.nbr = side.rowptr, side.col
        eid = side.eid[: 2 * P].to(torch.int64)
        self.ok = usable.to(device=dev, dtype=torch.bool)[eid % P].to(torch.uint8).contiguous() if P else \
            torch.zeros(1, dtype=torch.uint8, device=dev)
        self.feat = feat.contiguous()
        self.device = dev

    def sizes(self, keys: torch.Tensor):
        """(nodes, pairs) per sample of ``keys [K, 2]`` as HOST int64 tensors -- one device read for the whole key list.
        A loader that keeps them (``net1.KeyLoader``) can tell ``batch`` the size of every batch it asks for, and the
        per-batch device read below disappears (sample sizes depend on the key only, not on the batch it is in)."""
        lib = load()
        dev = self.device
        keys = keys.to(device=dev, dtype=torch.int32).contiguous()
        K = keys.size(0)
        i32 = dict(dtype=torch.int32, device=dev)
        node_off, pair_off = torch.empty(K + 1, **i32), torch.empty(K + 1, **i32)
        ws = torch.empty(max(2 * K, 1), **i32)
        check(lib.npi_subgraph_sizes(ptr(self.ptr), ptr(self.nbr), ptr(self.ok), ptr(keys), K, ptr(node_off), ptr(pair_off),
                                     ptr(ws), stream_ptr(dev)), "npi_subgraph_sizes")
        # the workspace holds, per sample, the usable partners of its RNA and of its protein (the scan only reads them)
        cnt = ws[: 2 * K].view(2, K).to(torch.int64).cpu() if K else torch.zeros((2, 0), dtype=torch.int64)
        both = cnt[0] + cnt[1]
        return both + 2, both + 1                               # nodes: the pair itself + partners; pairs: target + partners

    def batch(self, keys: torch.Tensor, return_node_id: bool = False, n_nodes: Optional[int] = None,
              n_pairs: Optional[int] = None, pad_features: bool = True):
        """``keys [B, 2]`` (rna_serial, protein_serial) -> the ``GraphBatch`` of the B enclosing subgraphs (it unpacks as
        ``x, edge_index, batch``; with ``return_node_id`` the pair ``(GraphBatch, node_id)``).
        ``n_nodes`` / ``n_pairs``: the batch's totals when the caller already knows them (``sizes``): no device read.  The fill
        kernels compare them with the device's own totals: a mismatch (totals of other keys) writes NOTHING and raises a status
        bit that the next device read of this package reports as a ``ValueError`` (``graph.check_pending``; at once under
        ``graph.set_debug(True)``).  ``pad_features=False``: ``x`` with its natural row pitch (no zero pad columns)."""
        lib = load()
        dev = self.device
        keys = keys.to(device=dev, dtype=torch.int32).contiguous()
        B = keys.size(0)
        st = stream_ptr(dev)
        i32 = dict(dtype=torch.int32, device=dev)
        node_off = torch.empty(B + 1, **i32)
        pair_off = torch.empty(B + 1, **i32)
        ws = torch.empty(max(2 * B, 1), **i32)
        check(lib.npi_subgraph_sizes(ptr(self.ptr), ptr(self.nbr), ptr(self.ok), ptr(keys), B, ptr(node_off), ptr(pair_off),
                                     ptr(ws), st), "npi_subgraph_sizes")
        # output sizes are data dependent: one device read per batch, unless the caller knows them
        if n_nodes is not None and n_pairs is not None:
            n, npairs = int(n_nodes), int(n_pairs)
            if n > 2 ** 31 - 1 or 2 * npairs > 2 ** 31 - 1:
                raise OverflowError("InteractionGraph.batch: the batch has more than 2^31 - 1 rows; use fewer keys per call")
            # (the fill kernels check these totals against node_off[-1] / pair_off[-1] themselves: see the status word below)
        else:
            n, npairs = (int(v) for v in torch.stack([node_off[-1], pair_off[-1]]).tolist())
        if n < 0:
            raise OverflowError("InteractionGraph.batch: the batch has more than 2^31 - 1 rows; use fewer keys per call")
        Ff = self.feat.size(1)
        node_id = torch.empty(max(n, 1), **i32)
        bvec = torch.empty(n, dtype=torch.int64, device=dev)
        ei = torch.empty((2, 2 * npairs), dtype=torch.int64, device=dev)
        # an odd feature width (178) is stored with a row pitch of the next multiple of 128 and ZERO pad columns when that
        # costs less than half as much again: x is the [n, 1 + Ff] view, `GraphBatch.pad_base` the buffer; sage_conv then runs its
        # GEMMs on the padded width (matrix-core kernels instead of the guarded ones: functional.linear_fwd)
        Fx = 1 + Ff
        ld = (Fx + 127) // 128 * 128
        if not pad_features or 2 * ld > 3 * Fx:
            ld = Fx
        full = torch.empty((n, ld), dtype=torch.float32, device=dev)
        status = torch.empty(1, **i32)
        check(lib.npi_subgraph_fill(ptr(self.ptr), ptr(self.nbr), ptr(self.ok), ptr(keys), B, ptr(node_off), ptr(pair_off),
                                    ptr(node_id), ptr(bvec), ptr(ei[0]), ptr(ei[1]), n, npairs, ptr(status), st),
              "npi_subgraph_fill")
        check(lib.npi_subgraph_features(ptr(self.feat), self.feat.stride(0), Ff, ptr(node_id), ptr(bvec), ptr(node_off), B, n,
                                        ptr(full), full.stride(0), st), "npi_subgraph_features")
        if n_nodes is not None and n_pairs is not None:
            from . import graph as _graph
            _graph.note_status(status)                         # read now under set_debug, else at the next device read
        # every pair is emitted in both directions (symmetric: graph.CSRGraph.symmetric)
        gb = GraphBatch(full if ld == Fx else full[:, :Fx], ei, bvec, B, symmetric=True, pad_base=None if ld == Fx else full)
        if return_node_id:
            return gb, node_id[:n]
        return gb
