"""How a layer ARRANGES its launches -- never what they compute.

Every field selects between two arrangements of the same arithmetic (a second HIP stream or one, an epilogue or a separate
pass, one layout of the hub rows or the other).  A ``Schedule`` is an explicit argument of the layers (``nn.SAGEConv(...,
schedule=)``, ``functional.sage_conv(..., schedule=)``, ``dist.ShardedGraph(..., schedule=)``); nothing in the package reads
a process-wide switch or an environment variable to pick a path.  Each alternative is exercised by a ``-m gpu`` test
(``tests/test_gpu_schedule.py``) against the default.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Optional


@dataclass(frozen=True)
class Schedule:
    # single-GPU layers -----------------------------------------------------------------------------------------------------
    #: dW (MFMA-bound, launched as about one workgroup per CU) on the launch stream and the HBM-bound backward aggregation on a
    #: second HIP stream, so that the two share every CU instead of queueing (C4: 8.0 -> 7.1 ms per step); also the small
    #: HBM-bound passes of the GATConv backward under its GEMMs.  Only from ``overlap_min_rows`` rows on.
    overlap_streams: bool = True
    overlap_min_rows: int = 100_000
    #: the projection GEMMs of the single-GPU layers on the exact-f32 MFMA kernels (``NPI_GEMM_EXACT_F32``: 1/16 of the 16-bit
    #: matrix rate) instead of the f32-accurate splits: for a model whose activations may hold ``Inf`` -- an operand split turns
    #: ``Inf`` into ``NaN`` where ``torch.matmul`` keeps ``Inf``.  Takes the per-op path without store-epilogue
    #: fusions.  (Round 5 had a module global for this.)
    gemm_exact_f32: bool = False
    #: rows from which the f32 projection GEMMs of the single-GPU layers run on two fp16 pieces per operand (NPI_GEMM_SPLIT_F16X2:
    #: three matrix products per tile pair instead of six, the same f32-rounding-level error -- 256 features, where the launch
    #: that writes the left operand writes its row scales too); None: never.  Below, the GEMMs are launch-bound
    f16x2_min_rows: Optional[int] = 100_000
    #: large SAGEConv / GCNConv layers whose forward projection runs on fp16 x 2 (256 features): the backward as
    #: ``dX = (A^T dOut) W^T`` -- the transposed aggregation on dOut itself, writing the row scales of its output, then the data GEMM
    #: on fp16 x 2 (three matrix products instead of six) -- instead of ``A^T (dOut W^T)`` with the GEMM on bf16 x 3
    aggregate_first_backward: bool = True
    #: one-head GATConv: the attention terms of dX in the store epilogue of its GEMM (rank 2) instead of a read-modify-write pass
    #: over d hfeat; from ``gat_rank2_min_rows`` rows on (below, the three [2, .] products cost more than the pass they replace)
    gat_rank2_epilogue: bool = True
    gat_rank2_min_rows: int = 100_000
    #: one-head GATConv forward: a_dst / a_src of every node from the accumulators of h = x W in the GEMM's store epilogue
    #: instead of a pass over h (shapes one column tile covers: 128 or 256 output channels)
    gat_scores_epilogue: bool = True
    #: one-head GATConv forward (<= 256 channels): the softmax statistics INSIDE the aggregation launch (every item computes its
    #: entries' scores; parts of cut rows are merged with rescaling: ``npi_gat_aggregate_fused``) instead of a statistics pass
    #: that leaves every entry's score for the aggregation to read back
    gat_fused_stats: bool = True
    #: one-head GATConv backward (rank-2 path, two streams): the by-source row sum of dz on the side stream, in front of the
    #: by-target one and beside the dW GEMM, instead of on the launch stream in front of dW
    gat_src_rowsum_beside_dw: bool = False
    #: one-head GATConv backward: the by-source row sums of dz (g_src) taken INSIDE the fused by-source pass by the lanes that
    #: compute dz (a segmented scan per 64 entries; rows cut by an item boundary through segscan.hip's chain kernel) instead of a
    #: pass of their own over dz (round 6: 0.09 ms of the C4 layer, 0.35 ms per layer at the C5 size)
    gat_src_rowsum_fused: bool = True
    # sharded layers (dist.py) --------------------------------------------------------------------------------------------------
    #: cuts without hub-hub edges: the reduce-scatter delivers the COMPLETE hub rows straight into the output (no merge pass)
    direct_hub_rows: bool = True
    #: the partial (side B) aggregation on its own HIP stream, beside the all-gather and the full side
    partial_stream: bool = True
    #: the light rows' projection launched before the reduce-scattered hub rows have arrived
    split_projection: bool = True
    #: backward of the sharded SAGE / GCN layers (with ``split_projection``): the hub rows of dAgg are projected first and their
    #: all-gather is issued before the light rows' GEMM.  OFF by default: on one GPU it costs a rank 3.5 % (an extra one-round GEMM;
    #: the stand-in copy shares HBM with the light rows' GEMM), and on a node the collective's workgroups find no registers on a
    #: CU while the persistent GEMM workgroup (8 x 230 VGPRs) is resident, so the head start is not theirs to use (EXPERIMENTS A9)
    early_hub_gather: bool = False
    #: CUs (a multiple of 8) the projection GEMMs of the sharded SAGE / GCN layers leave free when W > 1: a persistent GEMM whose
    #: workgroup finds its CU held by a collective's kernel starts late with its full share of the tiles (a stand-in that holds 16
    #: CUs: 0.107 -> 0.15 ms for a rank's 125 k rows; EXPERIMENTS A16).  0 until a node trace shows the collision: on one GPU,
    #: where nothing holds a CU, 16 cost 16 / 256 more tiles per workgroup
    gemm_reserve_cus: int = 0
    #: the light rows' projection of ``split_projection`` runs beside the reduce-scatter BY DESIGN, so it leaves this many CUs to
    #: the collective's kernel whatever ``gemm_reserve_cus`` says (at W = 8 and C4 its 879 tiles are four rounds on 240 workgroups
    #: as on 256: free on one GPU)
    split_projection_reserve_cus: int = 16
    #: GATConv on the direct layout with the fused packed backward (needs ``direct_hub_rows``)
    gat_direct: bool = True

    def __post_init__(self):
        # the flag carries n / 8 in eight bits and the launch clamps at 128: say so here instead of rounding silently
        for name in ("gemm_reserve_cus", "split_projection_reserve_cus"):
            n = getattr(self, name)
            if not isinstance(n, int) or n < 0 or n > 128 or n % 8:
                raise ValueError(f"Schedule.{name} must be a multiple of 8 in [0, 128] (one CU per XCD at a time), got {n!r}")

    def but(self, **changes) -> "Schedule":
        return replace(self, **changes)


DEFAULT = Schedule()
#: the round-2 arrangement of the sharded layers: classic hub layout, one extra stream, one GEMM per direction, no store epilogue.
#: ``bench.py --gpus N`` falls back on it when its pre-flight step fails on a configuration no box has run yet.
CONSERVATIVE = Schedule(direct_hub_rows=False, partial_stream=False, split_projection=False, gat_direct=False,
                        gat_rank2_epilogue=False)
