"""CPU oracle for the NPI-GNN conv hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker.  ``npi_gnn_amd`` never imports it.

What it restates
----------------
The arithmetic of the reference's hot path is NOT in ``/root/reference``: ``Net_1``
(reference ``src/classes.py:45-82``) calls ``torch_geometric.nn.SAGEConv`` (constructed at
``src/classes.py:48,50,52``, called at ``:62,66,70``), and the gather / scatter / matmul run
inside the un-vendored pip dependency **torch-geometric == 1.4.2** (pinned at reference
``README.md:11``; pytorch 1.4.0 at ``README.md:9``).  This file restates the published
PyG 1.4.2 algorithm in plain torch CPU ops, dispatching the same op sequence PyG does
(``index_select`` -> scatter (``index_add_``) -> divide -> ``matmul``), so that it doubles
as the "reference CPU path" timed by ``bench.py``.

Pinning status
--------------
* ``sage_conv`` / ``topk_pool`` / ``readout`` / ``net1_forward``: **pinned** by the
  reference's own checkpoints + logs (``oracle/kat.py``; confusion matrices of
  ``result/<proj>/log_<k>.txt`` reproduced exactly, per-sample probabilities of
  ``data/case_study/*/logs/case_predict_negative.txt`` to <= 1e-5).
* ``gcn_conv`` / ``gat_conv`` / the ``edge_weight`` paths / backward: **parity unpinned** --
  no reference artifact exercises them (``GCNConv`` is imported but never constructed,
  ``src/classes.py:1``; ``GATConv`` appears nowhere).  They restate the PyG 1.4.2
  formulas from the published source; gradients are pinned by ``torch.autograd.gradcheck``
  in fp64 only.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# torch_geometric.utils (1.4.2) restated
# --------------------------------------------------------------------------------------
def add_remaining_self_loops(edge_index: Tensor, edge_weight: Optional[Tensor] = None,
                             fill_value: float = 1.0, num_nodes: Optional[int] = None
                             ) -> Tuple[Tensor, Optional[Tensor]]:
    """PyG 1.4.2 ``utils.loop.add_remaining_self_loops``: drop existing self loops, append
    ``(i, i)`` for every node AT THE END; an existing self-loop's weight survives as the
    appended loop's weight.  Used by SAGEConv.forward and GCNConv.norm."""
    N = int(num_nodes) if num_nodes is not None else int(edge_index.max()) + 1
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop_index = torch.arange(N, dtype=edge_index.dtype, device=edge_index.device)
    if edge_weight is not None:
        assert edge_weight.numel() == edge_index.size(1)
        inv_mask = ~mask
        loop_weight = torch.full((N,), fill_value, dtype=edge_weight.dtype,
                                 device=edge_weight.device)
        remaining = edge_weight[inv_mask]
        if remaining.numel() > 0:
            loop_weight[row[inv_mask]] = remaining
        edge_weight = torch.cat([edge_weight[mask], loop_weight], dim=0)
    edge_index = torch.cat([edge_index[:, mask], loop_index.unsqueeze(0).repeat(2, 1)], dim=1)
    return edge_index, edge_weight


def remove_self_loops(edge_index: Tensor) -> Tensor:
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask]


def add_self_loops(edge_index: Tensor, num_nodes: int) -> Tensor:
    loop_index = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loop_index.unsqueeze(0).repeat(2, 1)], dim=1)


def scatter_add(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """torch_scatter.scatter_add along dim 0 (CPU semantics: serial accumulation in edge order)."""
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def scatter_mean(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """torch_scatter.scatter_mean: sum / max(count, 1)."""
    out = scatter_add(src, index, dim_size)
    cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
    cnt.index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype, device=src.device))
    cnt = cnt.clamp_(min=1)
    return out / cnt.view((-1,) + (1,) * (src.dim() - 1))


def scatter_max(src: Tensor, index: Tensor, dim_size: int, fill: float = -1e38) -> Tensor:
    out = torch.full((dim_size,) + tuple(src.shape[1:]), fill, dtype=src.dtype, device=src.device)
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    return out.scatter_reduce(0, idx, src, reduce="amax", include_self=True)


def segment_softmax(src: Tensor, index: Tensor, num_nodes: int) -> Tensor:
    """PyG 1.4.2 ``utils.softmax``: exp(src - max_seg) / (sum_seg + 1e-16)."""
    mx = scatter_max(src, index, num_nodes)
    out = (src - mx.index_select(0, index)).exp()
    den = scatter_add(out, index, num_nodes).index_select(0, index) + 1e-16
    return out / den


# --------------------------------------------------------------------------------------
# conv layers (functional; parameters passed in)
# --------------------------------------------------------------------------------------
def sage_aggregate(x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None) -> Tensor:
    """Rows a2-a4 of SURVEY.md section 8(a): self-loop insert -> gather -> scatter-mean.
    Mean over in-neighbours U {i}, aggregated at the TARGET ``edge_index[1]`` of messages
    gathered at the SOURCE ``edge_index[0]`` (flow source_to_target)."""
    N = x.size(0)
    ei, ew = add_remaining_self_loops(edge_index, edge_weight, 1.0, N)
    x_j = x.index_select(0, ei[0])                       # materialised [E+N, F] as PyG does
    if ew is not None:
        x_j = ew.view(-1, 1) * x_j
    return scatter_mean(x_j, ei[1], N)


def sage_conv(x: Tensor, edge_index: Tensor, weight: Tensor, bias: Optional[Tensor] = None,
              edge_weight: Optional[Tensor] = None, normalize: bool = False) -> Tensor:
    """PyG 1.4.2 ``SAGEConv(in, out, normalize=False, concat=False, bias=True).forward``.
    Call sites: reference ``src/classes.py:62,66,70``.  ``weight`` is ``[in, out]``
    (``x @ W`` orientation -- the checkpoints under ``result/`` prove it)."""
    out = torch.matmul(sage_aggregate(x, edge_index, edge_weight), weight)
    if bias is not None:
        out = out + bias
    if normalize:
        out = F.normalize(out, p=2.0, dim=-1)
    return out


def sage_conv_concat(x: Tensor, edge_index: Tensor, weight: Tensor, bias: Optional[Tensor] = None,
                     edge_weight: Optional[Tensor] = None, normalize: bool = False) -> Tensor:
    """PyG 1.4.2 ``SAGEConv(in, out, concat=True).forward``: NO self loops are added (``add_remaining_self_loops`` runs only
    ``if not self.concat``; the edge list is used as it is, existing self loops included), the mean over the in-neighbours
    (``scatter_mean``: 0 for a node without one) is CONCATENATED behind the node's own features, and ``weight`` is
    ``[2 * in, out]``: ``out = cat([x, mean_j x_j]) @ W + b``.  Parity unpinned: the reference constructs its layers with the
    default ``concat=False`` (``src/classes.py:48-52``)."""
    N = x.size(0)
    x_j = x.index_select(0, edge_index[0])
    if edge_weight is not None:
        x_j = edge_weight.view(-1, 1) * x_j
    aggr = scatter_mean(x_j, edge_index[1], N)
    out = torch.matmul(torch.cat([x, aggr], dim=-1), weight)
    if bias is not None:
        out = out + bias
    if normalize:
        out = F.normalize(out, p=2.0, dim=-1)
    return out


def gcn_norm(edge_index: Tensor, num_nodes: int, edge_weight: Optional[Tensor] = None,
             improved: bool = False, dtype=torch.float32) -> Tuple[Tensor, Tensor]:
    """PyG 1.4.2 ``GCNConv.norm`` -- NOTE the degree is scatter-added over ``row`` (= source)."""
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    fill = 2.0 if improved else 1.0
    ei, ew = add_remaining_self_loops(edge_index, edge_weight, fill, num_nodes)
    row, col = ei[0], ei[1]
    deg = scatter_add(ew, row, num_nodes)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    return ei, dis[row] * ew * dis[col]


def gcn_conv(x: Tensor, edge_index: Tensor, weight: Tensor, bias: Optional[Tensor] = None,
             edge_weight: Optional[Tensor] = None, improved: bool = False, normalize: bool = True) -> Tensor:
    """PyG 1.4.2 ``GCNConv.forward``: project FIRST, then ``out[i] = sum_e norm_e * (xW)[src e] + b``.  ``normalize=False``:
    ``norm = edge_weight`` on the edge list AS IT IS (no self loop added; ``message`` multiplies only when a norm is given).
    Parity unpinned (module docstring)."""
    N = x.size(0)
    xw = torch.matmul(x, weight)
    if normalize:
        ei, norm = gcn_norm(edge_index, N, edge_weight, improved, x.dtype)
    else:
        ei, norm = edge_index, (edge_weight if edge_weight is not None else torch.ones(edge_index.size(1), dtype=x.dtype))
    msg = norm.view(-1, 1) * xw.index_select(0, ei[0])
    out = scatter_add(msg, ei[1], N)
    if bias is not None:
        out = out + bias
    return out


def gat_conv(x: Tensor, edge_index: Tensor, weight: Tensor, att: Tensor,
             bias: Optional[Tensor] = None, heads: int = 1, concat: bool = True,
             negative_slope: float = 0.2, keep_scale: Optional[Tensor] = None) -> Tensor:
    """PyG 1.4.2 ``GATConv.forward``.  ``att`` is ``[1, H, 2C]``: the first C
    entries multiply the TARGET features x_i, the last C the SOURCE features x_j.
    ``keep_scale`` ``[E', H]`` (E' = the columns of the self-loop-augmented edge list, loops last): what
    ``F.dropout(alpha, p, training=True)`` multiplies alpha by -- 0 for a dropped weight, 1 / (1 - p) for a kept
    one -- made explicit so that a test can hand the same draw to both sides; None = dropout 0 / evaluation.
    Parity unpinned (module docstring)."""
    N = x.size(0)
    H = heads
    C = weight.size(1) // H
    ei = add_self_loops(remove_self_loops(edge_index), N)
    h = torch.matmul(x, weight)
    x_j = h.index_select(0, ei[0]).view(-1, H, C)
    x_i = h.index_select(0, ei[1]).view(-1, H, C)
    alpha = (torch.cat([x_i, x_j], dim=-1) * att).sum(dim=-1)          # [E', H]
    alpha = F.leaky_relu(alpha, negative_slope)
    alpha = segment_softmax(alpha, ei[1], N)
    if keep_scale is not None:
        alpha = alpha * keep_scale
    msg = x_j * alpha.view(-1, H, 1)
    out = scatter_add(msg, ei[1], N)                                    # [N, H, C]
    out = out.reshape(N, H * C) if concat else out.mean(dim=1)
    if bias is not None:
        out = out + bias
    return out


def keep_scale_from_entries(edge_index: Tensor, num_nodes: int, eid: Tensor, rowidx: Tensor, keep: Tensor) -> Tensor:
    """Carry a dropout draw made per CSR ENTRY (the build's by-target order: ``eid`` = original column of ``edge_index`` or -1 for
    the appended self loop, ``rowidx`` = the entry's target, ``keep [nnz, H]``; all on the CPU, padding already cut off) over to
    the order ``gat_conv`` walks: the columns of ``edge_index`` that are not self loops, then the N appended loops."""
    eid, rowidx = eid.long(), rowidx.long()
    kept = edge_index[0] != edge_index[1]
    pos, n_kept = torch.cumsum(kept, 0) - 1, int(kept.sum())
    idx = torch.where(eid >= 0, pos[eid.clamp(min=0)], n_kept + rowidx)
    if idx.numel() != n_kept + num_nodes or idx.unique().numel() != idx.numel():
        raise ValueError("keep_scale_from_entries: the entries are not the self-loop-augmented edge list, one entry per column")
    ks = torch.empty(n_kept + num_nodes, keep.size(1), dtype=torch.float64)
    ks[idx] = keep.double()
    return ks


# --------------------------------------------------------------------------------------
# the rest of Net_1 (needed only so the KAT can run end to end)
# --------------------------------------------------------------------------------------
def topk_pool(x: Tensor, edge_index: Tensor, batch: Tensor, w: Tensor, ratio: float = 0.5):
    """PyG 1.4.2 ``TopKPooling(C, ratio)``; call sites reference ``src/classes.py:63,67,71``.
    score = tanh(x.w/||w||); per graph keep ceil(ratio*n) highest; gate x by the score;
    ``filter_adj`` on the ORIGINAL edge_index (no conv self loops)."""
    score = torch.tanh((x * w.view(1, -1)).sum(dim=-1) / w.norm(p=2))
    B = int(batch.max()) + 1 if batch.numel() else 0
    n_per = torch.bincount(batch, minlength=B)
    k_per = torch.ceil(ratio * n_per.to(torch.float64)).to(torch.long)
    # descending by score within graph, graphs in order: sort by (batch asc, score desc)
    order = torch.argsort(score, descending=True, stable=True)
    order = order[torch.argsort(batch[order], stable=True)]
    starts = torch.cumsum(n_per, 0) - n_per
    rank = torch.arange(x.size(0)) - starts[batch[order]]
    perm = order[rank < k_per[batch[order]]]
    xo = x[perm] * score[perm].view(-1, 1)
    bo = batch[perm]
    remap = torch.full((x.size(0),), -1, dtype=torch.long)
    remap[perm] = torch.arange(perm.numel())
    r, c = remap[edge_index[0]], remap[edge_index[1]]
    keep = (r >= 0) & (c >= 0)
    return xo, torch.stack([r[keep], c[keep]]), bo, perm, score[perm]


def readout(x: Tensor, batch: Tensor, num_graphs: int) -> Tensor:
    """cat[global_max_pool, global_mean_pool] (reference ``src/classes.py:64,68,72``)."""
    mx = scatter_max(x, batch, num_graphs)
    mean = scatter_mean(x, batch, num_graphs)
    return torch.cat([mx, mean], dim=1)


def net1_forward(sd: dict, x: Tensor, edge_index: Tensor, batch: Tensor, num_graphs: int,
                 conv=sage_conv, return_layers: bool = False):
    """Eval-mode ``Net_1.forward`` (reference ``src/classes.py:59-82``) from a reference
    ``state_dict`` (keys ``convK.weight [in,out]``, ``convK.bias``, ``poolK.weight [1,128]``,
    ``linK.weight [out,in]``).  ``conv`` can be swapped for the HIP-backed functional conv so
    the same KAT exercises the product path."""
    layers = []
    acc = None
    for k in (1, 2, 3):
        h = conv(x, edge_index, sd[f"conv{k}.weight"], sd[f"conv{k}.bias"])
        layers.append(h)
        x = F.relu(h)
        x, edge_index, batch, _, _ = topk_pool(x, edge_index, batch, sd[f"pool{k}.weight"], 0.5)
        r = readout(x, batch, num_graphs)
        acc = r if acc is None else acc + r
    z = F.relu(F.linear(acc, sd["lin1.weight"], sd["lin1.bias"]))
    z = F.relu(F.linear(z, sd["lin2.weight"], sd["lin2.bias"]))
    z = F.linear(z, sd["lin3.weight"], sd["lin3.bias"])
    logp = F.log_softmax(z, dim=-1)
    return (logp, layers) if return_layers else logp


def metrics_from_confusion(TP: int, FN: int, TN: int, FP: int):
    """Formulas of reference ``src/methods.py:107-126``."""
    tot = TP + TN + FP + FN
    acc = (TP + TN) / tot if tot else 0
    pre = TP / (TP + FP) if (TP + FP) else 0
    sen = TP / (TP + FN) if (TP + FN) else 0
    den = math.sqrt((TP + FP) * (TP + FN) * (TN + FP) * (TN + FN))
    mcc = (TP * TN - FP * FN) / den if den else 0
    spe = TN / (FP + TN) if (FP + TN) else 0
    return acc, pre, sen, spe, mcc


# --------------------------------------------------------------------------------------
# one conv LAYER fwd+bwd the way the reference's train loop drives it -- the CPU baseline
# --------------------------------------------------------------------------------------
def sage_layer_fwd_bwd(x: Tensor, edge_index: Tensor, weight: Tensor, bias: Tensor,
                       grad_out: Tensor):
    """conv -> autograd backward (reference ``src/train_with_twoDataset.PY:52-54``) for one
    layer; returns (out, dX, dW, db)."""
    x = x.detach().requires_grad_(True)
    w = weight.detach().requires_grad_(True)
    b = bias.detach().requires_grad_(True)
    out = sage_conv(x, edge_index, w, b)
    out.backward(grad_out)
    return out.detach(), x.grad, w.grad, b.grad
