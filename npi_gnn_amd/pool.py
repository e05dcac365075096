"""Between-layer steps of the reference's ``Net_1`` on MI355X, forward only (inference):
PyG 1.4.2 ``TopKPooling(in_channels, ratio)`` and ``global_max_pool`` / ``global_mean_pool``
(reference ``src/classes.py:49,51,53,63-64,67-68,71-72``).

With these and ``npi_gnn_amd.nn.SAGEConv`` the whole ``Net_1`` forward (``src/classes.py:59-82``) is
GPU-resident for the evaluation / case-study flows (``src/methods.py:87-96``, ``src/test.py``,
``src/case_study*.py``).  Training THROUGH the pooling layer (its backward) is not implemented:
in ``train()`` mode the module raises; in ``eval()`` mode the output is detached.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn
from torch.nn import Parameter

from ._lib import check, load, ptr, require_gpu, stream_ptr


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def graph_ptr(batch: torch.Tensor, num_graphs: Optional[int] = None) -> torch.Tensor:
    """int32 ``[B+1]`` segment starts of a PyG ``batch`` vector (non-decreasing int64)."""
    dev = require_gpu(batch)
    if batch.dtype != torch.int64:
        raise TypeError("batch must be a LongTensor")
    N = batch.numel()
    B = int(num_graphs) if num_graphs is not None else (int(batch[-1].item()) + 1 if N else 0)
    gp = torch.empty(B + 1, dtype=torch.int32, device=dev)
    check(load().npi_graph_bounds(ptr(batch.contiguous()), N, B, ptr(gp), stream_ptr(dev)), "npi_graph_bounds")
    return gp


def topk_pool(x: torch.Tensor, edge_index: torch.Tensor, batch: torch.Tensor, weight: torch.Tensor,
              ratio: float = 0.5, num_graphs: Optional[int] = None):
    """``TopKPooling.forward`` -> ``(x', edge_index', None, batch', perm, score[perm])``."""
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
    check(lib.npi_topk_select(ptr(score), ptr(gp), N, B, float(ratio), ptr(out_ptr), ptr(perm), ptr(remap),
                              ptr(status), st), "npi_topk_select")
    # sizes of the outputs are data dependent: one device read, as PyG's own implementation has
    n_out = int(out_ptr[-1].item()) if B else 0
    if int(status.item()) & 2:
        raise NotImplementedError("TopKPooling: a graph has more than 16384 nodes")
    xo = torch.empty((n_out, F), dtype=torch.float32, device=dev)
    batch_o = torch.empty(n_out, dtype=torch.int64, device=dev)
    score_o = torch.empty(n_out, dtype=torch.float32, device=dev)
    check(lib.npi_topk_gather(ptr(x), x.stride(0), ptr(score), ptr(batch.contiguous()), ptr(perm), ptr(out_ptr), B, F,
                              n_out, ptr(xo), xo.stride(0), ptr(batch_o), ptr(score_o), st), "npi_topk_gather")
    E = edge_index.size(1)
    src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
    out_ei = torch.empty((2, max(E, 1)), dtype=torch.int64, device=dev)
    count = torch.empty(1, **i32)
    ws = torch.empty(int(lib.npi_filter_adj_workspace_elems(E)), **i32)
    check(lib.npi_filter_adj(ptr(src), ptr(dst), E, ptr(remap), ptr(out_ei[0]), ptr(out_ei[1]), ptr(count), ptr(ws), st),
          "npi_filter_adj")
    e_out = int(count.item())
    return xo, out_ei[:, :e_out].contiguous(), None, batch_o, perm[:n_out].long(), score_o


def global_max_mean_pool(x: torch.Tensor, batch: torch.Tensor, num_graphs: Optional[int] = None) -> torch.Tensor:
    """``cat([global_max_pool(x, batch), global_mean_pool(x, batch)], dim=1)`` -> ``[B, 2F]``."""
    dev = require_gpu(x, batch)
    x = _f32(x.detach())
    gp = graph_ptr(batch, num_graphs)
    B, F = gp.numel() - 1, x.size(1)
    out = torch.empty((B, 2 * F), dtype=torch.float32, device=dev)
    check(load().npi_readout_max_mean(ptr(x), x.stride(0), ptr(gp), B, F, ptr(out), stream_ptr(dev)),
          "npi_readout_max_mean")
    return out


def global_max_pool(x, batch, size=None):
    return global_max_mean_pool(x, batch, size)[:, : x.size(1)]


def global_mean_pool(x, batch, size=None):
    return global_max_mean_pool(x, batch, size)[:, x.size(1):]


class TopKPooling(nn.Module):
    """``TopKPooling(in_channels, ratio=0.5)``: parameter ``weight [1, in_channels]`` (PyG 1.4.2 layout,
    so ``pool1.weight`` of the reference checkpoints loads unchanged)."""

    def __init__(self, in_channels: int, ratio: float = 0.5, **kwargs):
        super().__init__()
        self.in_channels, self.ratio = in_channels, ratio
        self.weight = Parameter(torch.empty(1, in_channels))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        bound = 1.0 / math.sqrt(self.in_channels)
        self.weight.data.uniform_(-bound, bound)

    def forward(self, x, edge_index, edge_attr=None, batch=None):
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("TopKPooling on MI355X is forward-only (call model.eval()); the backward "
                                      "through the pooling layer is not implemented")
        if edge_attr is not None:
            raise NotImplementedError("TopKPooling: edge_attr is not used by NPI-GNN")
        if batch is None:
            batch = torch.zeros(x.size(0), dtype=torch.int64, device=x.device)
        return topk_pool(x, edge_index, batch, self.weight, self.ratio)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, ratio={self.ratio})"
