"""``nn.Module``s with the PyG 1.4.2 conv signatures, parameter names and initialisers, so that
NPI-GNN's ``Net_1`` (reference ``src/classes.py:45-82``) and its train loop
(``src/train_with_twoDataset.PY:46-57``) run unchanged with::

    from npi_gnn_amd.nn import SAGEConv, GCNConv

``state_dict`` keys are ``weight [in, out]`` (``x @ W`` orientation, not ``nn.Linear``'s) and
``bias [out]``, so the reference's checkpoints (``result/<proj>/model_<k>_fold/<epoch>``, loaded at
``src/test.py:41``) load as they are.  ``forward`` also accepts a prebuilt ``CSRGraph`` in place of
``edge_index`` (static full-batch graphs: sort once), and a ``GraphBatch`` in place of ``x`` -- then it
returns the ``GraphBatch`` with the new features (``conv(gb)``: the CSR is built once per batch and shared
with the pooling layer behind the conv).  Plain tensors carry no hidden state: ``conv(x, edge_index)``
sorts the edge list on every call, as PyG's scatter walks it on every call.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn
from torch.nn import Parameter

from . import functional as F_
from .graph import CSRGraph, GraphBatch, as_graph
from .schedule import DEFAULT, Schedule


def _uniform(size: int, tensor: Optional[torch.Tensor]) -> None:
    """PyG ``inits.uniform``: U(-1/sqrt(size), 1/sqrt(size))."""
    if tensor is not None:
        bound = 1.0 / math.sqrt(size)
        tensor.data.uniform_(-bound, bound)


def _glorot(tensor: Optional[torch.Tensor]) -> None:
    """PyG ``inits.glorot``: U(+-sqrt(6/(fan_in+fan_out))) over the last two dims."""
    if tensor is not None:
        stdv = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-stdv, stdv)


def _only_batch(gb: GraphBatch, edge_index, who: str) -> GraphBatch:
    if edge_index is not None:
        raise TypeError(f"{who}: a GraphBatch carries its own edge_index")
    return gb


class SAGEConv(nn.Module):
    """``SAGEConv(in_channels, out_channels, normalize=False, concat=False, bias=True)`` --
    mean over in-neighbours and the node itself, then ``@ weight + bias``.  Both parameters are
    initialised U(+-1/sqrt(weight.size(0))) as in PyG 1.4.2.  ``concat=True``: no self loop is added, the mean over the
    in-neighbours is concatenated behind the node's own features and ``weight`` is ``[2 in, out]`` -- an option the reference
    never sets, served as a COMPOSED layer (the kernels of the default layer plus torch elementwise ops on the activations in
    its backward: three activation-sized passes more than a fused layer would make), not at the default layer's kernel quality."""

    def __init__(self, in_channels: int, out_channels: int, normalize: bool = False, concat: bool = False,
                 bias: bool = True, schedule: Schedule = DEFAULT, **kwargs):
        super().__init__()
        self.schedule = schedule                  # how the launches are arranged (schedule.Schedule); never what they compute
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.normalize = normalize
        self.concat = concat
        # concat=True (PyG 1.4.2; not what the reference constructs): [x_i | mean_j x_j] @ weight[2 in, out], no self loops added
        self.weight = Parameter(torch.empty(2 * in_channels if concat else in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        _uniform(self.weight.size(0), self.weight)
        _uniform(self.weight.size(0), self.bias)

    def forward(self, x, edge_index=None, edge_weight=None, size=None, *, relu: bool = False):
        """``relu=True`` (an extension of the PyG signature): ``F.relu(conv(x, edge_index))`` with the ReLU applied in the
        projection GEMM's epilogue -- the same values, one launch and one activation-sized tensor fewer."""
        if size is not None:
            raise NotImplementedError("SAGEConv: the bipartite `size` form is not used by NPI-GNN")
        if isinstance(x, GraphBatch):
            gb = _only_batch(x, edge_index, "SAGEConv")
            return gb.with_x(F_.sage_conv(gb.x, gb if self.concat else gb.graph(), self.weight, self.bias, normalize=self.normalize,
                                          edge_weight=edge_weight, relu=relu, pad_base=gb.pad_base, schedule=self.schedule,
                                          concat=self.concat))
        return F_.sage_conv(x, edge_index, self.weight, self.bias, normalize=self.normalize, edge_weight=edge_weight,
                            relu=relu, schedule=self.schedule, concat=self.concat)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"


class GATConv(nn.Module):
    """``GATConv(in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0,
    bias=True)`` (PyG 1.4.2): ``weight [in, heads*out]`` and ``att [1, heads, 2*out]`` glorot,
    ``bias`` zeros (``[heads*out]`` if concat else ``[out]``).  Not used by the reference
    (BASELINE.json configs[4] only).  ``dropout > 0`` in training mode: the composed variant ``functional._GatDropoutFn``."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, dropout: float = 0.0, bias: bool = True, schedule: Schedule = DEFAULT, **kwargs):
        super().__init__()
        if not 0.0 <= dropout < 1.0:
            raise ValueError("GATConv: dropout must be in [0, 1)")
        self.schedule = schedule
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.weight = Parameter(torch.empty(in_channels, heads * out_channels))
        self.att = Parameter(torch.empty(1, heads, 2 * out_channels))
        if bias:
            self.bias = Parameter(torch.empty(heads * out_channels if concat else out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        _glorot(self.weight)
        _glorot(self.att)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, edge_index=None, size=None, *, relu: bool = False, x_scales=None, return_scales: bool = False):
        """``relu=True`` (an extension of the PyG signature, as in ``SAGEConv``): ``F.relu(conv(x, edge_index))`` fused.
        ``x_scales`` / ``return_scales`` (extensions): ``functional.row_scales(x)`` of a feature matrix that does not change between
        steps, or the ``out_scales`` of the layer in front -- ``functional.gat_conv``."""
        if size is not None:
            raise NotImplementedError("GATConv: bipartite `size` is not implemented")
        gb = None
        if isinstance(x, GraphBatch):
            gb = _only_batch(x, edge_index, "GATConv")
            x, edge_index = gb.x, gb.graph()
        keep = None
        if self.dropout > 0 and self.training:
            # F.dropout(alpha, p, training=True): a fresh mask per entry and head.  The fast kernels never hold alpha (both
            # directions recompute it per entry), so a training step with attention dropout takes the composed variant of the
            # layer (functional._GatDropoutFn); in evaluation dropout is the identity and the layer runs as usual
            edge_index = as_graph(edge_index, x.size(0))
            keep = F_.gat_dropout_keep(edge_index, self.heads, self.dropout)
        out = F_.gat_conv(x, edge_index, self.weight, self.att, self.bias, self.heads, self.concat,
                          self.negative_slope, relu=relu, schedule=self.schedule, keep=keep, x_scales=x_scales,
                          return_scales=return_scales)
        if gb is not None:
            return (gb.with_x(out[0]), out[1]) if return_scales else gb.with_x(out)
        return out

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, heads={self.heads})"


class GCNConv(nn.Module):
    """``GCNConv(in_channels, out_channels, improved=False, cached=False, bias=True,
    normalize=True)``; ``weight`` glorot, ``bias`` zeros (PyG 1.4.2)."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False, cached: bool = False,
                 bias: bool = True, normalize: bool = True, schedule: Schedule = DEFAULT, **kwargs):
        super().__init__()
        self.schedule = schedule
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.improved = improved
        self.cached = cached
        self.normalize = normalize
        self.weight = Parameter(torch.empty(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        _glorot(self.weight)
        if self.bias is not None:
            self.bias.data.zero_()
        self.cached_result = None
        self.cached_num_edges = None

    def forward(self, x, edge_index=None, edge_weight=None):
        if isinstance(x, GraphBatch):
            gb = _only_batch(x, edge_index, "GCNConv")
            return gb.with_x(self.forward(gb.x, gb.graph() if self.normalize else gb.edge_index, edge_weight))
        norm = None
        if self.cached and self.cached_result is not None:
            E = edge_index.num_edges if isinstance(edge_index, CSRGraph) else edge_index.size(1)
            if E != self.cached_num_edges:
                raise RuntimeError(
                    f"Cached {self.cached_num_edges} number of edges, but found {E}. Please disable "
                    "the caching behavior of this layer by removing the `cached=True` argument in its "
                    "constructor.")
            norm = self.cached_result
        if norm is None:
            if self.normalize:
                norm = F_.GCNNorm(as_graph(edge_index, x.size(0)), edge_weight, self.improved)
            else:                                      # PyG 1.4.2: norm = edge_weight on the edge list as it is (no self loops)
                norm = F_.PlainWeights(edge_index, x.size(0), edge_weight)
            if self.cached:
                self.cached_result = norm
                self.cached_num_edges = norm.graph.num_edges
        return F_.gcn_conv(x, None, self.weight, self.bias, norm=norm, schedule=self.schedule)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"
