"""npi_gnn_amd -- MI355X-native message-passing engine for NPI-GNN's conv hot path.

Only what the path needs: ``csrc/`` (hand-written gfx950 kernels + the C ABI of
``include/npi_gnn.h``), ``graph`` (device CSR build), ``functional`` / ``nn`` (the PyG
``nn.Conv`` interface the reference calls at ``src/classes.py:48-52,62-70``) and ``dist``
(row sharding over RCCL).  The package directory is ``npi_gnn_amd`` because a hyphen cannot
be imported.
"""
from ._lib import LIB_PATH, NpiError, load  # noqa: F401
from .graph import CSRGraph, GraphBatch, as_graph, set_debug  # noqa: F401
from .functional import GCNNorm, gat_conv, gcn_conv, sage_conv, segsum  # noqa: F401
from .nn import GATConv, GCNConv, SAGEConv  # noqa: F401
from .schedule import Schedule  # noqa: F401
from .graphed import GraphedStack  # noqa: F401

__all__ = ["CSRGraph", "GraphBatch", "as_graph", "set_debug", "GCNNorm", "gat_conv", "gcn_conv", "sage_conv", "segsum",
           "GATConv", "GCNConv", "SAGEConv", "Schedule", "GraphedStack", "NpiError", "load", "LIB_PATH"]
