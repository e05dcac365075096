#!/bin/bash
python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
import npi_gnn_amd as npi
from npi_gnn_amd import dist as ND
from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
from npi_gnn_amd.virtual import gat_stack_reference, sharded_stack_errors, stack_distance
dev = torch.device("cuda:0")
N5, E5, F = 4_000_000, 100_000_000, 256
ei = bipartite_edge_index(N5, E5, seed=2).to(dev)
torch.manual_seed(11)
x = torch.randn(N5, F, device=dev)
g = torch.Generator().manual_seed(23)
params = [((torch.randn(F, F, generator=g) / 16).to(dev), (torch.randn(1, 1, 2 * F, generator=g) * 0.3).to(dev), (torch.randn(F, generator=g) * 0.1).to(dev)) for _ in range(3)]
go = torch.randn(N5, F, generator=g).to(dev)
hub = protein_mask(N5).to(dev)
ref = gat_stack_reference(ei, N5, params, x, go, relu=True)
floors = [stack_distance(gat_stack_reference(ei, N5, params, x, go, relu=True, permute_seed=s), ref) for s in (5, 6, 7)]
errs = sharded_stack_errors(8, ei, N5, hub, lambda sg: [ND.ShardedGATLayer(sg, W, a, b) for W, a, b in params], x, go, *ref, dev, relu_between=True)
errs.pop("lockstep_passes")
for k in sorted(errs):
    print(f"{k:18s} sharded {errs[k]:.2e}   floors " + " ".join(f"{f[k]:.2e}" for f in floors))
PY
