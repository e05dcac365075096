#!/usr/bin/env python3
"""One Net_1 training step on the first batch of 200 enclosing subgraphs of NPInter2 fold 0 (bench.py configs.R_net1_step), eager and
replayed from a HIP graph; under rocprofv3 --kernel-trace the kernels of a replayed step.  usage: tools/net1_step_probe.py [n]"""
import os, sys, time
import torch
import torch.nn.functional as F_
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd import net1
from npi_gnn_amd.subgraph import InteractionGraph
dev = torch.device("cuda:0")
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
fz = torch.load(os.path.join(G, "npinter2_folds.pt"), map_location="cpu", weights_only=False)
fb = fz["fold0"]
pairs, label, Nn = fz["pairs"].long(), fz["label"].long(), fz["num_nodes"]
test = torch.cat([fb["test_pos"], fb["test_neg"]]).long()
usable = ~torch.isin(pairs[:, 0] * Nn + pairs[:, 1], test[:, 0] * Nn + test[:, 1])
feat = torch.cat([fb["node2vec"], fz["kmer"]], dim=1)
ig = InteractionGraph(pairs.to(dev), usable.to(dev), feat.to(dev), num_nodes=Nn)
keys, yk = pairs[usable][:800].to(dev), label[usable][:800].to(dev)
loader = net1.KeyLoader(ig, keys, yk, 200)
torch.manual_seed(0)
model = net1.Net_1(feat.size(1) + 1, 2).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-3, device=dev), weight_decay=1e-3, capturable=True, fused=True)
ep = net1.GraphedEpoch(model, loader, opt, dev)
ep(); ep()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = ep.graphs[0]
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): g.replay()
torch.cuda.synchronize()
print(f"Net_1 step replayed: {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
