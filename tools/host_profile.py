#!/usr/bin/env python3
"""Where the HOST's time goes in one step of a sharded layer (virtual rank 0 of 8, stand-in collectives): cProfile over the
step loop only, the backward forced onto the calling thread.  usage: tools/host_profile.py [sage|gat] [steps]"""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd import dist as ND
from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
from npi_gnn_amd.virtual import StubCollectives
conv = sys.argv[1] if len(sys.argv) > 1 else "sage"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
N, E, F, W = 1_000_000, 20_000_000, 256, 8
stub = StubCollectives(W, copy_stream=torch.cuda.Stream(device=dev)); stub.__enter__()
ei = bipartite_edge_index(N, E, seed=20260310).to(dev)
g = torch.Generator().manual_seed(3)
Wm = ((torch.rand(F, F, generator=g) * 2 - 1) / F ** 0.5).to(dev)
b = ((torch.rand(F, generator=g) * 2 - 1) / F ** 0.5).to(dev)
sg = ND.ShardedGraph(ei, N, 0, W, dev, hub_mask=protein_mask(N).to(dev))
layer = ND.ShardedSAGELayer(sg, Wm, b) if conv == "sage" else ND.ShardedGATLayer(sg, Wm, (torch.randn(1, 1, 2 * F, generator=g) * 0.1).to(dev), b)
x = torch.randn(sg.n_local, F, device=dev).requires_grad_(True)
go = torch.randn(sg.n_local, F, device=dev)
def step():
    layer.zero_grad(); x.grad = None
    layer(x).backward(go)
for _ in range(10): step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
th = (time.perf_counter() - t0) / steps * 1e3
torch.cuda.synchronize()
print(f"{conv}: host issue {th:.3f} ms/step (backward on the calling thread), wall {(time.perf_counter() - t0) / steps * 1e3:.3f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(steps): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print("per-step figures = totals / %d" % steps)
print(s.getvalue()[:9000])
