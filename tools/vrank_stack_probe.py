#!/usr/bin/env python3
"""3-layer GATConv stack (no ReLU): single GPU vs 8 virtual ranks (lock step), and -- sizes up to 200k nodes -- both against the
fp64 CPU oracle.  usage: tools/vrank_stack_probe.py [N E]..."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import dist as ND
from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
from npi_gnn_amd.virtual import sharded_stack_errors
from oracle import ref_conv as R
from npi_gnn_amd.schedule import DEFAULT
SCH = DEFAULT.but(**eval("dict(" + os.environ.get("SCHED", "") + ")"))

dev = torch.device("cuda:0")
F, W, L = 256, 8, int(os.environ.get("LAYERS", "3"))
sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
for N, E in sizes:
    ei = bipartite_edge_index(N, E, seed=2)
    g = torch.Generator().manual_seed(23)
    ps = [((torch.randn(F, F, generator=g) / 16), (torch.randn(1, 1, 2 * F, generator=g) * 0.3), (torch.randn(F, generator=g) * 0.1)) for _ in range(L)]
    x = torch.randn(N, F, generator=g)
    go = torch.randn(N, F, generator=g)
    hub = protein_mask(N).to(dev)
    eid = ei.to(dev)
    graph = npi.CSRGraph(eid, N)
    pd = [tuple(t.to(dev).requires_grad_(True) for t in p) for p in ps]
    xin = x.to(dev).requires_grad_(True)
    h = xin
    for Wm, a, b in pd:
        h = npi.gat_conv(h, graph, Wm, a, b, heads=1)
    h.backward(go.to(dev))
    ref_out, ref_dx = h.detach(), xin.grad
    ref_grads = [{"weight": Wm.grad, "att": a.grad, "bias": b.grad} for Wm, a, b in pd]
    errs = sharded_stack_errors(W, eid, N, hub, lambda sg: [ND.ShardedGATLayer(sg, Wm.detach(), a.detach(), b.detach()) for Wm, a, b in pd],
                                x.to(dev), go.to(dev), ref_out, ref_dx, ref_grads, dev, relu_between=False, schedule=SCH)
    print(f"N={N} E={E} layers={L} {os.environ.get('SCHED', '')}: sharded vs single GPU:", {k: f"{v:.1e}" for k, v in errs.items() if not k.endswith('.l2') and k != 'lockstep_passes'})
    # the noise floor of the fp32 stack itself: the SAME single-GPU stack on the same graph with the edge list in another order
    # (other summation orders inside every row, the same mathematics)
    perm = torch.randperm(E, generator=g).to(dev)
    graph2 = npi.CSRGraph(eid[:, perm].contiguous(), N)
    pd2 = [tuple(t.to(dev).requires_grad_(True) for t in p) for p in ps]
    xin2 = x.to(dev).requires_grad_(True)
    h2 = xin2
    for Wm, a, b in pd2:
        h2 = npi.gat_conv(h2, graph2, Wm, a, b, heads=1)
    h2.backward(go.to(dev))
    def rel_(a_, r_):
        return float((a_.detach() - r_.detach()).abs().max() / r_.detach().abs().max())
    print("   single GPU, edges permuted, vs single GPU: out %.1e dX %.1e" % (rel_(h2, ref_out), rel_(xin2.grad, ref_dx)),
          {f"layer{k}.{n}": f"{rel_(t.grad, ref_grads[k][n]):.1e}" for k in range(L) for n, t in zip(("weight", "att", "bias"), pd2[k])})
    del graph2, h2, xin2, pd2
    if N <= 200_000:
        p64 = [tuple(t.double().requires_grad_(True) for t in p) for p in ps]
        x64 = x.double().requires_grad_(True)
        h64 = x64
        for Wm, a, b in p64:
            h64 = R.gat_conv(h64, ei, Wm, a, b, heads=1)
        h64.backward(go.double())
        def rel(a_, r_):
            return float((a_.detach().cpu().double() - r_).abs().max() / r_.abs().max())
        print("   single GPU vs fp64 oracle: out %.1e dX %.1e" % (rel(ref_out, h64.detach()), rel(ref_dx, x64.grad)),
              {f"layer{k}.{n}": f"{rel(ref_grads[k][n].reshape(t.grad.shape), t.grad):.1e}" for k in range(L) for n, t in zip(("weight", "att", "bias"), p64[k])})
        # the sharded stack against the oracle: rebuild from the error helper with the oracle as reference
        e2 = sharded_stack_errors(W, eid, N, hub, lambda sg: [ND.ShardedGATLayer(sg, Wm.detach(), a.detach(), b.detach()) for Wm, a, b in pd],
                                  x.to(dev), go.to(dev), h64.detach().float().to(dev), x64.grad.float().to(dev),
                                  [{"weight": Wm.grad.float().to(dev), "att": a.grad.float().to(dev), "bias": b.grad.float().to(dev)} for Wm, a, b in p64],
                                  dev, relu_between=False, schedule=SCH)
        print("   sharded vs fp64 oracle:", {k: f"{v:.1e}" for k, v in e2.items() if not k.endswith('.l2') and k != 'lockstep_passes'})
