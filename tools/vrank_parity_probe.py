#!/usr/bin/env python3
"""Where do W virtual ranks (exact lock step) and the single-GPU GATConv differ, and which one is closer to fp64?
One layer, no ReLU, on the synthetic bipartite graph at several sizes; errors split by row class (light / hub) and the bias
gradient against its exact fp64 value (column sums of the output gradient).  usage: tools/vrank_parity_probe.py [N E]..."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import dist as ND
from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
from npi_gnn_amd.virtual import LockStep
from npi_gnn_amd.schedule import DEFAULT
SCH = DEFAULT.but(**eval("dict(" + os.environ.get("SCHED", "") + ")"))

dev = torch.device("cuda:0")
F, W = 256, int(os.environ.get("WORLD", "8"))
sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(1_000_000, 20_000_000), (4_000_000, 100_000_000)]
for N, E in sizes:
    t0 = time.time()
    ei = bipartite_edge_index(N, E, seed=2).to(dev)
    g = torch.Generator().manual_seed(23)
    Wm = (torch.randn(F, F, generator=g) / 16).to(dev)
    att = (torch.randn(1, 1, 2 * F, generator=g) * float(os.environ.get("ATT", "0.3"))).to(dev)
    b = (torch.randn(F, generator=g) * float(os.environ.get("BIAS", "0.1"))).to(dev)
    x = torch.randn(N, F, generator=g).to(dev)
    go = torch.randn(N, F, generator=g).to(dev)
    hub = protein_mask(N).to(dev)
    conv = npi.GATConv(F, F).to(dev)
    with torch.no_grad():
        conv.weight.copy_(Wm); conv.att.copy_(att); conv.bias.copy_(b)
    graph = npi.CSRGraph(ei, N)
    xr = x.clone().requires_grad_(True)
    out = conv(xr, graph)
    out.backward(go)
    ref = dict(out=out.detach(), dx=xr.grad, dw=conv.weight.grad.clone(), datt=conv.att.grad.clone(), db=conv.bias.grad.clone())
    db64 = go.double().sum(0)
    # the fp32 noise floor of the layer on this data: the same single-GPU layer with the edge list in another order
    conv2 = npi.GATConv(F, F).to(dev)
    with torch.no_grad():
        conv2.weight.copy_(Wm); conv2.att.copy_(att); conv2.bias.copy_(b)
    xr2 = x.clone().requires_grad_(True)
    o2 = conv2(xr2, npi.CSRGraph(ei[:, torch.randperm(E, device=dev)].contiguous(), N))
    o2.backward(go)
    fl = dict(out=o2.detach(), dx=xr2.grad, dw=conv2.weight.grad, datt=conv2.att.grad)
    del graph, out, xr
    torch.cuda.empty_cache()
    with LockStep(W) as ls:
        sgs = [ND.ShardedGraph(ei, N, r, W, dev, hub_mask=hub, schedule=SCH) for r in range(W)]

        def run(r):
            sg = sgs[r]
            layer = ND.ShardedGATLayer(sg, Wm, att, b)
            xl = x[sg.own].clone().requires_grad_(True)
            o = layer(xl)
            o.backward(go[sg.own])
            return o.detach(), xl.grad, layer.weight.grad, layer.att.grad, layer.bias.grad
        res = ls.run(run)
    part = sgs[0].part

    def rel(a, r_):
        return float((a - r_).abs().max() / r_.abs().max())
    o = torch.empty_like(ref["out"]); dx = torch.empty_like(ref["dx"])
    for r, sg in enumerate(sgs):
        o[sg.own] = res[r][0]; dx[sg.own] = res[r][1]

    light = ~hub
    deg = torch.bincount(ei[1], minlength=N)
    print(f"N={N} E={E}: {time.time() - t0:.0f} s, {ls.passes} passes; max in-degree {int(deg.max())}")
    print("  floor (single GPU, permuted edges): out %.2e dX %.2e dX.l2 %.2e dW %.2e datt %.2e" % (
        rel(fl["out"], ref["out"]), rel(fl["dx"], ref["dx"]), float((fl["dx"] - ref["dx"]).double().norm() / ref["dx"].double().norm()),
        rel(fl["dw"], ref["dw"]), rel(fl["datt"].view(-1), ref["datt"].view(-1))))
    for name, a, r_ in (("out", o, ref["out"]), ("dX", dx, ref["dx"])):
        d = (a - r_).abs()
        worst = int(d.max(1)[0].argmax())
        print(f"  {name}: all {rel(a, r_):.2e}  light rows {float(d[light].max() / r_.abs().max()):.2e}  hub rows {float(d[hub].max() / r_.abs().max()):.2e}"
              f"  L2 {float((a - r_).double().norm() / r_.double().norm()):.2e}  L2 light {float((a - r_)[light].double().norm() / r_[light].double().norm()):.2e}"
              f"  L2 hub {float((a - r_)[hub].double().norm() / r_[hub].double().norm()):.2e}   worst row {worst} (hub {bool(hub[worst])}, in-degree {int(deg[worst])})")
    print(f"  dW {rel(res[0][2], ref['dw']):.2e}  datt {rel(res[0][3].view(-1), ref['datt'].view(-1)):.2e}  db {rel(res[0][4], ref['db']):.2e}"
          f"   db vs fp64: sharded {float((res[0][4].double() - db64).abs().max() / db64.abs().max()):.2e}, single GPU {float((ref['db'].double() - db64).abs().max() / db64.abs().max()):.2e}")
    del sgs, res, ref, o, dx, x, go, ei
    torch.cuda.empty_cache()
