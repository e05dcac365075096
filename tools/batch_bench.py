#!/usr/bin/env python3
"""The reference's REAL call pattern (SURVEY.md 8(a) "R"): one PyG batch of 200 one-hop enclosing
subgraphs (double stars: every edge touches local node 0 or 1; mean 212 nodes / 422 directed edges per
subgraph, F = 178 -> 128), edge_index changing per layer => CSR rebuilt per conv call.
Times conv1 fwd+bwd and the whole Net_1 inference on the GPU and the same through the CPU oracle."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd import pool as NP  # noqa: E402
from oracle import ref_conv as R  # noqa: E402


def make_batch(n_graphs=200, seed=0):
    g = torch.Generator().manual_seed(seed)
    xs, eis, bs, off = [], [], [], 0
    for k in range(n_graphs):
        n = int(torch.empty(1).log_normal_(4.3, 1.1, generator=g).clamp(3, 924).item())   # median ~74, mean ~200
        a = int(torch.randint(1, n - 1, (1,), generator=g).item())      # neighbours of node 0: 2..a+1 ; rest -> node 1
        src = torch.cat([torch.zeros(1, dtype=torch.long), torch.zeros(a - 1 if a > 1 else 0, dtype=torch.long),
                         torch.ones(n - 1 - a, dtype=torch.long)])
        dst = torch.cat([torch.ones(1, dtype=torch.long), torch.arange(2, a + 1), torch.arange(a + 1, n)])
        ei = torch.stack([torch.cat([src, dst]), torch.cat([dst, src])]) + off
        xs.append(torch.randn(n, 178, generator=g))
        eis.append(ei)
        bs.append(torch.full((n,), k, dtype=torch.long))
        off += n
    return torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)


def timeit(fn, n=20, sync=None):
    for _ in range(3):
        fn()
    if sync:
        sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    if sync:
        sync()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    train_only = "--train-only" in sys.argv       # for rocprofv3: only the training step, 30 + 3 iterations
    dev = torch.device("cuda:0")
    x, ei, batch = make_batch()
    N, E = x.size(0), ei.size(1)
    print(f"batch: 200 graphs, N={N}, E={E}, F=178")
    torch.manual_seed(0)
    conv = npi.SAGEConv(178, 128)
    W, b = conv.weight.detach().clone(), conv.bias.detach().clone()
    go = torch.randn(N, 128)
    conv = conv.to(dev)
    xd, eid, god = x.to(dev).requires_grad_(True), ei.to(dev), go.to(dev)

    def gpu_layer():
        xd.grad = None
        conv.zero_grad()
        conv(xd, eid).backward(god)
    if not train_only:
        small_parts(gpu_layer, x, ei, W, b, go, E, conv, xd, eid, god, N)
    net1_parts(dev, x, ei, batch, xd, eid, train_only)


def small_parts(gpu_layer, x, ei, W, b, go, E, conv, xd, eid, god, N):
    t_gpu = timeit(gpu_layer, 50, torch.cuda.synchronize)
    t_cpu = timeit(lambda: R.sage_layer_fwd_bwd(x, ei, W, b, go), 5)
    print(f"conv1 fwd+bwd incl. CSR build: GPU {t_gpu:.3f} ms ({E / t_gpu / 1e3:.1f} M edges/s)   "
          f"CPU oracle {t_cpu:.1f} ms ({E / t_cpu / 1e3:.2f} M edges/s, {torch.get_num_threads()} threads)")
    graph = npi.CSRGraph(eid, N)
    _ = graph.by_src

    def gpu_layer_prebuilt():
        xd.grad = None
        conv.zero_grad()
        conv(xd, graph).backward(god)
    print(f"conv1 fwd+bwd, prebuilt CSR:   GPU {timeit(gpu_layer_prebuilt, 50, torch.cuda.synchronize):.3f} ms")
    print(f"CSRGraph build (both sides):   GPU {timeit(lambda: npi.CSRGraph(eid, N).by_src, 50, torch.cuda.synchronize):.3f} ms")


def net1_parts(dev, x, ei, batch, xd, eid, train_only):
    # whole Net_1 inference
    sd = {f"conv{k}.weight": torch.randn(178 if k == 1 else 128, 128) * 0.1 for k in (1, 2, 3)}
    sd.update({f"conv{k}.bias": torch.zeros(128) for k in (1, 2, 3)})
    sd.update({f"pool{k}.weight": torch.randn(1, 128) for k in (1, 2, 3)})
    sd.update({"lin1.weight": torch.randn(128, 256) * 0.1, "lin1.bias": torch.zeros(128), "lin2.weight": torch.randn(64, 128) * 0.1,
               "lin2.bias": torch.zeros(64), "lin3.weight": torch.randn(2, 64) * 0.1, "lin3.bias": torch.zeros(2)})
    sdd = {k: v.to(dev) for k, v in sd.items()}
    bd = batch.to(dev)

    def gpu_net1():
        with torch.no_grad():
            h, e, bb, acc = xd.detach(), eid, bd, None
            for k in (1, 2, 3):
                h = F.relu(npi.sage_conv(h, e, sdd[f"conv{k}.weight"], sdd[f"conv{k}.bias"]))
                h, e, _, bb, _, _ = NP.topk_pool(h, e, bb, sdd[f"pool{k}.weight"], 0.5, num_graphs=200)
                r = NP.global_max_mean_pool(h, bb, 200)
                acc = r if acc is None else acc + r
            z = F.relu(F.linear(acc, sdd["lin1.weight"], sdd["lin1.bias"]))
            z = F.relu(F.linear(z, sdd["lin2.weight"], sdd["lin2.bias"]))
            return F.log_softmax(F.linear(z, sdd["lin3.weight"], sdd["lin3.bias"]), -1)
    if not train_only:
        ref = R.net1_forward(sd, x, ei, batch, 200)
        out = gpu_net1().cpu()
        print("Net_1 forward GPU vs oracle max |d logp| =", float((out - ref).abs().max()))
        t_gpu = timeit(gpu_net1, 30, torch.cuda.synchronize)
        t_cpu = timeit(lambda: R.net1_forward(sd, x, ei, batch, 200), 5)
        print(f"Net_1 inference per batch: GPU {t_gpu:.3f} ms   CPU oracle {t_cpu:.1f} ms")

    # whole Net_1 TRAINING step (reference src/train_with_twoDataset.PY:49-55: forward, nll_loss, backward, Adam)
    y = torch.randint(0, 2, (200,))
    yd = y.to(dev)

    def net1(sdict, xx, ee, bb, gpu):
        h, e, acc = xx, ee, None
        for k in (1, 2, 3):
            if gpu:
                h = F.relu(npi.sage_conv(h, e, sdict[f"conv{k}.weight"], sdict[f"conv{k}.bias"]))
                h, e, _, bb, _, _ = NP.topk_pool(h, e, bb, sdict[f"pool{k}.weight"], 0.5, num_graphs=200)
                r = NP.global_max_mean_pool(h, bb, 200)
            else:
                h = F.relu(R.sage_conv(h, e, sdict[f"conv{k}.weight"], sdict[f"conv{k}.bias"]))
                h, e, bb, _, _ = R.topk_pool(h, e, bb, sdict[f"pool{k}.weight"], 0.5)
                r = R.readout(h, bb, 200)
            acc = r if acc is None else acc + r
        z = F.relu(F.linear(acc, sdict["lin1.weight"], sdict["lin1.bias"]))
        z = F.relu(F.linear(z, sdict["lin2.weight"], sdict["lin2.bias"]))
        return F.log_softmax(F.linear(z, sdict["lin3.weight"], sdict["lin3.bias"]), -1)

    pg = {k: v.clone().requires_grad_(True) for k, v in sdd.items()}
    pc = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    og, oc = torch.optim.Adam(pg.values(), lr=1e-3), torch.optim.Adam(pc.values(), lr=1e-3)

    def gpu_train():
        og.zero_grad(set_to_none=True)
        F.nll_loss(net1(pg, xd.detach(), eid, bd, True), yd).backward()
        og.step()

    def cpu_train():
        oc.zero_grad(set_to_none=True)
        F.nll_loss(net1(pc, x, ei, batch, False), y).backward()
        oc.step()
    t_gpu = timeit(gpu_train, 30, torch.cuda.synchronize)
    t_cpu = float("nan") if train_only else timeit(cpu_train, 5)
    print(f"Net_1 training step per batch (fwd + loss + bwd + Adam): GPU {t_gpu:.3f} ms   CPU oracle {t_cpu:.1f} ms")


if __name__ == "__main__":
    main()
