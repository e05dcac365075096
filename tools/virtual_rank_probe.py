#!/usr/bin/env python3
"""One virtual rank of a W-rank run on this GPU (collectives = local copies on a stream of their own, npi_gnn_amd.virtual): wall time per step, GPU time per step
(events), and -- under `rocprofv3 --kernel-trace --stats` -- the kernels of a rank's step.
usage: python tools/virtual_rank_probe.py [--world 8] [--rank 0] [--conv sage|gat] [--partition hubs] [--steps 20] [--capture]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--conv", default="sage")
    ap.add_argument("--partition", default="hubs")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=20_000_000)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--capture", action="store_true")
    ap.add_argument("--inline-copies", action="store_true", help="stand-in copies on the compute stream (default: their own stream)")
    ap.add_argument("--wire-gbps", type=float, default=0.0, help="emulate the exchanges: hold CUs for latency + wire bytes / this rate")
    ap.add_argument("--held-cus", type=int, default=16)
    ap.add_argument("--wire-sweep", default="", help="comma-separated rates: after the plain run, one emulated run per rate (one line each)")
    ap.add_argument("--hp-copies", action="store_true", help="the stand-in collectives on HIGH-priority streams")
    ap.add_argument("--two-lanes", action="store_true", help="the small exchanges on a second communicator (ShardedGraph(small_group=))")
    ap.add_argument("--sched", default="", help="Schedule overrides, e.g. 'split_projection=False,partial_stream=False'")
    ap.add_argument("--check-replay", action="store_true",
                    help="with --capture: the replayed step's out / dX / dW must be BIT-equal to the eager step's (prints 'replay == eager')")
    ap.add_argument("--rccl", action="store_true",
                    help="a world of ONE through a real RCCL process group (every collective of the layer issued: dist.ALWAYS_COMMUNICATE) "
                         "instead of the stand-in collectives")
    a = ap.parse_args()
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
    dev = torch.device("cuda:0")
    N, E, F, W, r = a.nodes, a.edges, a.hidden, a.world, a.rank

    from npi_gnn_amd.virtual import SMALL_LANE, StubCollectives
    prio = -1 if a.hp_copies else 0
    if a.rccl:
        import tempfile
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="npi_probe_store_"), rank=0, world_size=1,
                                device_id=dev)
        ND.ALWAYS_COMMUNICATE = True
        W, r = 1, 0
        stub = None
    else:
        stub = StubCollectives(W, copy_stream=None if a.inline_copies else torch.cuda.Stream(device=dev, priority=prio),
                               wire_gbps=a.wire_gbps or None, held_cus=a.held_cus,
                               copy_stream2=torch.cuda.Stream(device=dev, priority=prio) if (a.two_lanes and not a.inline_copies) else None)
        stub.__enter__()                                   # for the life of the process
    ei = bipartite_edge_index(N, E, seed=20260310).to(dev)
    g = torch.Generator().manual_seed(3)
    Wm = ((torch.rand(F, F, generator=g) * 2 - 1) / F ** 0.5).to(dev)
    b = ((torch.rand(F, generator=g) * 2 - 1) / F ** 0.5).to(dev)
    from npi_gnn_amd.schedule import DEFAULT
    sch = DEFAULT.but(**eval("dict(" + a.sched + ")"))
    sg = ND.ShardedGraph(ei, N, r, W, dev, hub_mask=protein_mask(N).to(dev) if a.partition == "hubs" else None, schedule=sch,
                         small_group=SMALL_LANE if a.two_lanes else None)
    del ei
    if a.conv == "sage":
        layer = ND.ShardedSAGELayer(sg, Wm, b)
    elif a.conv == "gcn":
        layer = ND.ShardedGCNLayer(sg, Wm, b)
    else:
        layer = ND.ShardedGATLayer(sg, Wm, (torch.randn(1, 1, 2 * F, generator=g) * 0.1).to(dev), b)
    x = torch.randn(sg.n_local, F, device=dev).requires_grad_(True)
    go = torch.randn(sg.n_local, F, device=dev)

    keep = {}

    def step():
        layer.zero_grad()
        x.grad = None
        out = layer(x)
        out.backward(go)
        keep.update(out=out.detach(), dx=x.grad, dw=layer.weight.grad)     # (detached: the graph must not outlive the step)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    eager = {k: v.detach().clone() for k, v in keep.items()}
    if a.capture:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        run = gr.replay
        if a.check_replay:
            for _ in range(2):
                gr.replay()
            torch.cuda.synchronize()
            bad = [k for k, v in keep.items() if not torch.equal(v, eager[k])]
            print("replay == eager" if not bad else f"replay != eager: {bad}", flush=True)
    else:
        run = step
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        run()
    e1.record()
    t_host = (time.perf_counter() - t0) / a.steps * 1e3
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / a.steps * 1e3
    print(f"world {W} rank {r} {a.partition} {a.conv}: n_local {sg.n_local} entries {sg.local_nnz}  wall {t_wall:.3f} ms/step, "
          f"events {e0.elapsed_time(e1) / a.steps:.3f} ms/step, host issue {t_host:.3f} ms/step, capture={a.capture}")
    for bw in [float(v) for v in a.wire_sweep.split(",") if v and stub is not None]:
        stub.wire_gbps = bw
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run()
        torch.cuda.synchronize()
        print(f"emulated wire {bw:g} GB/s, {a.held_cus} CUs held: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms/step")


if __name__ == "__main__":
    main()
