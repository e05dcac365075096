"""The N > 1 path through REAL RCCL: one process per GPU, min(device_count, 8) ranks.

Collected everywhere, skipped unless the box has at least two GPUs (RCCL refuses two ranks on one device; the one-GPU
boxes exercise the same code with gloo / virtual ranks / a world of one: tests/test_dist_gpu.py).  Every rank ships only
its slice of the edge list to its GPU (dist.route_edges), runs the sharded layer forward + backward, then runs the
single-GPU layer of this package over the whole graph on its own GPU and compares the rows it owns and every gradient.
The rank processes are fresh children (spawn): none of them has touched a GPU before it picks its own."""
import json
import os
import time
import queue
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_GPUS = torch.cuda.device_count()          # counting devices does not initialise the GPU
needs_two = pytest.mark.skipif(N_GPUS < 2, reason=f"needs >= 2 GPUs for RCCL with more than one rank (found {N_GPUS})")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, layer_kind, N, E, F, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import npi_gnn_amd as npi
        from npi_gnn_amd import dist as ND
        from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
        ND._COMM_PROFILE = comm = []
        ei = bipartite_edge_index(N, E, seed=13)
        g = torch.Generator().manual_seed(5)
        x, go = torch.randn(N, F, generator=g), torch.randn(N, F, generator=g)
        W = (torch.randn(F, F, generator=g) / F ** 0.5).to(dev)
        b = (torch.randn(F, generator=g) * 0.1).to(dev)
        mine = ei[:, rank * E // world: (rank + 1) * E // world]          # the only part of the edge list this GPU sees
        att = None
        if layer_kind == "edges":
            sg = ND.EdgeShardedGraph(mine, N, rank, world, dev, sliced=True)
            layer = ND.EdgeShardedSAGELayer(sg, W, b)
            xl = x.to(dev).requires_grad_(True)
            rows = torch.arange(sg.lo, sg.hi, device=dev)
            conv = npi.SAGEConv(F, F)
        else:
            hub = None if layer_kind == "rows" else protein_mask(N)
            sg = ND.ShardedGraph(mine, N, rank, world, dev, hub_mask=hub, sliced=True,
                                 small_group=dist.new_group() if layer_kind in ("sage", "gat1") else None)   # two communicators, as bench.py
            rows = sg.own
            xl = sg.shard(x).to(dev).requires_grad_(True)
            if layer_kind in ("sage", "rows"):
                layer, conv = ND.ShardedSAGELayer(sg, W, b), npi.SAGEConv(F, F)
            elif layer_kind == "gcn":
                layer, conv = ND.ShardedGCNLayer(sg, W, b), npi.GCNConv(F, F)
            else:
                H = int(layer_kind[3:])
                att = (torch.randn(1, H, 2 * (F // H), generator=g) * 0.2).to(dev)
                layer, conv = ND.ShardedGATLayer(sg, W, att, b, heads=H), npi.GATConv(F, F // H, heads=H)
        for _ in range(2):                                         # twice: the second step reuses every cached buffer
            layer.zero_grad()
            xl.grad = None
            out = layer(xl)
            out.backward(go.to(dev)[rows])
        conv = conv.to(dev)
        with torch.no_grad():
            conv.weight.copy_(W)
            conv.bias.copy_(b)
            if att is not None:
                conv.att.copy_(att)
        xr = x.to(dev).requires_grad_(True)
        ref = conv(xr, npi.CSRGraph(ei.to(dev), N))
        ref.backward(go.to(dev))
        torch.cuda.synchronize()

        def rel(a, r):
            return float((a.detach() - r.detach()).abs().max() / r.detach().abs().max())
        dx_ref = xr.grad if layer_kind == "edges" else xr.grad[rows]
        errs = {"out": rel(out, ref[rows]), "dX": rel(xl.grad, dx_ref), "dW": rel(layer.weight.grad, conv.weight.grad),
                "db": rel(layer.bias.grad, conv.bias.grad)}
        if att is not None:
            errs["datt"] = rel(layer.att.grad, conv.att.grad)
        q.put((rank, errs, sorted({t for t, _, _ in comm})))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@needs_two
@pytest.mark.parametrize("layer_kind", ["sage", "rows", "gcn", "gat1", "gat2", "edges"])
def test_sharded_layers_through_rccl(layer_kind):
    world = min(N_GPUS, 8)
    N, E, F = 60_003, 1_200_000, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, layer_kind, N, E, F, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        # a rank that dies leaves the others blocked in a collective: poll the queue in short slices and watch the exit codes
        # instead of waiting 600 s for a message that will never come
        deadline = time.time() + 600
        while len(got) < world:
            try:
                rank, errs, tags = q.get(timeout=2)
                got[rank] = (errs, tags)
            except queue.Empty:
                dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
                assert not dead, f"rank(s) died (rank, exit code): {dead}"
                assert time.time() < deadline, f"only ranks {sorted(got)} of {world} reported within 600 s"
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:                       # whatever happened: no rank process (and no GPU it holds) outlives the test
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
    assert sorted(got) == list(range(world))
    for rank, (errs, tags) in got.items():
        for name, e in errs.items():
            assert e < 1e-5, f"rank {rank}: {name} differs from the single-GPU layer by {e} (relative)"
        assert any(t.startswith("fwd_") for t in tags) and any(t.startswith("bwd_") for t in tags)


@needs_two
@pytest.mark.parametrize("partition,conv", [("hubs", "sage"), ("rows", "sage"), ("edges", "sage"), ("hubs", "gat"), ("hubs", "gcn")])
def test_bench_checks_its_own_output_for_more_than_one_gpu(partition, conv):
    """bench.py --gpus N: the JSON line carries parity_max_err (the sharded layer against the single-GPU one)."""
    world = min(N_GPUS, 8)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
                        "--nodes", "200000", "--edges", "4000000", "--partition", partition, "--conv", conv,
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][-1]
    assert line["n_gpus"] == world
    assert line["parity_max_err"] is not None and line["parity_max_err"] < 1e-5, line.get("parity")


def test_bench_parity_block_with_one_rank_through_the_sharded_code(dev):
    """The same self-check on the one-GPU box: --force-sharded runs npi_gnn_amd.dist with a world of one."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for partition, conv in (("hubs", "sage"), ("edges", "sage"), ("hubs", "gat")):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-sharded", "--steps", "2", "--warmup", "1",
                            "--nodes", "200000", "--edges", "4000000", "--partition", partition, "--conv", conv,
                            "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][-1]
        assert line["parity_max_err"] is not None and line["parity_max_err"] < 1e-5, line.get("parity")
        if conv == "sage":
            assert line["parity"]["by_tensor"]["out_rows_fp64_formula"] < 1e-5


def test_bench_rehearsal_of_the_node_run_on_one_gpu(dev):
    """NPI_BENCH_RCCL_SOLO=1: a world of one through what `bench.py --gpus N` goes through -- a real RCCL process group and the
    second communicator for the small exchanges, every collective of the layer, the set-up that times the schedule
    candidates, the all-reduced timings, the parity block."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["NPI_BENCH_RCCL_SOLO"] = "1"
    for conv in ("sage", "gat"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-sharded", "--steps", "2", "--warmup", "1",
                            "--nodes", "200000", "--edges", "4000000", "--conv", conv, "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][-1]
        chosen = (line["config"]["autotune"] or {}).get("chosen")            # (GATConv: no candidates, the default runs)
        assert line["config"]["fallback"] is None, line["config"]
        assert line["config"]["communicators"] == (1 if chosen == "one communicator" else 2), line["config"]
        assert line["parity_max_err"] is not None and line["parity_max_err"] < 1e-5, line.get("parity")
        if conv == "sage":
            tuned = line["config"]["autotune"]
            assert not tuned.get("error") and set(tuned["ms_per_step"]) >= {"default", "gemm_reserve_cus=16", "early_hub_gather",
                                                                          "one communicator"}, tuned
            assert line["exchange"]["by_collective_ms_per_step"], line["exchange"]     # waits on real collectives were timed


def test_bench_falls_back_to_the_conservative_schedule_when_its_preflight_step_fails(dev):
    """bench.py's sharded path runs ONE pre-flight step; a failure (injected here) makes every rank rebuild its shard on the
    round-2 schedule (classic hub layout, no third stream, one GEMM per direction) and say so in the line -- the numbers of the
    fallback run still pass the in-bench parity check."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for inject in ("1", "0"):
        env["NPI_BENCH_INJECT_FAILURE"] = inject
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-sharded", "--nodes", "200000", "--edges",
                            "4000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][-1]
        if inject == "1":
            assert "injected pre-flight failure" in line["config"]["fallback"]
        else:
            assert line["config"]["fallback"] is None
        assert line["parity_max_err"] is not None and line["parity_max_err"] < 1e-5, line.get("parity")
