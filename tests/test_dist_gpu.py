"""Two ranks, ONE MI355X (the GPU box has a single device, and RCCL refuses two ranks on one GPU): the
sharded layer with the real HIP backend in two processes, gloo as the transport for the exchange.
Covers everything of the multi-GPU path except RCCL itself."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_conv as R

pytestmark = pytest.mark.gpu


def _case(N, E, F, kind, seed=0):
    g = torch.Generator().manual_seed(seed)
    hub = None
    if kind == "any":                     # arbitrary digraph: every row is exchanged
        ei = torch.randint(0, N, (2, E), generator=g)
        ei[1, : E // 4] = 3
        ei = torch.cat([ei, ei.flip(0)], dim=1)
    else:                                 # ncRNA-protein shape, protein side replicated
        from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
        ei = bipartite_edge_index(N, E, seed=7)
        hub = protein_mask(N)
    x = torch.randn(N, F, generator=g)
    W = torch.randn(F, F, generator=g) / F ** 0.5
    b = torch.randn(F, generator=g)
    go = torch.randn(N, F, generator=g)
    return ei, x, W, b, go, hub


def _worker(rank, world, port, N, E, F, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from npi_gnn_amd import dist as ND
        dev = torch.device("cuda:0")
        ei, x, W, b, go, hub = _case(N, E, F, kind)
        sg = ND.ShardedGraph(ei, N, rank, world, dev, hub_mask=hub)
        layer = ND.ShardedSAGELayer(sg, W.to(dev), b.to(dev))
        xl = sg.shard(x).to(dev).requires_grad_(True)
        out = layer(xl)
        out.backward(sg.shard(go).to(dev))
        torch.cuda.synchronize()
        # numpy arrays are pickled by value (torch tensors travel through shared-memory files that
        # vanish when this process exits)
        q.put((rank,) + tuple(t.detach().cpu().numpy().copy() for t in (out, xl.grad, layer.weight.grad, layer.bias.grad)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("kind,N,E", [("any", 4001, 30000), ("bipartite", 6003, 60000)])
def test_two_ranks_one_gpu_hip_backend(dev, kind, N, E):
    world, F = 2, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, E, F, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out, dx, dw, db = q.get(timeout=300)
        res[r] = tuple(torch.from_numpy(a) for a in (out, dx, dw, db))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from npi_gnn_amd import dist as ND
    ei, x, W, b, go, hub = _case(N, E, F, kind)
    ref_out, ref_dx, ref_dw, ref_db = R.sage_layer_fwd_bwd(x, ei, W, b, go)
    part = ND.HubPartition(N, world, hub)
    assert torch.allclose(part.unshard([res[r][0] for r in range(world)]), ref_out, atol=1e-4, rtol=1e-4)
    assert torch.allclose(part.unshard([res[r][1] for r in range(world)]), ref_dx, atol=1e-4, rtol=1e-4)
    for r in range(world):
        assert torch.allclose(res[r][2], ref_dw, atol=1e-2, rtol=1e-3)
        assert torch.allclose(res[r][3], ref_db, atol=1e-2, rtol=1e-3)


@pytest.mark.parametrize("hubs", [False, True])
def test_world_one_sharded_path_equals_single_gpu_conv(dev, hubs):
    """bench.py --force-sharded: the W=1 sharded layer (both partition modes) against the plain conv."""
    import npi_gnn_amd as npi
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
    N, E, F = 20000, 300000, 128
    ei = bipartite_edge_index(N, E, seed=3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, F, generator=g)
    W = (torch.randn(F, F, generator=g) / F ** 0.5).to(dev)
    b = torch.randn(F, generator=g).to(dev)
    go = torch.randn(N, F, generator=g)
    sg = ND.ShardedGraph(ei, N, 0, 1, dev, hub_mask=protein_mask(N) if hubs else None)
    layer = ND.ShardedSAGELayer(sg, W, b)
    xl = sg.shard(x).to(dev).requires_grad_(True)
    out = layer(xl)
    out.backward(sg.shard(go).to(dev))
    conv = npi.SAGEConv(F, F).to(dev)
    with torch.no_grad():
        conv.weight.copy_(W)
        conv.bias.copy_(b)
    xr = x.to(dev).requires_grad_(True)
    ref = conv(xr, npi.CSRGraph(ei.to(dev), N))
    ref.backward(go.to(dev))
    own = sg.own
    assert torch.allclose(out, ref[own], atol=1e-5, rtol=1e-5)
    assert torch.allclose(xl.grad, xr.grad[own], atol=1e-5, rtol=1e-5)
    assert torch.allclose(layer.weight.grad, conv.weight.grad, atol=1e-3, rtol=1e-4)
    assert torch.allclose(layer.bias.grad, conv.bias.grad, atol=1e-3, rtol=1e-4)


def _rccl_solo_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import npi_gnn_amd as npi
        from npi_gnn_amd import dist as ND
        from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
        ND.ALWAYS_COMMUNICATE = True          # a single rank normally copies; go through RCCL instead
        ND._COMM_PROFILE = comm = []          # bench.py's exposed-communication hook
        N, E, F = 20000, 300000, 128
        ei = bipartite_edge_index(N, E, seed=3)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(N, F, generator=g)
        W = (torch.randn(F, F, generator=g) / F ** 0.5).to(dev)
        b = torch.randn(F, generator=g).to(dev)
        go = torch.randn(N, F, generator=g)
        errs = []
        for hubs in (True, False):
            sg = ND.ShardedGraph(ei, N, 0, 1, dev, hub_mask=protein_mask(N) if hubs else None)
            layer = ND.ShardedSAGELayer(sg, W, b)
            xl = sg.shard(x).to(dev).requires_grad_(True)
            out = layer(xl)
            out.backward(sg.shard(go).to(dev))
            conv = npi.SAGEConv(F, F).to(dev)
            with torch.no_grad():
                conv.weight.copy_(W)
                conv.bias.copy_(b)
            xr = x.to(dev).requires_grad_(True)
            ref = conv(xr, npi.CSRGraph(ei.to(dev), N))
            ref.backward(go.to(dev))
            torch.cuda.synchronize()
            def rel(a, b):                   # max error relative to the largest reference magnitude
                return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max())
            errs.append((rel(out, ref[sg.own]), rel(xl.grad, xr.grad[sg.own]), rel(layer.weight.grad, conv.weight.grad)))
        tags = sorted({t for t, _, _ in comm})
        stall_ms = [e0.elapsed_time(e1) for _, e0, e1 in comm]
        q.put((errs, tags, min(stall_ms), max(stall_ms)))
    finally:
        dist.destroy_process_group()


def test_collectives_through_rccl_with_one_rank(dev):
    """all_gather_into_tensor / reduce_scatter_tensor / all_reduce exactly as the N>1 path issues them
    (views of the table, async work handles), on the real RCCL backend with a world of one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_solo_worker, args=(_free_port(), q))
    p.start()
    errs, tags, stall_min, stall_max = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    # every wait on a collective is bracketed by HIP events on the waiting stream (hubs: gather + reduce-scatter per direction)
    assert tags == ["bwd_all_gather", "bwd_all_reduce_db", "bwd_all_reduce_dw", "bwd_reduce_scatter", "fwd_all_gather",
                    "fwd_reduce_scatter"]
    assert 0.0 <= stall_min <= stall_max < 1000.0
    for e_out, e_dx, e_dw in errs:
        assert e_out < 1e-5 and e_dx < 1e-5 and e_dw < 1e-5
