"""Two ranks, ONE MI355X (the GPU box has a single device, and RCCL refuses two ranks on one GPU): the
sharded layer with the real HIP backend in two processes, gloo as the transport for the exchange.
Covers everything of the multi-GPU path except RCCL itself."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import GRAD_REL, rel_max
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


def _att(F, H=1):
    return torch.randn(1, H, 2 * (F // H), generator=torch.Generator().manual_seed(9)) * 0.2


def _make_layer(ND, layer_kind, sg, W, b, dev, F):
    if layer_kind == "sage":
        return ND.ShardedSAGELayer(sg, W.to(dev), b.to(dev))
    if layer_kind == "gcn":
        return ND.ShardedGCNLayer(sg, W.to(dev), b.to(dev))
    H = int(layer_kind[3:])
    return ND.ShardedGATLayer(sg, W.to(dev), _att(F, H).to(dev), b.to(dev), heads=H)


def _reference(layer_kind, ei, x, W, b, go, F, fp64=False):
    """the oracle's layer; ``fp64``: evaluated (and returned) in double -- the reference of the parameter-gradient bars"""
    keep = (lambda t: t) if fp64 else (lambda t: t.float())
    if layer_kind == "sage":
        if fp64:
            return R.sage_layer_fwd_bwd(x.double(), ei, W.double(), b.double(), go.double())
        return R.sage_layer_fwd_bwd(x, ei, W, b, go)
    xr, Wr, br = (t.double().clone().requires_grad_(True) for t in (x, W, b))      # fp64: hub rows are long
    if layer_kind == "gcn":
        out = R.gcn_conv(xr, ei, Wr, br)
        out.backward(go.double())
        return tuple(keep(t) for t in (out.detach(), xr.grad, Wr.grad, br.grad))
    H = int(layer_kind[3:])
    att = _att(F, H).double().requires_grad_(True)
    out = R.gat_conv(xr, ei, Wr, att, br, heads=H)
    out.backward(go.double())
    return tuple(keep(t) for t in (out.detach(), xr.grad, Wr.grad, br.grad, att.grad))


def _worker(rank, world, port, N, E, F, kind, layer_kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from npi_gnn_amd import dist as ND
        dev = torch.device("cuda:0")
        ei, x, W, b, go, hub = _case(N, E, F, kind)
        sg = ND.ShardedGraph(ei, N, rank, world, dev, hub_mask=hub)
        layer = _make_layer(ND, layer_kind, sg, W, b, dev, F)
        xl = sg.shard(x).to(dev).requires_grad_(True)
        out = layer(xl)
        out.backward(sg.shard(go).to(dev))
        torch.cuda.synchronize()
        extra = (layer.att.grad,) if layer_kind.startswith("gat") else ()
        # numpy arrays are pickled by value (torch tensors travel through shared-memory files that
        # vanish when this process exits)
        q.put((rank,) + tuple(t.detach().cpu().numpy().copy() for t in (out, xl.grad, layer.weight.grad, layer.bias.grad) + extra))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("kind,N,E,layer_kind", [("any", 4001, 30000, "sage"), ("bipartite", 6003, 60000, "sage"),
                                                 ("bipartite", 6003, 60000, "gcn"), ("any", 4001, 30000, "gcn"),
                                                 ("bipartite", 6003, 60000, "gat1"), ("any", 4001, 30000, "gat1"),
                                                 ("bipartite", 6003, 60000, "gat4")])
def test_two_ranks_one_gpu_hip_backend(dev, kind, N, E, layer_kind):
    world, F = 2, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, E, F, kind, layer_kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        got = q.get(timeout=300)
        res[got[0]] = tuple(torch.from_numpy(a) for a in got[1:])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from npi_gnn_amd import dist as ND
    ei, x, W, b, go, hub = _case(N, E, F, kind)
    ref = _reference(layer_kind, ei, x, W, b, go, F)
    ref_out, ref_dx, ref_dw, ref_db = ref[:4]
    part = ND.HubPartition(N, world, hub)
    assert torch.allclose(part.unshard([res[r][0] for r in range(world)]), ref_out, atol=1e-4, rtol=1e-4)
    assert torch.allclose(part.unshard([res[r][1] for r in range(world)]), ref_dx, atol=1e-4, rtol=1e-4)
    ref64 = _reference(layer_kind, ei, x, W, b, go, F, fp64=True)          # parameter gradients: 1e-5 of max against fp64
    for r in range(world):
        assert rel_max(res[r][2], ref64[2]) <= GRAD_REL and rel_max(res[r][3], ref64[3]) <= GRAD_REL
        if layer_kind.startswith("gat"):
            assert rel_max(res[r][4], ref64[4]) <= GRAD_REL


@pytest.mark.parametrize("world", [1, 3, 8])
@pytest.mark.parametrize("layer_kind", ["gcn", "gat1", "gat2"])
def test_virtual_ranks_gcn_gat_on_one_gpu(dev, world, layer_kind):
    """SURVEY.md 8(e), one-GPU form: the W shards of a GCN / GAT layer run one after the other in ONE process, the
    collectives replaced by an in-process exchange (a gloo-free stand-in that holds every rank's buffers), HIP backend."""
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
    N, E, F = 5003, 60000, 128
    ei = bipartite_edge_index(N, E, seed=21)
    g = torch.Generator().manual_seed(4)
    x, go = torch.randn(N, F, generator=g), torch.randn(N, F, generator=g)
    W, b = torch.randn(F, F, generator=g) / F ** 0.5, torch.randn(F, generator=g)
    ref = _reference(layer_kind, ei, x, W, b, go, F)
    outs, dxs, dws = _run_virtual(ND, world, layer_kind, ei, N, F, x, go, W, b, protein_mask(N), dev)
    part = ND.HubPartition(N, world, protein_mask(N))
    assert torch.allclose(part.unshard(outs), ref[0], atol=1e-4, rtol=1e-4)
    assert torch.allclose(part.unshard(dxs), ref[1], atol=1e-4, rtol=1e-4)
    dw64 = _reference(layer_kind, ei, x, W, b, go, F, fp64=True)[2]
    for dw in dws:
        assert rel_max(dw, dw64) <= GRAD_REL


def _run_virtual(ND, world, layer_kind, ei, N, F, x, go, W, b, hub, dev, schedule=None):
    """W ranks in ONE process without threads, in exact lock step (npi_gnn_amd.virtual.LockStep: the ranks run one after the
    other, pass after pass; collective number k returns its true result once every rank's input to it is known)."""
    from npi_gnn_amd.schedule import DEFAULT
    from npi_gnn_amd.virtual import LockStep
    with LockStep(world, max_passes=64) as ls:
        sgs = [ND.ShardedGraph(ei, N, r, world, dev, hub_mask=hub, schedule=schedule or DEFAULT) for r in range(world)]

        def run_rank(r):
            sg = sgs[r]
            layer = _make_layer(ND, layer_kind, sg, W, b, dev, F)
            xl = sg.shard(x).to(dev).requires_grad_(True)
            out = layer(xl)
            out.backward(sg.shard(go).to(dev))
            torch.cuda.synchronize()
            return out.detach(), xl.grad, layer.weight.grad
        res = ls.run(run_rank)
    return [r[0].cpu() for r in res], [r[1].cpu() for r in res], [r[2].cpu() for r in res]


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
    assert rel_max(layer.weight.grad, conv.weight.grad) <= GRAD_REL and rel_max(layer.bias.grad, conv.bias.grad) <= GRAD_REL


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
        # GCN and GAT layers through the same real collectives (incl. the MAX all-reduce of the hub-row maxima)
        # ... with the small exchanges on a SECOND RCCL communicator (ShardedGraph(small_group=), as bench.py's N > 1 run)
        sg = ND.ShardedGraph(ei, N, 0, 1, dev, hub_mask=protein_mask(N), small_group=dist.new_group())
        assert sg.small_group is not sg.group
        att = _att(F, 2).to(dev)
        for kind in ("gcn", "gat"):
            if kind == "gcn":
                layer, conv = ND.ShardedGCNLayer(sg, W, b), npi.GCNConv(F, F).to(dev)
            else:
                layer, conv = ND.ShardedGATLayer(sg, W, att, b, heads=2), npi.GATConv(F, F // 2, heads=2).to(dev)
                with torch.no_grad():
                    conv.att.copy_(att)
            with torch.no_grad():
                conv.weight.copy_(W)
                conv.bias.copy_(b)
            xl = sg.shard(x).to(dev).requires_grad_(True)
            out = layer(xl)
            out.backward(sg.shard(go).to(dev))
            xr = x.to(dev).requires_grad_(True)
            ref = conv(xr, npi.CSRGraph(ei.to(dev), N))
            ref.backward(go.to(dev))
            torch.cuda.synchronize()
            errs.append((rel(out, ref[sg.own]), rel(xl.grad, xr.grad[sg.own]), rel(layer.weight.grad, conv.weight.grad)))
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
    # (dW and db are summed in ONE exchange)
    assert tags == ["bwd_all_gather", "bwd_all_reduce_params", "bwd_reduce_scatter", "fwd_all_gather", "fwd_reduce_scatter"]
    assert 0.0 <= stall_min <= stall_max < 1000.0
    assert len(errs) == 4                   # SAGE (hubs, rows), GCN, GAT
    for e_out, e_dx, e_dw in errs:
        assert e_out < 1e-5 and e_dx < 1e-5 and e_dw < 1e-5


def test_world_one_edge_sharded_baseline_equals_single_gpu_conv(dev):
    """bench.py --partition edges --force-sharded: the north-star's baseline split with one rank."""
    import npi_gnn_amd as npi
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd.synth import bipartite_edge_index
    N, E, F = 20000, 300000, 128
    ei = bipartite_edge_index(N, E, seed=3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, F, generator=g)
    W = (torch.randn(F, F, generator=g) / F ** 0.5).to(dev)
    b = torch.randn(F, generator=g).to(dev)
    go = torch.randn(N, F, generator=g).to(dev)
    sg = ND.EdgeShardedGraph(ei, N, 0, 1, dev)
    layer = ND.EdgeShardedSAGELayer(sg, W, b)
    xl = x.to(dev).requires_grad_(True)
    out = layer(xl)
    out.backward(go)
    conv = npi.SAGEConv(F, F).to(dev)
    with torch.no_grad():
        conv.weight.copy_(W)
        conv.bias.copy_(b)
    xr = x.to(dev).requires_grad_(True)
    ref = conv(xr, npi.CSRGraph(ei.to(dev), N))
    ref.backward(go)
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5)
    assert torch.allclose(xl.grad, xr.grad, atol=1e-5, rtol=1e-5)
    assert rel_max(layer.weight.grad, conv.weight.grad) <= GRAD_REL


@pytest.mark.parametrize("layer_kind", ["sage", "gat1"])
def test_virtual_ranks_at_the_c4_size(dev, layer_kind):
    """the sharded layers at the FULL C4 size (N = 1M, E = 20M, hidden 256) -- two virtual ranks on the
    one GPU, hub cut, lock-step replay of the collectives -- against the single-GPU layer of the same kernels."""
    import npi_gnn_amd as npi
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
    N, E, F, world = 1_000_000, 20_000_000, 256, 2
    ei = bipartite_edge_index(N, E, seed=20260310)
    g = torch.Generator().manual_seed(6)
    x, go = torch.randn(N, F, generator=g), torch.randn(N, F, generator=g)
    W, b = torch.randn(F, F, generator=g) / F ** 0.5, torch.randn(F, generator=g) * 0.1
    hub = protein_mask(N)
    outs, dxs, dws = _run_virtual(ND, world, layer_kind, ei, N, F, x, go, W, b, hub, dev)
    part = ND.HubPartition(N, world, hub)
    out, dx = part.unshard(outs), part.unshard(dxs)
    if layer_kind == "sage":
        conv = npi.SAGEConv(F, F).to(dev)
    else:
        conv = npi.GATConv(F, F, heads=1).to(dev)
        with torch.no_grad():
            conv.att.copy_(_att(F, 1))
    with torch.no_grad():
        conv.weight.copy_(W)
        conv.bias.copy_(b)
    xr = x.to(dev).requires_grad_(True)
    ref = conv(xr, npi.CSRGraph(ei.to(dev), N))
    ref.backward(go.to(dev))

    def rel(a, r):
        return float((a - r.cpu()).abs().max() / r.abs().max())
    assert rel(out, ref.detach()) < 1e-5
    assert rel(dx, xr.grad) < 2e-5
    for dw in dws:
        assert rel(dw, conv.weight.grad) < 1e-4


@pytest.mark.parametrize("layer_kind", ["sage"])     # ("gat1": 18 s of the suite -- under `-m slow`, tests/test_slow_campaigns.py; the sharded
def test_eight_virtual_ranks_at_a_quarter_of_c4_match_the_single_gpu_layer(dev, layer_kind):      # GATConv at > 100,000 rows per rank
    # stays under `-m gpu` in test_virtual_ranks_at_the_c4_size[gat1] and the eight lock-step ranks of the C5 stack)
    """The 8-rank hub cut on a graph big enough for everything the C4 run meets -- rows of 10^4+ entries cut over hundreds of
    64-entry items on every side, > 100,000 rows per rank (the split projection, the third stream, the rank-2 store epilogue of the
    sharded GATConv and its dW under the row sums are all on), 256 channels -- in lock step on one GPU against this package's
    single-GPU layer on the whole graph: every rank's rows of out and dX at 1e-5, dW at 1e-4 of its scale."""
    import npi_gnn_amd as npi
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
    N, E, F, world = 1_000_000, 5_000_000, 256, 8
    ei = bipartite_edge_index(N, E, seed=17)
    g = torch.Generator().manual_seed(8)
    x, go = torch.randn(N, F, generator=g), torch.randn(N, F, generator=g)
    W, b = torch.randn(F, F, generator=g) / F ** 0.5, torch.randn(F, generator=g)
    hub = protein_mask(N)
    conv = (npi.SAGEConv(F, F) if layer_kind == "sage" else npi.GATConv(F, F, heads=1)).to(dev)
    with torch.no_grad():
        conv.weight.copy_(W.to(dev))
        conv.bias.copy_(b.to(dev))
        if layer_kind != "sage":
            conv.att.copy_(_att(F, 1).to(dev))
    xr = x.to(dev).requires_grad_(True)
    ref = conv(xr, npi.CSRGraph(ei.to(dev), N))
    ref.backward(go.to(dev))
    ref_out, ref_dx, ref_dw = ref.detach().cpu(), xr.grad.cpu(), conv.weight.grad.cpu()
    del conv, xr, ref
    torch.cuda.empty_cache()
    outs, dxs, dws = _run_virtual(ND, world, layer_kind, ei, N, F, x, go, W, b, hub, dev)
    part = ND.HubPartition(N, world, hub)
    so, sx = float(ref_out.abs().max()), float(ref_dx.abs().max())
    assert float((part.unshard(outs) - ref_out).abs().max()) <= 1e-5 * so
    assert float((part.unshard(dxs) - ref_dx).abs().max()) <= 1e-5 * sx
    for dw in dws:
        assert float((dw - ref_dw).abs().max()) <= 1e-4 * float(ref_dw.abs().max())


@pytest.mark.gpu
def test_emulated_wire_holds_cus_for_the_stated_time_and_changes_no_result(dev):
    """StubCollectives(wire_gbps=): every stand-in exchange is preceded by a no-op kernel that holds CUs for latency + wire bytes /
    rate (npi_hold_cus) -- the layer's numbers are those of the plain stand-ins, a rank's step gets longer by about the wire time
    that is not hidden, and npi_hold_cus itself lasts what it is asked to."""
    import time
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd._lib import check, load, stream_ptr
    from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
    from npi_gnn_amd.virtual import StubCollectives
    check(load().npi_hold_cus(16, 1_000, None, stream_ptr(dev)), "npi_hold_cus")             # (the first launch uploads the kernel)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); check(load().npi_hold_cus(16, 300_000, None, stream_ptr(dev)), "npi_hold_cus"); e1.record()
    torch.cuda.synchronize()
    assert 0.29 <= e0.elapsed_time(e1) <= 0.6
    N, E, F, W = 200_000, 4_000_000, 128, 4
    ei = bipartite_edge_index(N, E, seed=5).to(dev)
    g = torch.Generator().manual_seed(1)
    Wm, b = (torch.randn(F, F, generator=g) / F ** 0.5).to(dev), torch.randn(F, generator=g).to(dev)
    res = {}
    for bw in (None, 5.0):
        with StubCollectives(W, copy_stream=torch.cuda.Stream(device=dev), wire_gbps=bw) as stub:
            sg = ND.ShardedGraph(ei, N, 1, W, dev, hub_mask=protein_mask(N).to(dev))
            layer = ND.ShardedSAGELayer(sg, Wm, b)
            x = torch.randn(sg.n_local, F, device=dev, generator=torch.Generator(device=dev).manual_seed(2)).requires_grad_(True)
            go = torch.ones(sg.n_local, F, device=dev)
            for _ in range(3):
                layer.zero_grad(); x.grad = None
                out = layer(x); out.backward(go)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                layer.zero_grad(); x.grad = None
                out = layer(x); out.backward(go)
            torch.cuda.synchronize()
            res[bw] = (out.detach().clone(), x.grad.clone(), (time.perf_counter() - t0) / 5)
    assert torch.equal(res[None][0], res[5.0][0]) and torch.equal(res[None][1], res[5.0][1])
    # a GATConv rank step with its small exchanges on their own lane: the same numbers as on one lane, emulated wire or not
    from npi_gnn_amd.virtual import SMALL_LANE
    att = (torch.randn(1, 1, 2 * F, generator=g) * 0.1).to(dev)
    gat = {}
    for lanes, bw in ((1, None), (2, None), (2, 50.0)):
        with StubCollectives(W, copy_stream=torch.cuda.Stream(device=dev), wire_gbps=bw,
                             copy_stream2=torch.cuda.Stream(device=dev) if lanes == 2 else None):
            sg = ND.ShardedGraph(ei, N, 1, W, dev, hub_mask=protein_mask(N).to(dev), small_group=SMALL_LANE if lanes == 2 else None)
            layer = ND.ShardedGATLayer(sg, Wm, att, b)
            x = torch.randn(sg.n_local, F, device=dev, generator=torch.Generator(device=dev).manual_seed(2)).requires_grad_(True)
            for _ in range(2):
                layer.zero_grad(); x.grad = None
                out = layer(x); out.backward(torch.ones_like(out))
            torch.cuda.synchronize()
            gat[(lanes, bw)] = (out.detach().clone(), x.grad.clone(), layer.weight.grad.clone(), layer.att.grad.clone())
    for key in ((2, None), (2, 50.0)):
        assert all(torch.equal(a_, b_) for a_, b_ in zip(gat[(1, None)], gat[key])), key
    # four exchanges of ~7.7 MB at 5 GB/s = 6 ms of wire time per step, far above what the host's launch pace can hide (at
    # 50 GB/s the 0.7 ms disappeared behind a 0.96 ms host-bound step on a slower box: round 5)
    assert res[5.0][2] > res[None][2] + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["8 virtual ranks", "world 1 through RCCL"])
def test_the_default_sharded_sage_step_captures_into_a_hip_graph_and_replays_bit_equal(dev, how):
    """capturing the DEFAULT sharded SAGEConv schedule took the process down in capture_end (the partial
    stream forked from the side stream forked from the launch stream).  While a capture is on, the layer now runs its partial
    side in line (HipBackend.partial_stream) -- the same kernels, the same numbers: a captured step replays bit-equal to the
    eager one, for rank 1 of 8 with stand-in collectives and for a world of one through a real RCCL process group.  In a
    CHILD process: a crash of the HIP runtime must fail this test, not end the session."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "virtual_rank_probe.py"), "--conv", "sage", "--capture", "--check-replay",
           "--steps", "5", "--hidden", "128"]
    if how.startswith("8"):
        # (the stand-in collectives keep their own copy stream in the eager steps; inside the capture they run in line --
        # virtual.StubCollectives._issue -- as a third level of stream forks is what takes hipStreamEndCapture down)
        cmd += ["--world", "8", "--rank", "1", "--nodes", "1000000", "--edges", "8000000"]
    else:
        cmd += ["--rccl", "--nodes", "200000", "--edges", "4000000"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    assert "replay == eager" in p.stdout and "capture=True" in p.stdout, p.stdout[-1500:]
