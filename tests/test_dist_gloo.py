"""world_size-2 (and 3) gloo runs of the sharded layer on CPU: the partition, the all-gather
exchange and the gradient all-reduce of npi_gnn_amd.dist, with the local kernels replaced by a torch
stand-in (the HIP backend needs a GPU).  Checked against the single-process oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from npi_gnn_amd import dist as ND
from oracle import ref_conv as R


class TorchBackend:
    """Stand-in for HipBackend: same contract, plain torch ops."""

    def make_side(self, key, val, n_rows, n_cols):
        return key, val, n_rows, n_cols

    def segsum(self, side, table, mean=False):
        key, val, n_rows, n_cols = side
        assert table.size(0) == n_cols
        out = torch.zeros(n_rows, table.size(1)).index_add_(0, key, table[val])
        if mean:
            out = out / torch.bincount(key, minlength=n_rows).clamp(min=1).float().view(-1, 1)
        return out

    def linear_fwd(self, a, w, b):
        return a @ w + (b if b is not None else 0)

    def linear_bwd_data(self, dc, w, rowscale):
        return (dc @ w.t()) * rowscale.view(-1, 1)

    def linear_bwd_weight(self, a, dc, want_bias):
        return a.t() @ dc, (dc.sum(0) if want_bias else None)


def _case(N, E, F, kind, seed=0):
    """kind: 'any' = arbitrary digraph (every row is exchanged), 'bipartite' = ncRNA-protein shape with
    the protein side as hubs, 'auto' = the same with hubs found from degrees"""
    g = torch.Generator().manual_seed(seed)
    hub = None
    if kind == "any":
        ei = torch.randint(0, N, (2, E), generator=g)      # includes some self loops and duplicates
        ei[1, : E // 4] = 3                                 # a hub row
    else:
        n_rna = N - max(N // 8, 2)
        rna = torch.randint(0, n_rna, (E // 2,), generator=g)
        pro = n_rna + (torch.rand(E // 2, generator=g) ** 3 * (N - n_rna)).long().clamp(max=N - n_rna - 1)
        ei = torch.cat([torch.stack([rna, pro]), torch.stack([pro, rna])], 1)
        ei = torch.cat([ei, torch.tensor([[N - 1, 0], [N - 1, 0]])], 1)      # self loops on a hub and a light node
        if kind == "bipartite":
            hub = torch.arange(N) >= n_rna
        else:
            hub = ND.auto_hubs(ei, N, ratio=2.0, max_fraction=0.9)
            assert hub is not None and 0 < int(hub.sum()) < N
    x = torch.randn(N, F, generator=g)
    W = torch.randn(F, F, generator=g) / F ** 0.5
    b = torch.randn(F, generator=g)
    go = torch.randn(N, F, generator=g)
    return ei, x, W, b, go, hub


def _worker(rank, world, port, N, E, F, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ei, x, W, b, go, hub = _case(N, E, F, kind)
        sg = ND.ShardedGraph(ei, N, rank, world, torch.device("cpu"), backend=TorchBackend(), hub_mask=hub)
        layer = ND.ShardedSAGELayer(sg, W, b)
        xl = sg.shard(x).clone().requires_grad_(True)
        out = layer(xl)
        out.backward(sg.shard(go))
        # numpy arrays are pickled by value (torch tensors travel through shared-memory files that
        # vanish when this process exits)
        q.put((rank,) + tuple(t.detach().numpy().copy() for t in (out, xl.grad, layer.weight.grad, layer.bias.grad)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,N,kind", [(2, 101, "any"), (3, 64, "any"), (2, 203, "bipartite"),
                                          (3, 160, "bipartite"), (2, 120, "auto"), (8, 333, "bipartite"),
                                          (4, 61, "any")])
def test_sharded_layer_matches_single_process_oracle(world, N, kind):
    E, F = 900, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, E, F, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out, dx, dw, db = q.get(timeout=180)
        res[r] = tuple(torch.from_numpy(a) for a in (out, dx, dw, db))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ei, x, W, b, go, hub = _case(N, E, F, kind)
    ref_out, ref_dx, ref_dw, ref_db = R.sage_layer_fwd_bwd(x, ei, W, b, go)
    part = ND.HubPartition(N, world, hub)
    out = part.unshard([res[r][0] for r in range(world)])
    dx = part.unshard([res[r][1] for r in range(world)])
    assert torch.allclose(out, ref_out, atol=1e-5, rtol=1e-5)
    assert torch.allclose(dx, ref_dx, atol=1e-5, rtol=1e-5)
    for r in range(world):                                  # gradients are all-reduced: identical everywhere
        assert torch.allclose(res[r][2], ref_dw, atol=1e-4, rtol=1e-5)
        assert torch.allclose(res[r][3], ref_db, atol=1e-4, rtol=1e-5)


def test_partition_maps():
    hub = torch.tensor([0, 0, 0, 1, 0, 0, 1, 1, 0, 0], dtype=torch.bool)
    part = ND.HubPartition(10, 2, hub)
    assert (part.nL, part.nH, part.h_per, part.hub_rows) == (7, 3, 2, 4)
    assert [part.n_light(r) for r in range(2)] == [4, 3] and [part.n_hub(r) for r in range(2)] == [2, 1]
    assert part.own_ids(0).tolist() == [0, 2, 5, 9, 3, 7] and part.own_ids(1).tolist() == [1, 4, 8, 6]
    assert part.hub_row(torch.tensor([3, 6, 7])).tolist() == [0, 2, 1]
    x = torch.arange(10.0).view(-1, 1)
    assert torch.equal(part.unshard([part.shard(x, r) for r in range(2)]), x)
    # all-hub partition == plain strided destination rows
    full = ND.HubPartition(10, 4)
    assert full.nL == 0 and full.own_ids(1).tolist() == [1, 5, 9]
    assert full.hub_row(torch.arange(10)).tolist() == [0, 3, 6, 9, 1, 4, 7, 10, 2, 5]
    # every non-loop edge lands in exactly one side of one rank, per direction
    ei = torch.tensor([[0, 3, 6, 7, 4, 4, 3], [3, 1, 7, 2, 6, 4, 3]])
    tot = 0
    for r in range(2):
        a, b = ND.local_sides(ei[0], ei[1], part, r)
        tot += a[0].numel() - a[2] + b[0].numel()          # side A carries one loop per local row
    assert tot == 5                                         # (4,4) and (3,3) are self loops
    with pytest.raises(ValueError):
        ND.local_sides(torch.tensor([0]), torch.tensor([1]), part, 0)     # light-light edge


def test_auto_hubs_falls_back_when_there_is_no_small_hub_side():
    g = torch.Generator().manual_seed(0)
    ei = torch.randint(0, 50, (2, 400), generator=g)
    assert ND.auto_hubs(ei, 50) is None
