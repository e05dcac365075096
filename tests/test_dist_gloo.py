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

    def __init__(self, by_dst, by_src, n_local, table_rows, loop_col_offset):
        self.kd, self.vd = by_dst
        self.ks, self.vs = by_src
        self.n, self.loop = n_local, loop_col_offset + torch.arange(n_local)
        self.cnt = torch.bincount(self.kd, minlength=n_local).float() + 1.0

    def _agg(self, k, v, table):
        out = torch.zeros(self.n, table.size(1)).index_add_(0, k, table[v])
        return out + table[self.loop]

    def aggregate_mean(self, table):
        return self._agg(self.kd, self.vd, table) / self.cnt.view(-1, 1)

    def aggregate_t(self, table):
        return self._agg(self.ks, self.vs, table)

    def inv_count(self):
        return 1.0 / self.cnt

    def linear_fwd(self, a, w, b):
        return a @ w + (b if b is not None else 0)

    def linear_bwd_data(self, dc, w, rowscale):
        return (dc @ w.t()) * rowscale.view(-1, 1)

    def linear_bwd_weight(self, a, dc, want_bias):
        return a.t() @ dc, (dc.sum(0) if want_bias else None)


def _case(N, E, F, seed=0):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, N, (2, E), generator=g)          # includes some self loops and duplicates
    ei[1, : E // 4] = 3                                     # a hub row
    x = torch.randn(N, F, generator=g)
    W = torch.randn(F, F, generator=g) / F ** 0.5
    b = torch.randn(F, generator=g)
    go = torch.randn(N, F, generator=g)
    return ei, x, W, b, go


def _worker(rank, world, port, N, E, F, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ei, x, W, b, go = _case(N, E, F)
        sg = ND.ShardedGraph(ei, N, rank, world, torch.device("cpu"), backend_factory=TorchBackend)
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


@pytest.mark.parametrize("world,N", [(2, 101), (3, 64)])
def test_sharded_layer_matches_single_process_oracle(world, N):
    E, F = 900, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, E, F, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out, dx, dw, db = q.get(timeout=180)
        res[r] = tuple(torch.from_numpy(a) for a in (out, dx, dw, db))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ei, x, W, b, go = _case(N, E, F)
    ref_out, ref_dx, ref_dw, ref_db = R.sage_layer_fwd_bwd(x, ei, W, b, go)
    part = ND.StridedPartition(N, world)
    out = part.unshard([res[r][0] for r in range(world)])
    dx = part.unshard([res[r][1] for r in range(world)])
    assert torch.allclose(out, ref_out, atol=1e-5, rtol=1e-5)
    assert torch.allclose(dx, ref_dx, atol=1e-5, rtol=1e-5)
    for r in range(world):                                  # gradients are all-reduced: identical everywhere
        assert torch.allclose(res[r][2], ref_dw, atol=1e-4, rtol=1e-5)
        assert torch.allclose(res[r][3], ref_db, atol=1e-4, rtol=1e-5)


def test_partition_maps():
    part = ND.StridedPartition(10, 4)
    assert [part.n_local(r) for r in range(4)] == [3, 3, 2, 2] and part.n_per == 3
    ids = torch.arange(10)
    tab = part.padded(ids)
    assert tab.tolist() == [0, 3, 6, 9, 1, 4, 7, 10, 2, 5]
    x = torch.arange(10.0).view(-1, 1)
    assert torch.equal(part.unshard([part.shard(x, r) for r in range(4)]), x)
    # every rank's by-dst edges cover exactly the non-loop edges once
    ei = torch.tensor([[0, 1, 2, 3, 4, 4], [1, 1, 3, 2, 4, 0]])
    tot = sum(ND.local_edges(ei, part, r)[0][0].numel() for r in range(4))
    assert tot == 4            # (1,1) and (4,4) are self loops
