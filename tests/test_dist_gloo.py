"""world_size-2 (and 3) gloo runs of the sharded layer on CPU: the partition, the all-gather
exchange and the gradient all-reduce of npi_gnn_amd.dist, with the local kernels replaced by a torch
stand-in (the HIP backend needs a GPU).  Checked against the single-process oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from npi_gnn_amd import dist as ND
from oracle import ref_conv as R


class TorchBackend:
    """Stand-in for HipBackend: same contract, plain torch ops (a side is its COO entry list)."""

    def make_side(self, key, val, n_rows, n_cols):
        return key, val, n_rows, n_cols

    def row_lengths(self, side):
        return torch.bincount(side[0], minlength=side[2])

    def row_of_entry(self, side):
        return side[0]

    def col_of_entry(self, side):
        return side[1]

    @staticmethod
    def _table(table, table2):
        return table if table2 is None else torch.cat([table, table2])

    def segsum(self, side, table, mean=False, table2=None, w=None, bias=None, out=None):
        key, val, n_rows, n_cols = side
        t = self._table(table, table2)
        assert t.size(0) >= n_cols
        msg = t[val] if w is None else t[val] * w.view(-1, 1)
        res = torch.zeros(n_rows, t.size(1)).index_add_(0, key, msg)
        if mean:
            res = res / torch.bincount(key, minlength=n_rows).clamp(min=1).float().view(-1, 1)
        res = res + bias if bias is not None else res
        if out is not None:
            assert out.shape == res.shape
            out.copy_(res)
            return out
        return res

    def linear_fwd(self, a, w, b, out=None):
        res = a @ w + (b if b is not None else 0)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def linear_bwd_data(self, dc, w, rowscale):
        out = dc @ w.t()
        return out * rowscale.view(-1, 1) if rowscale is not None else out

    def linear_bwd_weight(self, a, dc, want_bias, shared=False):
        return a.t() @ dc, (dc.sum(0) if want_bias else None)

    def colsum(self, x):
        return x.sum(0)

    # ---- GATConv pieces, same index-space contract as HipBackend ----
    def gat_scores(self, h, att2, H, C):
        hv = h.view(-1, H, C)
        return (hv * att2[:, :C]).sum(-1), (hv * att2[:, C:]).sum(-1)

    def gat_stats(self, side, a_row, a_col, H, slope):
        key, val, n_rows, _ = side
        z = F.leaky_relu(a_row[key] + a_col[val], slope)
        m = torch.full((n_rows, H), -3.0e38).scatter_reduce(0, key.view(-1, 1).expand(-1, H), z, "amax", include_self=True)
        empty = torch.bincount(key, minlength=n_rows) == 0
        m = torch.where(empty.view(-1, 1), torch.zeros_like(m), m)           # the kernel's answer for an empty row
        s = torch.zeros(n_rows, H).index_add_(0, key, torch.exp(z - m[key]))
        return m, s

    def _alpha(self, side, a_dst, a_src, m, s, slope, by_source):
        key, val = side[0], side[1]
        tgt, src = (val, key) if by_source else (key, val)
        z = a_dst[tgt] + a_src[src]
        return torch.exp(F.leaky_relu(z, slope) - m[tgt]) / (s[tgt] + 1e-16), z, tgt

    def gat_aggregate(self, side, table, table2, H, C, a_dst, a_src, m, s, slope, by_source, bias=None,
                      g_dst=None, g_src=None, att=None, out=None):
        key, val, n_rows, _ = side
        t = self._table(table, table2)
        alpha, _, _ = self._alpha(side, a_dst, a_src, m, s, slope, by_source)
        msg = t[val].view(-1, H, C) * alpha.view(-1, H, 1)
        out_ = torch.zeros(n_rows, H, C).index_add_(0, key, msg)
        if g_dst is not None:
            out_ = out_ + g_dst.view(-1, H, 1) * att[:, :C].view(1, H, C) + g_src.view(-1, H, 1) * att[:, C:].view(1, H, C)
        res = out_.reshape(n_rows, H * C)
        res = res + bias if bias is not None else res
        if out is not None:
            out.copy_(res)
            return out
        return res

    def gat_rowdot(self, a, b, bias, H, C):
        bb = b - bias if bias is not None else b
        return (a.view(-1, H, C) * bb.view(-1, H, C)).sum(-1)

    def gat_rowdot_colsum(self, a, b, bias, H, C, want_colsum=True):
        return self.gat_rowdot(a, b, bias, H, C), (a.sum(0) if want_colsum else None)

    def gat_edge_grad(self, side, col_feat, col_feat2, row_feat, H, C, a_dst, a_src, m, s, D, slope, swap):
        key, val = side[0], side[1]
        alpha, z, tgt = self._alpha(side, a_dst, a_src, m, s, slope, bool(swap))
        p = (row_feat[key].view(-1, H, C) * self._table(col_feat, col_feat2)[val].view(-1, H, C)).sum(-1)
        return alpha * (p - D[tgt]) * torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope))

    def seg_rowsum(self, side, vals, H, map_=None):
        n = side[0].numel()
        v = vals[map_.long()[:n]] if map_ is not None else vals[:n]
        return torch.zeros(side[2], H).index_add_(0, side[0], v)

    # ---- one head on the direct layout (dist._ShardedGatDirectFn): stand-ins for the round-3 kernels
    def entry_source_index(self, side):
        return torch.arange(side[0].numel())

    def gat_stats_scores(self, side, a_row, a_col, slope):
        key, val = side[0], side[1]
        m, s = self.gat_stats(side, a_row, a_col, 1, slope)
        return m, s, F.leaky_relu(a_row[key] + a_col[val], slope)

    def gat_aggregate_scores(self, side, table, table2, C, scores, m, s, bias=None, out=None):
        key, val, n_rows, _ = side
        t = self._table(table, table2)
        w = torch.exp(scores - m[key]) / (s[key] + 1e-16)
        res = torch.zeros(n_rows, C).index_add_(0, key, t[val] * w)
        res = res + bias if bias is not None else res
        if out is not None:
            out.copy_(res)
            return out
        return res

    def gat_pack(self, a_dst, m, s, D, out=None):
        t = torch.cat([a_dst.reshape(-1, 1), m.reshape(-1, 1), 1.0 / (s.reshape(-1, 1) + 1e-16), D.reshape(-1, 1)], dim=1)
        if out is not None:
            out.copy_(t)
            return out
        return t

    def gat_backward_fused(self, side, dout, dout2, hrow, C, tpack, a_src_rows, slope, out=None, H=1):
        key, val, n_rows, _ = side                                    # key = source row j, val = target column i
        t = self._table(dout, dout2).view(-1, H, C)
        tp = tpack.view(-1, H, 4)[val]                                # [nnz, H, 4]
        z = tp[..., 0] + a_src_rows.view(-1, H)[key]
        alpha = torch.exp(F.leaky_relu(z, slope) - tp[..., 1]) * tp[..., 2]
        res = torch.zeros(n_rows, H, C).index_add_(0, key, t[val] * alpha.unsqueeze(-1)).reshape(n_rows, H * C)
        dot = (t[val] * hrow.view(-1, H, C)[key]).sum(-1)
        dz = alpha * (dot - tp[..., 3]) * torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope))
        if out is not None:
            out.copy_(res)
            res = out
        return res, dz.reshape(-1)

    def gat_rank1_add(self, dh, g_dst, g_src, att2, H, C):
        dh += (g_dst.view(-1, H, 1) * att2[:, :C].view(1, H, C) + g_src.view(-1, H, 1) * att2[:, C:].view(1, H, C)).reshape(dh.shape)
        return dh

    def gat_att_grad(self, h, g_dst, g_src, H, C):
        hv = h.view(-1, H, C)
        return torch.cat([(g_dst.view(-1, H, 1) * hv).sum(0), (g_src.view(-1, H, 1) * hv).sum(0)], dim=1)


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


def _att(Fd, H=1, seed=5):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, H, 2 * (Fd // H), generator=g) * 0.3


def _worker(rank, world, port, N, E, Fd, kind, layer_kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ei, x, W, b, go, hub = _case(N, E, Fd, kind)
        extra = ()
        two = layer_kind.endswith("+2c")                        # the small exchanges on a communicator of their own
        layer_kind = layer_kind[:-3] if two else layer_kind
        if layer_kind == "edges":
            sg = ND.EdgeShardedGraph(ei, N, rank, world, torch.device("cpu"), backend=TorchBackend())
            layer = ND.EdgeShardedSAGELayer(sg, W, b)
            xl = x.clone().requires_grad_(True)                    # replicated input
        else:
            sg = ND.ShardedGraph(ei, N, rank, world, torch.device("cpu"), backend=TorchBackend(), hub_mask=hub,
                                 small_group=dist.new_group() if two else None)
            if layer_kind == "gcnN":                            # F_in > F_out: PyG's literal order (project, then aggregate)
                W, b, go = W[:, : Fd // 2].contiguous(), b[: Fd // 2].contiguous(), go[:, : Fd // 2].contiguous()
            if layer_kind == "sage":
                layer = ND.ShardedSAGELayer(sg, W, b)
            elif layer_kind in ("gcn", "gcnN"):
                layer = ND.ShardedGCNLayer(sg, W, b)
            else:
                H = int(layer_kind[3:])
                layer = ND.ShardedGATLayer(sg, W, _att(Fd, H), b, heads=H)
            xl = sg.shard(x).clone().requires_grad_(True)
        out = layer(xl)
        out.backward(sg.shard(go))
        if layer_kind.startswith("gat"):
            extra = (layer.att.grad,)
        # numpy arrays are pickled by value (torch tensors travel through shared-memory files that
        # vanish when this process exits)
        q.put((rank,) + tuple(t.detach().numpy().copy() for t in (out, xl.grad, layer.weight.grad, layer.bias.grad) + extra))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reference(layer_kind, ei, x, W, b, go, Fd):
    if layer_kind in ("sage", "edges"):
        return R.sage_layer_fwd_bwd(x, ei, W, b, go)
    if layer_kind == "gcnN":
        W, b, go = W[:, : Fd // 2].contiguous(), b[: Fd // 2].contiguous(), go[:, : Fd // 2].contiguous()
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    if layer_kind in ("gcn", "gcnN"):
        out = R.gcn_conv(xr, ei, Wr, br)
        out.backward(go)
        return out.detach(), xr.grad, Wr.grad, br.grad
    H = int(layer_kind[3:])
    att = _att(Fd, H).requires_grad_(True)
    out = R.gat_conv(xr, ei, Wr, att, br, heads=H)
    out.backward(go)
    return out.detach(), xr.grad, Wr.grad, br.grad, att.grad


@pytest.mark.parametrize("world,N,kind,layer_kind", [
    (2, 101, "any", "sage"), (3, 64, "any", "sage"), (2, 203, "bipartite", "sage"), (3, 160, "bipartite", "sage"),
    (2, 120, "auto", "sage"), (8, 333, "bipartite", "sage"), (4, 61, "any", "sage"),
    (2, 203, "bipartite", "gcn"), (3, 64, "any", "gcn"), (8, 333, "bipartite", "gcn"),
    (2, 203, "bipartite", "gat1"), (3, 160, "bipartite", "gat2"), (3, 64, "any", "gat1"), (8, 333, "bipartite", "gat1"),
    (2, 120, "auto", "gat4"),
    (2, 101, "any", "edges"), (3, 160, "bipartite", "edges"), (8, 333, "bipartite", "edges"),
    # fewer hubs than ranks (5 proteins, 8 ranks: three ranks own no hub row at all)
    (8, 40, "bipartite", "sage"), (8, 40, "bipartite", "gcn"), (8, 40, "bipartite", "gat1"),
    (3, 160, "bipartite", "gcnN"), (2, 101, "any", "gcnN"),
    # ShardedGraph(small_group=): per-row scalars, the softmax's MAX and the parameter-gradient sums on a second communicator
    (2, 203, "bipartite", "sage+2c"), (3, 160, "bipartite", "gat1+2c"), (2, 120, "auto", "gat4+2c"), (2, 203, "bipartite", "gcn+2c"),
])
def test_sharded_layer_matches_single_process_oracle(world, N, kind, layer_kind):
    E, Fd = 900, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, E, Fd, kind, layer_kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        got = q.get(timeout=180)
        res[got[0]] = tuple(torch.from_numpy(a) for a in got[1:])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    layer_kind = layer_kind[:-3] if layer_kind.endswith("+2c") else layer_kind
    ei, x, W, b, go, hub = _case(N, E, Fd, kind)
    ref = _reference(layer_kind, ei, x, W, b, go, Fd)
    ref_out, ref_dx, ref_dw, ref_db = ref[:4]
    if layer_kind == "edges":
        per = (N + world - 1) // world
        out = torch.cat([res[r][0] for r in range(world)])
        assert out.shape == ref_out.shape and all(res[r][0].size(0) == min(per, max(N - r * per, 0)) for r in range(world))
        for r in range(world):                              # x is replicated: every rank holds the complete dX
            assert torch.allclose(res[r][1], ref_dx, atol=1e-5, rtol=1e-5)
        dx = res[0][1]
    else:
        part = ND.HubPartition(N, world, hub)
        out = part.unshard([res[r][0] for r in range(world)])
        dx = part.unshard([res[r][1] for r in range(world)])
    assert torch.allclose(out, ref_out, atol=1e-5, rtol=1e-5)
    assert torch.allclose(dx, ref_dx, atol=1e-5, rtol=1e-5)
    for r in range(world):                                  # gradients are all-reduced: identical everywhere
        assert torch.allclose(res[r][2], ref_dw, atol=1e-4, rtol=1e-5)
        assert torch.allclose(res[r][3], ref_db, atol=1e-4, rtol=1e-5)
        if layer_kind.startswith("gat"):
            assert torch.allclose(res[r][4], ref[4], atol=1e-4, rtol=1e-5)


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


def _route_worker(rank, world, port, N, E, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ei, _, _, _, _, hub = _case(N, E, 4, kind)
        Et = ei.size(1)
        mine = ei[:, rank * Et // world: (rank + 1) * Et // world]      # the rank never sees the other columns
        sg = ND.ShardedGraph(mine, N, rank, world, torch.device("cpu"), backend=TorchBackend(), hub_mask=hub, sliced=True)
        es = ND.EdgeShardedGraph(mine, N, rank, world, torch.device("cpu"), backend=TorchBackend(), sliced=True)
        sides = [sg.A, sg.At] + ([sg.B, sg.Bt] if sg.B is not None else [])
        q.put((rank, [(s[0].numpy().copy(), s[1].numpy().copy(), s[2], s[3]) for s in sides],
               sg.inv_cnt.numpy().copy(), sg.cnt_a_hub.numpy().copy(), sg._out_deg.numpy().copy(),
               es.inv_cnt.numpy().copy(), int(es.local_nnz)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,kind", [(2, 203, "bipartite"), (3, 64, "any"), (8, 333, "bipartite"), (4, 120, "auto")])
def test_routed_partition_equals_the_full_list_masks(world, N, kind):
    """dist.route_edges (every rank holds a slice of the edge list; one all-to-all per routed list) gives every rank the
    sides the full-list masks give: the same entries in the same (edge_index) order, the same degree vectors."""
    E = 900
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_route_worker, args=(r, world, port, N, E, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        got = q.get(timeout=180)
        res[got[0]] = got[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ei, _, _, _, _, hub = _case(N, E, 4, kind)
    keep = ei[0] != ei[1]
    in_deg = torch.bincount(ei[1][keep], minlength=N)
    total_edges = 0
    for r in range(world):
        sides, inv_cnt, cnt_a_hub, out_deg, e_inv, e_nnz = res[r]
        local = ND.ShardedGraph(ei, N, r, world, torch.device("cpu"), backend=TorchBackend(), hub_mask=hub)   # no group: masks
        part = local.part
        a, b = ND.local_sides(ei[0], ei[1], part, r)
        at, bt = ND.local_sides(ei[1], ei[0], part, r)
        want = [a, at] + ([b, bt] if local.B is not None else [])
        assert len(sides) == len(want)
        for got_side, ref_side in zip(sides, want):
            assert torch.equal(torch.from_numpy(got_side[0]), ref_side[0]) and torch.equal(torch.from_numpy(got_side[1]), ref_side[1])
            assert got_side[2:] == tuple(ref_side[2:])
        assert torch.equal(torch.from_numpy(inv_cnt), local.inv_cnt) and torch.equal(torch.from_numpy(cnt_a_hub), local.cnt_a_hub)
        assert torch.equal(torch.from_numpy(out_deg), torch.bincount(ei[0][keep], minlength=N))
        per = (N + world - 1) // world
        assert torch.allclose(torch.from_numpy(e_inv), 1.0 / (in_deg[r * per: (r + 1) * per].float() + 1.0))
        total_edges += e_nnz
    assert total_edges == int(keep.sum())                    # the edge slices partition the non-loop edges
    with pytest.raises(ValueError):                          # a slice without a process group cannot be routed
        ND.ShardedGraph(ei[:, :10], N, 0, 2, torch.device("cpu"), backend=TorchBackend(), hub_mask=hub, sliced=True)


def _stack_worker(rank, world, port, N, E, Fd, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ei, x, W, b, go, hub = _case(N, E, Fd, "bipartite")
        Et = ei.size(1)
        sg = ND.ShardedGraph(ei[:, rank * Et // world: (rank + 1) * Et // world], N, rank, world, torch.device("cpu"),
                             backend=TorchBackend(), hub_mask=hub, sliced=True)
        g = torch.Generator().manual_seed(11)
        Ws = [torch.randn(Fd, Fd, generator=g) / Fd ** 0.5 for _ in range(3)]
        bs = [torch.randn(Fd, generator=g) * 0.1 for _ in range(3)]
        if kind == "gat":
            layers = [ND.ShardedGATLayer(sg, Ws[k], _att(Fd, 1, seed=20 + k), bs[k], heads=1) for k in range(3)]
        else:
            layers = [ND.ShardedSAGELayer(sg, Ws[k], bs[k]) for k in range(3)]
        xl = sg.shard(x).clone().requires_grad_(True)
        hcur = xl
        for layer in layers:                                   # a rank's output rows ARE the next layer's input rows
            hcur = torch.relu(layer(hcur))
        hcur.backward(sg.shard(go))
        q.put((rank, hcur.detach().numpy().copy(), xl.grad.numpy().copy(),
               [l.weight.grad.numpy().copy() for l in layers]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "gat"), (4, "sage"), (8, "gat")])
def test_three_sharded_layers_chain_like_the_single_process_stack(world, kind):
    """BASELINE.json configs[4]'s 8-GPU form is a 3-layer GATConv stack: sharded layers compose directly (rows in = rows out,
    nothing is re-partitioned between layers); forward and every gradient of the chain against the single-process oracle."""
    N, E, Fd = 160, 900, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stack_worker, args=(r, world, port, N, E, Fd, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        got = q.get(timeout=180)
        res[got[0]] = got[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ei, x, W, b, go, hub = _case(N, E, Fd, "bipartite")
    g = torch.Generator().manual_seed(11)
    Ws = [torch.randn(Fd, Fd, generator=g).div(Fd ** 0.5).requires_grad_(True) for _ in range(3)]
    bs = [torch.randn(Fd, generator=g) * 0.1 for _ in range(3)]
    xr = x.clone().requires_grad_(True)
    hcur = xr
    for k in range(3):
        hcur = torch.relu(R.gat_conv(hcur, ei, Ws[k], _att(Fd, 1, seed=20 + k), bs[k], heads=1) if kind == "gat"
                          else R.sage_conv(hcur, ei, Ws[k], bs[k]))
    hcur.backward(go)
    part = ND.HubPartition(N, world, hub)
    out = part.unshard([torch.from_numpy(res[r][0]) for r in range(world)])
    dx = part.unshard([torch.from_numpy(res[r][1]) for r in range(world)])
    assert torch.allclose(out, hcur.detach(), atol=1e-5, rtol=1e-4)
    assert torch.allclose(dx, xr.grad, atol=1e-5, rtol=1e-4)
    for r in range(world):
        for k in range(3):
            assert torch.allclose(torch.from_numpy(res[r][2][k]), Ws[k].grad, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("world,H,Fd", [(2, 2, 64), (8, 4, 128), (3, 8, 256)])
def test_sharded_gat_several_heads_on_the_direct_layout(world, H, Fd):
    """2 / 4 / 8 heads of 32 channels: the shapes the fused multi-head backward serves, through dist._ShardedGatDirectFn
    (hub softmax merged relative to the all-reduced maximum, per-head packed target scalars, dz [nnz, H] through the local
    index maps) -- outputs and every gradient against the single-process oracle."""
    N, E = 160, 900
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    kind = f"gat{H}"
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, E, Fd, "bipartite", kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        got = q.get(timeout=180)
        res[got[0]] = tuple(torch.from_numpy(a) for a in got[1:])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ei, x, W, b, go, hub = _case(N, E, Fd, "bipartite")
    ref = _reference(kind, ei, x, W, b, go, Fd)
    part = ND.HubPartition(N, world, hub)
    out = part.unshard([res[r][0] for r in range(world)])
    dx = part.unshard([res[r][1] for r in range(world)])
    assert torch.allclose(out, ref[0], atol=1e-5, rtol=1e-4)
    assert torch.allclose(dx, ref[1], atol=1e-5, rtol=1e-4)
    for r in range(world):
        assert torch.allclose(res[r][2], ref[2], atol=1e-4, rtol=1e-4)
        assert torch.allclose(res[r][3], ref[3], atol=1e-4, rtol=1e-4)
        assert torch.allclose(res[r][4], ref[4], atol=1e-4, rtol=1e-4)
    # ... and it is the direct path that ran: same graph, one process, the Function's name
    sg = ND.ShardedGraph(ei, N, 0, 1, torch.device("cpu"), backend=TorchBackend(), hub_mask=hub)
    layer = ND.ShardedGATLayer(sg, W, _att(Fd, H), b, heads=H)
    assert type(layer(sg.shard(x).clone().requires_grad_(True)).grad_fn).__name__ == "_ShardedGatDirectFnBackward"
