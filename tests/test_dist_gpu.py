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


def _case(N, E, F, seed=0):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, : E // 4] = 3
    ei = torch.cat([ei, ei.flip(0)], dim=1)
    x = torch.randn(N, F, generator=g)
    W = torch.randn(F, F, generator=g) / F ** 0.5
    b = torch.randn(F, generator=g)
    go = torch.randn(N, F, generator=g)
    return ei, x, W, b, go


def _worker(rank, world, port, N, E, F, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from npi_gnn_amd import dist as ND
        dev = torch.device("cuda:0")
        ei, x, W, b, go = _case(N, E, F)
        sg = ND.ShardedGraph(ei, N, rank, world, dev)
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


def test_two_ranks_one_gpu_hip_backend(dev):
    world, N, E, F = 2, 4001, 30000, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, E, F, q)) for r in range(world)]
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
    ei, x, W, b, go = _case(N, E, F)
    ref_out, ref_dx, ref_dw, ref_db = R.sage_layer_fwd_bwd(x, ei, W, b, go)
    part = ND.StridedPartition(N, world)
    assert torch.allclose(part.unshard([res[r][0] for r in range(world)]), ref_out, atol=1e-4, rtol=1e-4)
    assert torch.allclose(part.unshard([res[r][1] for r in range(world)]), ref_dx, atol=1e-4, rtol=1e-4)
    for r in range(world):
        assert torch.allclose(res[r][2], ref_dw, atol=1e-2, rtol=1e-3)
        assert torch.allclose(res[r][3], ref_db, atol=1e-2, rtol=1e-3)
