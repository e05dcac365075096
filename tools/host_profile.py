#!/usr/bin/env python3
"""Where the HOST time of one Net_1 training step goes (the 200-subgraph batch is launch-bound):
cProfile over 100 steps of tools/batch_bench.py's training step, top entries by own time."""
import cProfile
import os
import pstats
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd import pool as NP  # noqa: E402
from batch_bench import make_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    x, ei, batch = make_batch()
    xd, eid, bd = x.to(dev), ei.to(dev), batch.to(dev)
    sd = {f"conv{k}.weight": torch.randn(178 if k == 1 else 128, 128) * 0.1 for k in (1, 2, 3)}
    sd.update({f"conv{k}.bias": torch.zeros(128) for k in (1, 2, 3)})
    sd.update({f"pool{k}.weight": torch.randn(1, 128) for k in (1, 2, 3)})
    sd.update({"lin1.weight": torch.randn(128, 256) * 0.1, "lin1.bias": torch.zeros(128), "lin2.weight": torch.randn(64, 128) * 0.1,
               "lin2.bias": torch.zeros(64), "lin3.weight": torch.randn(2, 64) * 0.1, "lin3.bias": torch.zeros(2)})
    pg = {k: v.to(dev).requires_grad_(True) for k, v in sd.items()}
    og = torch.optim.Adam(pg.values(), lr=1e-3)
    yd = torch.randint(0, 2, (200,)).to(dev)

    def step():
        og.zero_grad(set_to_none=True)
        h, e, bb, acc = xd, eid, bd, None
        for k in (1, 2, 3):
            h = F.relu(npi.sage_conv(h, e, pg[f"conv{k}.weight"], pg[f"conv{k}.bias"]))
            h, e, _, bb, _, _ = NP.topk_pool(h, e, bb, pg[f"pool{k}.weight"], 0.5, num_graphs=200)
            r = NP.global_max_mean_pool(h, bb, 200)
            acc = r if acc is None else acc + r
        z = F.relu(F.linear(acc, pg["lin1.weight"], pg["lin1.bias"]))
        z = F.relu(F.linear(z, pg["lin2.weight"], pg["lin2.bias"]))
        F.nll_loss(F.log_softmax(F.linear(z, pg["lin3.weight"], pg["lin3.bias"]), -1), yd).backward()
        og.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(40)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
