#!/usr/bin/env python3
"""Known-byte-count launches of the segsum kernel, to calibrate rocprofv3's FETCH_SIZE for ITS access
pattern (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE under-reports wide coalesced reads; 'calibrate
on a known byte count in your own access pattern before trusting an absolute').

  case A  graph with no edges: every row reads exactly its own 1 KiB row once (self loop)
          -> reads N*F*4 (+ 8 N index bytes), writes N*F*4; no reuse is possible (1 GiB > 256 MiB L3)
  case B  graph where row i has the single neighbour (i + N/2) mod N: reads 2 N F 4

Run under:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -o cal -- python3 tools/pmc_calibrate.py
The script prints the known byte counts; tools/rocprof_summary.py divides them by the counter."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd import functional as NF  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    N, F = 1_000_000, 256
    x = torch.randn(N, F, device=dev)
    out = {}
    ei = torch.zeros((2, 0), dtype=torch.long, device=dev)
    g = npi.CSRGraph(ei, N)
    for _ in range(3):
        NF.segsum(g, g.by_dst, x, mean=True)            # kernel variant <4,1,false,true,true>
    torch.cuda.synchronize()
    out["A_mean_selfloops_only"] = {"read_bytes": N * F * 4 + N * 8, "write_bytes": N * F * 4, "launches": 3}
    src = torch.arange(N, device=dev)
    dst = (src + N // 2) % N
    g2 = npi.CSRGraph(torch.stack([src, dst]), N)
    for _ in range(3):
        NF.segsum(g2, g2.by_dst, x)                     # kernel variant <4,1,false,false,true>
    torch.cuda.synchronize()
    out["B_sum_one_neighbour"] = {"read_bytes": 2 * N * F * 4 + N * 12, "write_bytes": N * F * 4, "launches": 3}
    print("CALIBRATION " + json.dumps(out))


if __name__ == "__main__":
    main()
