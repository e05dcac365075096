#!/usr/bin/env python3
"""Cycle stamps of the split-bf16 GEMM (build: tools/build_variant.sh wsp16 -DNPI_WS_PROBE=16; run with
NPI_GNN_LIB=npi_gnn_amd/build/variants/lib_wsp16.so).  Prints, per workgroup and k-step, where producer wave 4 and
consumer wave 0 spend their cycles at 1M x 256 x 256."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd import _lib  # noqa: E402
from npi_gnn_amd import functional as NF  # noqa: E402

lib = _lib.load()
rd = lib.npi_ws_probe_read
rd.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
dev = torch.device("cuda:0")
M, K, N = 1 << 20, 256, 256
x = torch.randn(M, K, device=dev)
w = torch.randn(K, N, device=dev) * 0.05
b = torch.randn(N, device=dev)
buf = (ctypes.c_ulonglong * 8)()
for name, fn in (("fwd (bias)", lambda: NF.linear_fwd(x, w, b)), ("bwd_data", lambda: NF.linear_bwd_data(x, w))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    rd(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R = 10
    for _ in range(R):
        fn()
    e1.record()
    torch.cuda.synchronize()
    rd(buf)
    v = [buf[i] / R / 256 for i in range(8)]          # per launch and workgroup
    steps = v[7]
    print(f"{name}: {e0.elapsed_time(e1) / R:.3f} ms per call; per workgroup {steps:.0f} k-steps")
    print(f"  producer wave: total {v[3]:.0f} cyc = {v[3] / steps:.0f} per step: wait empty {v[0] / steps:.0f}, wait loads {v[1] / steps:.0f}, split+store+signal {v[2] / steps:.0f}")
    print(f"  consumer wave: total {v[6]:.0f} cyc = {v[6] / steps:.0f} per step: wait full {v[4] / steps:.0f} per step, epilogue {v[5]:.0f} total = {v[5] / (steps / 16):.0f} per tile")
