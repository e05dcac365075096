#!/usr/bin/env python3
"""Cycle stamps of gemm_split_ws_kernel (a library built from the product source + tools/micro/gemm_f32_probes.patch with
-DNPI_WS_PROBE=16 -- `PATCH=tools/micro/gemm_f32_probes.patch tools/build_variant.sh stamps gemm_f32.hip -DNPI_WS_PROBE=16` --, loaded through NPI_GNN_LIB): where a producer wave
and a consumer wave spend a launch, for the bf16 x 3 and the fp16 x 2 variant.  usage: NPI_GNN_LIB=... tools/ws_stamps.py [rows]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd import functional as NF
from npi_gnn_amd._lib import load
lib = load()
lib.npi_ws_probe_read.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn(M, 256, device=dev, generator=g); w = torch.randn(256, 256, device=dev, generator=g) / 16
out = torch.empty(M, 256, device=dev)
sc = NF.row_scales(a)
ws3, _ = NF.prepare_weight(w, backward=False); ws2, _ = NF.prepare_weight(w, backward=False, f16=True)
buf = (ctypes.c_ulonglong * 8)()
for name, fn in (("bf16x3", lambda: NF.linear_fwd(a, w, ws=ws3, out=out)), ("fp16x2", lambda: NF.linear_fwd(a, w, ws=ws2, a_scales=sc, out=out))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.npi_ws_probe_read(buf)                      # reset
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    lib.npi_ws_probe_read(buf)
    v = [int(x) for x in buf]
    wgs, ms = 256 * n, e0.elapsed_time(e1) / n
    steps = v[7] / wgs
    print(f"{name}: {ms:.3f} ms; per workgroup and launch: k-steps {steps:.0f}; producer: wait-empty {v[0]/wgs:.0f} wait-loads {v[1]/wgs:.0f} "
          f"split+store {v[2]/wgs:.0f} total {v[3]/wgs:.0f} cycles; consumer: wait-full {v[4]/wgs:.0f} epilogue {v[5]/wgs:.0f} total {v[6]/wgs:.0f} cycles "
          f"-> clock {v[6]/wgs/(ms*1e3):.0f} MHz; per k-step: consumer {(v[6]-v[5])/wgs/steps:.0f} (+ epilogue {v[5]/wgs/(steps/16):.0f} per tile), "
          f"producer split+store {v[2]/wgs/steps:.0f}")
