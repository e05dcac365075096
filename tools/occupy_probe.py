#!/usr/bin/env python3
"""What a persistent GEMM (one workgroup per CU, static tile walk) does when `k` CUs are held by another kernel -- the situation
of a projection launched while a collective's workgroups are resident.  tools/micro/occupy.hip holds k CUs (64 KB LDS each) on a
side stream; the forward projection of the C4 shape (1M x 256 x 256) is timed alone and beside it.
usage: tools/occupy_probe.py   (build tools/micro/libocc.so first: see occupy.hip)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd import functional as NF
occ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libocc.so"))
occ.occ_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
for M in (1_000_000, 125_000):
    A = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev) / 16; out = torch.empty(M, 256, device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)
    def gemm_ms(k, cycles=3_000_000, reserve=0):
        torch.cuda.synchronize()
        if k:
            occ.occ_launch(k, cycles, sink.data_ptr(), side.cuda_stream)
            torch.cuda._sleep(200_000)                       # let the occupier become resident first
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); NF.linear_fwd(A, W, out=out, reserve_cus=reserve); e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    for _ in range(3): gemm_ms(0)
    print(f"M = {M}: alone {min(gemm_ms(0) for _ in range(5)):.3f} ms; " +
          "; ".join(f"{k} CUs held: {min(gemm_ms(k) for _ in range(3)):.3f} ms" for k in (8, 16, 32)))
    print(f"        with NPI_GEMM_RESERVE_CUS(16): alone {min(gemm_ms(0, reserve=16) for _ in range(5)):.3f} ms; " +
          "; ".join(f"{k} CUs held: {min(gemm_ms(k, reserve=16) for _ in range(3)):.3f} ms" for k in (8, 16)) +
          f";  RESERVE_CUS(32), 32 held: {min(gemm_ms(32, reserve=32) for _ in range(3)):.3f} ms")
