#!/usr/bin/env python3
"""Time the two gradient contractions of the bf16 backward at the C2 shape: hand-written (bwd_gemm16.hip)
vs the hipBLASLt path (lib=1).  HIP events over back-to-back launches."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import _lib

dev = "cuda:0"
L = _lib.lib()
fn = L.kge_debug_gemm16
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
               ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
               ctypes.c_void_p]
E, d = 14541, 512
for rows in (512, 1024, 2048):
    mp = (E + 7) // 8 * 8
    g16 = torch.randn(rows, mp, device=dev).to(torch.bfloat16)
    T = torch.randn(E, d, device=dev).to(torch.bfloat16)
    Q = torch.randn(rows, d, device=dev).to(torch.bfloat16)
    scratch = torch.empty(E * d * 4, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for which, x, shape in ((0, T, (rows, d)), (1, Q, (E, d))):
        out = torch.empty(shape, dtype=torch.float32, device=dev)
        for lib in (0, 1):
            def run():
                rc = fn(which, lib, d, rows, E, x.data_ptr(), x.stride(0), g16.data_ptr(), mp, out.data_ptr(),
                        scratch.data_ptr(), scratch.numel(), st)
                assert rc == 0, rc
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 50
            flops = 2.0 * rows * E * d
            print(f"rows={rows} {'dQ' if which == 0 else 'dT'} {['gemm16', 'hipBLASLt', 'gemm16 no DMA', 'gemm16 no compute', 'gemm16 neither'][lib]}: {us:7.1f} us "
                  f"{flops / us / 1e6:7.1f} TF/s", flush=True)
