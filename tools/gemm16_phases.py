#!/usr/bin/env python3
"""Per-phase cycle stamps of gemm16_kernel (bwd_gemm16.hip) at the C2 shape, median over workgroups."""
import ctypes, os, sys
import numpy as np
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
L.kge_debug_gemm16_stamps.argtypes = [ctypes.c_void_p]
L.kge_debug_gemm16_stamps.restype = None
E, d = 14541, 512
names = ["first stage landed", "K loop done", "halves exchanged", "stores issued", "stores acknowledged"]
for rows in (512, 1024):
    mp = (E + 7) // 8 * 8
    g16 = torch.randn(rows, mp, device=dev).to(torch.bfloat16)
    T = torch.randn(E, d, device=dev).to(torch.bfloat16)
    Q = torch.randn(rows, d, device=dev).to(torch.bfloat16)
    scratch = torch.empty(E * d * 4, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for which, x, shape in ((0, T, (rows, d)), (1, Q, (E, d))):
        out = torch.empty(shape, dtype=torch.float32, device=dev)
        stamps = torch.zeros(512 * 8, dtype=torch.int64, device=dev)
        for it in range(3):
            stamps.zero_()
            L.kge_debug_gemm16_stamps(stamps.data_ptr())
            rc = fn(which, 0, d, rows, E, x.data_ptr(), x.stride(0), g16.data_ptr(), mp, out.data_ptr(),
                    scratch.data_ptr(), scratch.numel(), st)
            torch.cuda.synchronize()
            L.kge_debug_gemm16_stamps(None)
            assert rc == 0
        s = stamps.cpu().numpy().reshape(512, 8)
        s = s[s[:, 5] != 0]
        t0 = s[:, 0].min()
        rel = s[:, :6] - s[:, :1]
        print(f"rows={rows} {'dQ' if which == 0 else 'dT'}: {len(s)} workgroups; start skew (max-min of stamp 0) "
              f"{int(s[:, 0].max() - t0)} ticks; last end - first start {int(s[:, 5].max() - t0)} ticks")
        for i, nm in enumerate(names):
            print(f"    {nm:22s} median {int(np.median(rel[:, i + 1])):7d}  max {int(rel[:, i + 1].max()):7d} ticks after the workgroup's start")
print("(ticks: shader clock cycles)")
