"""The hand-written gradient contractions of the bf16 ComplEx / DistMult backward (bwd_gemm16.hip:
gemm16_kernel<false> = dQ = G16 * T with split-K, gemm16_kernel<true> = dT = G16^T * Q16) against
float64 products of the same bf16 operands, and against the hipBLASLt path they replace (which stays as
the checker: `lib=1`).  Random (transpose-detecting) operands; ragged shapes on every axis: rows not a
multiple of the 128-row tile or the 64-step, m not a multiple of 8 / 64 / 128, G16 pitch > m."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [  # rows, m, d
    (512, 14541, 512),     # C2, one direction
    (1024, 14541, 512),    # C2, both directions in one product
    (1000, 4097, 256),
    (130, 77, 256),
    (1, 5, 256),
    (64, 64, 512),
    (333, 1000, 768),
]


def _call(which, lib, d, rows, m, x, g16, mp, scratch_mb=32):
    from kge_amd import _lib
    L = _lib.lib()
    fn = L.kge_debug_gemm16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                   ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_int64, ctypes.c_void_p]
    out = torch.full((rows if which == 0 else m, d), float("nan"), dtype=torch.float32, device=DEV)
    scratch = torch.empty(scratch_mb << 20, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    rc = fn(which, lib, d, rows, m, x.data_ptr(), x.stride(0), g16.data_ptr(), mp, out.data_ptr(),
            scratch.data_ptr() if scratch_mb else None, scratch.numel(), st)
    torch.cuda.synchronize()
    return rc, out


def _operands(rows, m, d, seed):
    g = torch.Generator().manual_seed(seed)
    mp = (m + 7) // 8 * 8 + 8 * (seed % 3)                        # pitch >= m, sometimes with extra columns
    g16 = torch.randn(rows, mp, generator=g).to(torch.bfloat16)
    g16[:, m:(m + 7) // 8 * 8] = 0   # the contract: zero up to the next multiple of 8; garbage (on purpose) beyond
    T = (torch.randn(m, d, generator=g) * 0.5).to(torch.bfloat16)
    Q = (torch.randn(rows, d, generator=g) * 0.5).to(torch.bfloat16)
    return mp, g16.to(DEV), T.to(DEV), Q.to(DEV)


@pytest.mark.parametrize("rows,m,d", CASES)
def test_dq_and_dt_against_float64(rows, m, d):
    mp, g16, T, Q = _operands(rows, m, d, rows + m)
    G = g16[:, :m].double()
    want_dq, want_dt = G @ T.double(), G.t() @ Q.double()
    for scratch_mb in (32, 0):                                   # split-K partials / no scratch: one split
        rc, dq = _call(0, 0, d, rows, m, T, g16, mp, scratch_mb)
        assert rc == 0
        err = float((dq.double() - want_dq).abs().max() / want_dq.abs().max())
        assert err <= 2e-6, ("dQ", scratch_mb, err)
    rc, dt = _call(1, 0, d, rows, m, Q, g16, mp)
    assert rc == 0
    err = float((dt.double() - want_dt).abs().max() / want_dt.abs().max())
    assert err <= 2e-6, ("dT", err)


def test_against_the_library_path_it_replaces():
    rows, m, d = 1024, 14541, 512
    mp, g16, T, Q = _operands(rows, m, d, 3)
    g16[:, m:] = 0                                               # the library reads the pitch as given
    for which, x in ((0, T), (1, Q)):
        rc_h, ours = _call(which, 0, d, rows, m, x, g16, mp)
        rc_l, lib = _call(which, 1, d, rows, m, x, g16, mp)
        assert rc_h == 0 and rc_l == 0
        scale = float(lib.abs().max())
        assert float((ours - lib).abs().max()) <= 1e-5 * scale
