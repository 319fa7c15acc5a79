"""The edges of the headline's group launch (pairs_bf16_v8_kernel, 8 two-sided batches, both query modes): per workgroup
the cycle stamps of its start and of its last store (kge_debug_v6_stamps), relative to the first workgroup's start --
how far the workgroups' ends are spread, i.e. how much of a launch some compute units stand idle (DESIGN 13.10:
SQ_BUSY_CU_CYCLES = 0.85 of the launch)."""
import ctypes, sys, torch
sys.path.insert(0, "/root/repo")
from kge_amd import engine, _lib
dev = torch.device("cuda", 0)
E, R, D, n, L = 14541, 237, 512, 512, 8
P = engine.score_pitch(E)
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
L_ = _lib.lib()
L_.kge_debug_v6_stamps.restype = None
L_.kge_debug_v6_stamps.argtypes = [ctypes.c_void_p]
for flags, tag in ((engine.FLAG_SPLIT_QUERY, "split"), (0, "single")):
    T = engine.Tables("complex", ent, rel, flags=flags)
    grp = torch.stack([torch.randint(hi, (n * L,), generator=g) for hi in (E, R, E)], 1).to(dev)
    q = engine.build_queries_group(T, "sp_po", grp, n, L, flags=flags)
    gbuf = torch.empty(L, n, 2 * P, device=dev)
    gout = gbuf.view(L, n, 2, P)[:, :, :, :E]
    for _ in range(20):
        engine.score_queries_group(T, q, gout)
    for rep in range(3):
        st = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        L_.kge_debug_v6_stamps(ctypes.c_void_p(st.data_ptr()))
        engine.score_queries_group(T, q, gout)
        torch.cuda.synchronize()
        L_.kge_debug_v6_stamps(None)
        v = st.view(4096, 64).cpu()
        v = v[(v[:, 0] != 0) & (v[:, 34] != 0)]
        # (the cycle counters of the eight XCDs are not aligned with each other: everything relative to the first start
        # inside the workgroup's own XCD -- blockIdx % 8)
        xcd = torch.arange(v.shape[0]) % 8
        t0 = torch.stack([v[xcd == x, 0].min() for x in range(8)])[xcd]
        start = (v[:, 0] - t0).double()
        end = (v[:, 34] - t0).double()
        dur = end - start
        span = torch.stack([end[xcd == x].max() for x in range(8)])
        print(f"{tag} rep {rep}: {v.shape[0]} workgroups; start behind the XCD's first: median {start.median():.0f} max {start.max():.0f}; "
              f"start -> last store issued: min {dur.min():.0f} median {dur.median():.0f} max {dur.max():.0f} cycles; the XCDs' spans (first "
              f"start -> last store) {[round(float(x)) for x in span]}; mean idle behind a workgroup's own last store "
              f"{float(((span[xcd] - end) / span[xcd]).mean()):.3f} of its XCD's span", flush=True)
