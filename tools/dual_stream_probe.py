"""Does the scoring launch of batch k + 1 overlap the tail of batch k's when the two are issued on two streams?
Each stream owns two Queries buffers (its launches build the queries of ITS next batch) and a score buffer; steps
alternate between the streams.  The launches are issued through pre-bound ctypes calls (one C call per step: the
host must not be the limit).  Prints us per step for 1-4 streams, one- and two-sided."""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from kge_amd import engine, _lib

E, R, d, n = 14541, 237, 512, 512
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
ent = torch.randn(E, d, generator=g).to(dev).bfloat16()
rel = torch.randn(2 * R, d, generator=g).to(dev).bfloat16()
T = engine.Tables("complex", ent, rel)
PITCH = engine.score_pitch(E)
L = _lib.lib()


def batch(seed):
    q = torch.Generator().manual_seed(seed)
    return tuple(torch.randint(hi, (n,), generator=q).to(dev).int() for hi in (E, R, E))


keep = []
for combine in ("sp_", "sp_po"):
    sides = 2 if combine == "sp_po" else 1
    for ns in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        calls = []  # calls[k][parity] = argument tuple
        for k in range(ns):
            q = [engine.Queries(T, combine, n, None), engine.Queries(T, combine, n, None)]
            b = [batch(10 + 2 * k), batch(11 + 2 * k)]
            engine.build_queries(T, combine, *b[0], out=q[0])
            out = torch.empty(n, sides * PITCH, device=dev)
            tc = T.c(None)
            per = []
            for par in range(2):  # score q[par], build q[1 - par] from b[1 - par]
                kp = []
                si, pi, oi = (engine._index(x, dev, kp) for x in b[1 - par])
                nxt = engine.KgeNextQueries(si, pi, oi, n, q[1 - par].buf.data_ptr(), q[1 - par].buf.numel())
                keep.extend([kp, nxt, tc, q, b, out])
                per.append((ctypes.byref(tc), engine._COMBINE[combine], q[par].buf.data_ptr(), n, engine._index(None, dev, kp), E,
                            out.data_ptr(), sides * PITCH, PITCH if sides == 2 else 0, ctypes.byref(nxt),
                            ctypes.c_void_p(streams[k].cuda_stream)))
            calls.append(per)
        torch.cuda.synchronize()
        f = L.kge_score_queries
        cnt = [0]

        def run(K):
            c = cnt[0]
            for i in range(K):
                rc = f(*calls[c % ns][(c // ns) & 1])
                c += 1
            assert rc == 0
            cnt[0] = c

        run(40 * ns)
        torch.cuda.synchronize()
        best, host = 1e9, 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            run(1200)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 1200 * 1e6)
            host = min(host, (t1 - t0) / 1200 * 1e6)
        print(f"{combine} streams={ns}: {best:.2f} us/step (host issue {host:.2f})", flush=True)
