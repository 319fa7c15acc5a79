"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/kge_amd.h declares (no compute calls without a GPU); argument
validation that does not touch the device; the product path refuses CPU tensors."""
import os
import re

import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from kge_amd import _lib
    _lib.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "kge_amd.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(kge_\w+)\s*\(", header, flags=re.M))
    from kge_amd import _lib
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.kge_abi_version() == 1
    assert lib.kge_status_string(0) == b"ok"
    assert lib.kge_status_string(-1) == b"invalid argument"


def test_debug_exports_are_exactly_the_declared_ones(lib):
    """Every exported kge_* symbol is declared: the boundary in include/kge_amd.h, the measurement hooks
    (kge_debug_*) in include/kge_amd_debug.h -- nothing rides along undeclared."""
    import subprocess
    from kge_amd import _lib
    dbg = open(os.path.join(ROOT, "include", "kge_amd_debug.h")).read()
    declared_dbg = set(re.findall(r"^(?:int|int64_t|void|double)\s+(kge_debug_\w+)\s*\(", dbg, flags=re.M))
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("kge_")}
    assert exported == set(_lib.PROTOTYPES) | declared_dbg, exported ^ (set(_lib.PROTOTYPES) | declared_dbg)
    assert declared_dbg and all(n.startswith("kge_debug_") for n in declared_dbg)


def test_kernel_selection_reads_no_environment_variable(lib):
    """The library's measurement switches are process-local state behind kge_debug_set_switch (round 6; until then ~25
    getenv() calls read on every launch): every name of kge_amd/csrc/switches.hpp is settable, unset by default, an
    unknown name is refused -- and the only getenv left in the sources is KGE_ROCTX (the roctx ranges)."""
    import glob
    from kge_amd import _lib
    names = re.findall(r"^\s+SW_(\w+?)(?: = 0)?,", open(os.path.join(ROOT, "kge_amd", "csrc", "switches.hpp")).read(), flags=re.M)
    assert len(names) >= 19 and "V8" in names and "CE_V8" in names
    for nm in names:
        assert _lib.get_switch(nm) is None, nm
        with _lib.switches(**{nm: 1}):
            assert _lib.get_switch(nm) == 1 and _lib.get_switch("KGE_" + nm) == 1
        assert _lib.get_switch(nm) is None
    assert lib.kge_debug_set_switch(b"NO_SUCH_SWITCH", 1) == -1
    envs = []
    for f in glob.glob(os.path.join(ROOT, "kge_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "kge_amd", "csrc", "*.cpp")):
        for line in open(f):
            code = line.split("//")[0]
            envs += re.findall(r'getenv\("(\w+)"\)', code)
    assert sorted(set(envs)) == ["KGE_ROCTX"], envs


def test_struct_layout_matches_header():
    import ctypes
    from kge_amd._lib import KgeIndex, KgeTables
    assert ctypes.sizeof(KgeTables) == 80   # 2 ptr + 2 i32 + 6 i64 + f32 + i32
    assert ctypes.sizeof(KgeIndex) == 24


def test_invalid_arguments_are_rejected_without_a_device(lib):
    import ctypes
    from kge_amd._lib import KgeIndex, KgeTables
    ix = KgeIndex(None, 1, 0, 1)
    assert lib.kge_score_spo(None, ix, ix, ix, 4, None, None) == -1
    t = KgeTables(None, None, 0, 7, 10, 3, 32, 32, 32, 32, 1.0, 0)   # bad scorer
    assert lib.kge_score_spo(ctypes.byref(t), ix, ix, ix, 4, None, None) == -1
    t = KgeTables(None, None, 0, 0, 10, 3, 33, 33, 33, 33, 1.0, 0)   # ComplEx with odd dim
    assert lib.kge_score_sp(ctypes.byref(t), ix, ix, 4, ix, 10, None, 10, None, 0, None) == -1
    t = KgeTables(None, None, 0, 3, 10, 3, 32, 32, 32, 32, 1.0, 0)   # RotatE rel_dim != dim/2
    assert lib.kge_score_sp(ctypes.byref(t), ix, ix, 4, ix, 10, None, 10, None, 0, None) == -1
    t = KgeTables(None, None, 1, 0, 10, 3, 512, 512, 512, 512, 1.0, 0)  # bf16 ComplEx d=512
    assert lib.kge_score_workspace_bytes(ctypes.byref(t), 512) == 2 * 512 * 512 * 2 + 512 * 8 * 8 + 256
    assert lib.kge_score_workspace_bytes(ctypes.byref(t), 33) == 2 * 128 * 512 * 2 + 512 * 8 * 8 + 256
    t = KgeTables(None, None, 0, 0, 10, 3, 512, 512, 512, 512, 1.0, 0)  # f32: no workspace
    assert lib.kge_score_workspace_bytes(ctypes.byref(t), 512) == 0
    assert lib.kge_rank_counts(None, 3, 2, 5, None, None, None, 0, None, 1e-5, 1e-4, None, None, None) == -1


def test_product_path_has_no_cpu_fallback(lib):
    from kge_amd import engine
    ent = torch.randn(10, 8)
    rel = torch.randn(3, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        engine.Tables("distmult", ent, rel)
    with pytest.raises(RuntimeError, match="no CPU path"):
        engine.rank_counts(torch.zeros(2, 3), torch.zeros(2))


def test_no_oracle_import_in_product_code():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "kge_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                for pat in (r"#include\s*[<\"].*oracle", r"^\s*(import|from)\s+oracle", r"libkge_oracle",
                            r"ref_harness", r"sys\.path.*oracle"):
                    assert not re.search(pat, src, flags=re.M), (f, pat)


def test_new_entries_validate_arguments_without_a_device(lib):
    """Fused-loss, evaluation and optimizer entry points: argument checks that return before any
    launch (no GPU here)."""
    import ctypes
    from kge_amd._lib import KgeIndex, KgeTables
    ix = KgeIndex(None, 1, 0, 1)
    good = KgeIndex(ctypes.c_void_p(16), 1, 0, 1)  # never dereferenced on these paths
    f32 = KgeTables(ctypes.c_void_p(16), ctypes.c_void_p(16), 0, 0, 10, 3, 512, 512, 512, 512, 1.0, 0)
    bf16 = KgeTables(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 0, 10, 3, 512, 512, 512, 512, 1.0, 0)
    transe = KgeTables(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 2, 10, 3, 512, 512, 512, 512, 1.0, 0)
    odd = KgeTables(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 0, 10, 3, 320, 320, 320, 320, 1.0, 0)
    # workspace queries: only bf16 ComplEx / DistMult with dim in {128, 256, 512}
    assert lib.kge_ce_workspace_bytes(ctypes.byref(bf16), 512) > 0
    assert lib.kge_ce_sp_po_workspace_bytes(ctypes.byref(bf16), 512) >= lib.kge_ce_workspace_bytes(ctypes.byref(bf16), 512)
    for t in (f32, transe, odd):
        assert lib.kge_ce_workspace_bytes(ctypes.byref(t), 512) == 0
        assert lib.kge_ce_sp_po_workspace_bytes(ctypes.byref(t), 512) == 0
    # unsupported tables / bad direction / missing index vectors
    assert lib.kge_ce_fwd(ctypes.byref(f32), 1, good, good, good, 4, None, None, None, 0, None) == -2
    assert lib.kge_ce_fwd(ctypes.byref(bf16), 0, good, good, good, 4, None, None, None, 0, None) == -1
    assert lib.kge_ce_fwd(ctypes.byref(bf16), 1, ix, good, good, 4, None, None, None, 0, None) == -1
    assert lib.kge_ce_fwd(ctypes.byref(bf16), 1, good, good, good, 4, None, None, None, 0, None) == -1  # no outputs
    assert lib.kge_ce_sp_po_fwd(ctypes.byref(transe), good, good, good, 4, None, None, None, 0, None) == -2
    assert lib.kge_kl_fwd(ctypes.byref(bf16), 1, good, good, 4, None, None, None, None, None, 0, None) == -1  # no labels
    assert lib.kge_bce_fwd(ctypes.byref(f32), 1, good, good, 4, ctypes.c_void_p(16), ctypes.c_void_p(16), 0.0,
                           ctypes.c_void_p(16), None, 0, None) == -2
    # empty batches are fine without a workspace
    assert lib.kge_ce_fwd(ctypes.byref(bf16), 1, ix, ix, ix, 0, None, None, None, 0, None) == 0
    assert lib.kge_ce_sp_po_fwd(ctypes.byref(bf16), ix, ix, ix, 0, None, None, None, 0, None) == 0
    # evaluation entries
    assert lib.kge_filter_lookup(None, 5, None, good, good, 7, 3, ctypes.c_void_p(16), ctypes.c_void_p(16), None) == -1
    assert lib.kge_filter_lookup(None, 0, None, ix, ix, 7, 0, None, None, None) == 0
    assert lib.kge_rank_counts_multi(None, 3, 2, 5, None, 0, None, None, None, 0, None, 1e-5, 1e-4, None, None, None) == -1
    assert lib.kge_rank_counts_multi(ctypes.c_void_p(16), 5, 2, 5, ctypes.c_void_p(16), 9, None, None, None, 0, None,
                                     1e-5, 1e-4, ctypes.c_void_p(16), ctypes.c_void_p(16), None) == -2  # > KGE_MAX_FILTERS
    # score + rank in one kernel: size arithmetic, argument checks, what it declines (all before any launch)
    # word-major filter bits: 2 sides x K sets x (n up to whole 64-row lines) x (m up to whole 64-column words, as
    # 32-bit words) x 4 bytes, + 16 (the last row's 8-byte load when there is ONE set); rows up to 64 cost nothing
    assert lib.kge_score_rank_bits_bytes(512, 14541, 2) == 2 * 2 * 512 * (228 * 2) * 4 + 16
    assert lib.kge_score_rank_bits_bytes(500, 14541, 2) == lib.kge_score_rank_bits_bytes(512, 14541, 2)
    assert lib.kge_score_rank_bits_bytes(512, 14541, 1) == 2 * 1 * 512 * (228 * 2) * 4 + 16
    assert lib.kge_score_rank_bits_bytes(512, 14541, 0) == 0 and lib.kge_score_rank_bits_bytes(0, 64, 2) == 0
    P = ctypes.c_void_p(16)
    rank_args = lambda t_, n, cb, m, k, lists=(None,) * 6: (ctypes.byref(t_), good, good, good, n, cb, m, P, P, k,
                                                            *lists, 1e-5, 1e-4, P, P, P, P, max(n, 1), None, 0, P, 1 << 20, None)
    assert lib.kge_score_rank_sp_po(*rank_args(bf16, 4, 0, bf16.num_ent + 1, 0)) == -1   # slice beyond the table
    assert lib.kge_score_rank_sp_po(*rank_args(bf16, 4, 0, bf16.num_ent, 3)) == -2        # > 2 filter sets
    assert lib.kge_score_rank_sp_po(*rank_args(bf16, 4, 0, bf16.num_ent, 1)) == -1        # filter lists missing
    # (float32 tables and TransE / RotatE count in the exact kernels since round 3: no decline to test without a device)
    assert lib.kge_score_rank_sp_po(*rank_args(odd, 4, 0, odd.num_ent, 0)) == -2           # bf16 at a dim without a counting kernel
    # (split queries at dim 256 / 512 are counted by pairs_bf16_v8_rank_kernel since round 4; at another dim they are not)
    split = KgeTables(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 0, 10, 3, 128, 128, 128, 128, 1.0, 32)
    assert lib.kge_score_rank_sp_po(*rank_args(split, 4, 0, split.num_ent, 0)) == -2       # split queries, dim 128
    # one evaluation batch in four launches: size arithmetic and argument checks
    assert lib.kge_eval_batch_scratch_bytes(ctypes.byref(bf16), 512, 2) >= 2 * 2 * 2 * 512 * 8 + 2 * 512 * 8 + 512 * 4 * 512 * 4
    assert lib.kge_eval_batch_scratch_bytes(ctypes.byref(bf16), 0, 2) == 0
    assert lib.kge_eval_batch_scratch_bytes(ctypes.byref(bf16), 512, 3) == 0
    ev_args = lambda t_, n, k, pol=0, counts=P, scratch=P, sb=1 << 30: (
        ctypes.byref(t_), good, good, good, n, k, None, 1e-5, 1e-4, pol, counts, P, 16, None, None, None, 0, scratch,
        sb, P, 1 << 20, None)
    assert lib.kge_eval_batch(*ev_args(bf16, 4, 3)) == -2                  # > 2 filter sets
    assert lib.kge_eval_batch(*ev_args(bf16, 4, 0, pol=9)) == -1           # unknown tie policy
    assert lib.kge_eval_batch(*ev_args(bf16, 4, 1)) == -1                  # filters missing
    assert lib.kge_eval_batch(*ev_args(bf16, 4, 0, counts=None)) == -1     # no counters
    assert lib.kge_eval_batch(*ev_args(bf16, 4, 0, sb=64)) == -5           # scratch too small
    assert lib.kge_eval_batch(*ev_args(bf16, 0, 0)) == 0                   # empty batch
    assert lib.kge_score_rank_sp_po(*rank_args(bf16, 0, 0, bf16.num_ent, 0)) == 0          # empty batch
    # band-and-rescore: the band needs split queries and a row-norm bound; NULL band = the plain entry points
    from kge_amd._lib import KgeRankBand
    assert ctypes.sizeof(KgeRankBand) == 32 and KgeRankBand.list_bytes.offset == 16 and KgeRankBand.status.offset == 24
    assert lib.kge_rank_band_list_bytes(0) == 0 and lib.kge_rank_band_list_bytes(512) == 64 * (4 + 32) * 4096
    assert lib.kge_rank_band_list_bytes(100000) == 64 * (2 * 391 + 32) * 4096
    band = KgeRankBand(16, 16, 1 << 30, 16)
    split512 = KgeTables(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 0, 10, 3, 512, 512, 512, 512, 1.0, 32)
    assert lib.kge_score_rank_sp_po_band(*rank_args(bf16, 4, 0, bf16.num_ent, 0), ctypes.byref(band)) == -1   # no split flag
    assert lib.kge_score_rank_sp_po_band(*rank_args(split512, 4, 0, 10, 0), ctypes.byref(KgeRankBand(None, 16, 1 << 30, 16))) == -1
    assert lib.kge_score_rank_sp_po_band(*rank_args(split512, 4, 0, 10, 0), ctypes.byref(KgeRankBand(16, 16, 1 << 30, 18))) == -1  # misaligned status
    assert lib.kge_score_rank_sp_po_band(*rank_args(split512, 4, 0, 10, 0), ctypes.byref(KgeRankBand(16, 24, 1 << 30, 16))) == -1  # misaligned list
    assert lib.kge_score_rank_sp_po_band(*rank_args(split512, 4, 0, 10, 0), ctypes.byref(KgeRankBand(16, 16, 4096, 16))) == -5    # list too small
    assert lib.kge_score_rank_sp_po_band(*rank_args(split, 4, 0, split.num_ent, 0), ctypes.byref(band)) == -2   # dim 128
    assert lib.kge_score_rank_sp_po_band(*rank_args(bf16, 0, 0, bf16.num_ent, 0), ctypes.byref(band)) == 0      # empty batch
    assert lib.kge_score_rank_sp_po_band(*rank_args(bf16, 4, 0, bf16.num_ent, 3), None) == -2                   # = the plain entry
    assert lib.kge_eval_batch_band(*ev_args(bf16, 4, 0, pol=9), ctypes.byref(band)) == -1
    assert lib.kge_eval_batch_band(*ev_args(bf16, 0, 0), ctypes.byref(band)) == 0
    assert lib.kge_table_max_row_norm(ctypes.byref(bf16), 0, bf16.num_ent, None, None) == -1               # no output
    assert lib.kge_table_max_row_norm(ctypes.byref(bf16), 1, bf16.num_ent, P, None) == -1                  # rows beyond the table
    f32t = KgeTables(ctypes.c_void_p(16), ctypes.c_void_p(16), 0, 0, 10, 3, 512, 512, 512, 512, 1.0, 0)
    assert lib.kge_table_max_row_norm(ctypes.byref(f32t), 0, 10, P, None) == -2                            # bf16 tables only
    assert lib.kge_rank_hist(None, None, 3, 4, 7, None, 10, 10, None, None) == -1   # unknown tie policy
    assert lib.kge_rank_hist(None, None, 0, 0, 0, None, 10, 10, None, None) == 0
    # optimizer step: null / misaligned arrays
    assert lib.kge_adagrad_step(None, None, None, 8, -0.1, 0.0, 1e-10, None, None) == -1
    assert lib.kge_adagrad_step(ctypes.c_void_p(20), ctypes.c_void_p(16), ctypes.c_void_p(16), 8, -0.1, 0.0, 1e-10,
                                None, None) == -1
    assert lib.kge_adagrad_step(None, None, None, 0, -0.1, 0.0, 1e-10, None, None) == 0
    from kge_amd._lib import KgeAdagradSeg
    seg = lambda p, cnt: KgeAdagradSeg(p, 16, 16, None, cnt, -0.1, 0.0, 1e-10)
    assert lib.kge_adagrad_step_multi(None, 0, None) == 0
    assert lib.kge_adagrad_step_multi(None, 1, None) == -1
    assert lib.kge_adagrad_step_multi((KgeAdagradSeg * 9)(*[seg(None, 0)] * 9), 9, None) == -1   # at most 8 segments
    assert lib.kge_adagrad_step_multi((KgeAdagradSeg * 2)(seg(None, 0), seg(20, 8)), 2, None) == -1  # misaligned
    assert lib.kge_adagrad_step_multi((KgeAdagradSeg * 2)(seg(None, 0), seg(None, 0)), 2, None) == 0  # nothing to do
    # ... with folded penalty terms: kind / exponent / accumulator / complex-row geometry are checked before any launch
    from kge_amd._lib import KgePenaltySeg
    import ctypes as _ct
    assert _ct.sizeof(KgePenaltySeg) == 32 and KgePenaltySeg.row_dim.offset == 16 and KgePenaltySeg.value.offset == 24
    one = lambda pen: lib.kge_adagrad_step_multi_penalty((KgeAdagradSeg * 1)(seg(16, 64)), (KgePenaltySeg * 1)(pen), 1, None)
    assert lib.kge_adagrad_step_multi_penalty(None, None, 0, None) == 0
    assert one(KgePenaltySeg(3, 2, 0.1, 0, 8)) == -1          # unknown kind
    assert one(KgePenaltySeg(1, 2, 0.1, 0, None)) == -1       # no accumulator
    assert one(KgePenaltySeg(1, 2, 0.1, 0, 12)) == -1         # misaligned accumulator
    assert one(KgePenaltySeg(1, 4, 0.1, 0, 8)) == -2          # exponent outside 1..3: unsupported
    assert one(KgePenaltySeg(2, 3, 0.1, 12, 8)) == -1         # complex rows: row_dim % 8 != 0
    assert one(KgePenaltySeg(2, 3, 0.1, 48, 8)) == -1         # ... count not a multiple of row_dim
    # both query types of a KvsAll batch in one backward: unsupported tables, unknown loss, missing pieces
    from kge_amd._lib import KgeLabelQueries
    lq = lambda n, lse=16: KgeLabelQueries(good, good, n, 16, 16, lse, None, 1.0, None)
    assert lib.kge_multilabel2_workspace_bytes(ctypes.byref(bf16), 512, 300) > lib.kge_ce_workspace_bytes(ctypes.byref(bf16), 512)
    assert lib.kge_multilabel2_workspace_bytes(ctypes.byref(transe), 512, 300) == 0
    two = lambda t, loss, a, b, ge=ctypes.c_void_p(16): lib.kge_multilabel2_bwd_accum(
        ctypes.byref(t), loss, 0.0, ctypes.byref(a), ctypes.byref(b), ge, ctypes.c_void_p(16), None, 0, None)
    assert two(transe, 0, lq(4), lq(4)) == -2
    assert two(bf16, 2, lq(4), lq(4)) == -1                  # neither kl nor bce
    assert two(bf16, 0, lq(4, None), lq(4)) == -1            # kl without its lse
    assert two(bf16, 0, lq(4), lq(4), None) == -1            # no gradient output
    assert two(bf16, 1, lq(4, None), lq(4, None)) == -5      # (bce needs no lse; no workspace: KGE_ERR_WORKSPACE)
    # the summed forms of the two-sided loss: same checks as kge_ce_sp_po_fwd, and the sum's address is required
    assert lib.kge_ce_sp_po_fwd_sum(ctypes.byref(transe), good, good, good, 4, None, None, None, 1.0,
                                    ctypes.c_void_p(16), None, 0, None) == -2
    assert lib.kge_ce_sp_po_fwd_sum(ctypes.byref(bf16), good, good, good, 4, ctypes.c_void_p(16), ctypes.c_void_p(16),
                                    None, 1.0, None, None, 0, None) == -1
    assert lib.kge_ce_sp_po_bwd_accum_sum(ctypes.byref(bf16), good, good, good, 4, ctypes.c_void_p(16), None, None,
                                          1.0, None, None, None, 0, None) == -1


def test_torch_extension_builds_and_binds_the_c_abi(lib):
    """kge_amd._C (csrc/torch_ext.cpp): the PyTorch-ROCm C++ extension north_star names -- builds in-tree against this
    interpreter's torch, links libkge_amd.so, refuses CPU tensors (no compute without a GPU here)."""
    from kge_amd import _lib
    assert os.path.exists(_lib.build_extension())
    ext = _lib.ext()
    assert ext.abi_version() == lib.kge_abi_version()
    for name in ("score_spo", "score_pairs", "queries_bytes", "build_queries_group", "score_queries_group"):
        assert callable(getattr(ext, name))
    ent, rel, ix = torch.randn(10, 8), torch.randn(3, 8), torch.zeros(4, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ext.score_spo(ent, rel, 1, 1.0, 0, ix, ix, ix)


def test_the_package_reads_only_the_listed_environment_variables():
    """VERDICT r5 (weak 7, 8; next 6): nothing in kge_amd/ -- Python or the C library -- selects a kernel, a backend or
    a code path from an environment variable.  What is read: torchrun's rendezvous variables (the sharded jobs' process
    group), KGE_AMD_BINDING (ctypes instead of the torch extension: the same C entry points) and, in the library,
    KGE_ROCTX (roctx ranges).  Measurement switches are kge_debug_set_switch (csrc/switches.hpp), class attributes and
    config options; tools/ map their old variable names onto those themselves."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kge_amd")
    allowed = {"WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KGE_AMD_BINDING", "KGE_ROCTX"}
    pat = re.compile(r"(?:environ(?:\.get)?\s*[\[(]|getenv\s*\(|[\"']\s+(?:not\s+)?in\s+os\.environ)")
    name = re.compile(r"[\"']([A-Z][A-Z0-9_]+)[\"']")
    seen = set()
    for dirpath, _, files in os.walk(root):
        for f in files:
            if not f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                continue
            for line in open(os.path.join(dirpath, f), errors="replace"):
                code = line.split("#")[0] if f.endswith(".py") else line.split("//")[0]
                if "os.environ" in code or "getenv" in code:  # ("environment" in a docstring is not a read)
                    assert pat.search(code) or "os.environ" in code, (f, line)
                    found = set(name.findall(code))
                    assert found, f"{f}: an environment read without a literal name: {line.strip()}"
                    seen |= found
    assert seen <= allowed, f"environment variables read by the package outside the documented set: {sorted(seen - allowed)}"
    assert "KGE_ROCTX" in seen and "WORLD_SIZE" in seen
