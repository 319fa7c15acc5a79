"""ctypes binding of libkge_amd.so (C ABI: include/kge_amd.h).

The product path has NO fallback: if the library is missing or cannot be loaded this
module raises, and every op in kge_amd.engine raises on non-GPU tensors.  torch is
imported first so that the HIP runtime already mapped by torch (torch/lib/libamdhip64.so,
SONAME libamdhip64.so.7) is the one our library binds to -- torch's streams and device
pointers must belong to the same runtime instance.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must be loaded before libkge_amd.so, see above)

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libkge_amd.so")

KGE_OK = 0
KGE_ERR_UNSUPPORTED = -2
COMPLEX, DISTMULT, TRANSE, ROTATE = 0, 1, 2, 3
SCORERS = {"complex": COMPLEX, "distmult": DISTMULT, "transe": TRANSE, "rotate": ROTATE}
F32, BF16 = 0, 1
I32, I64 = 0, 1
SPO, SP_, PO_ = 0, 1, 2
FLAG_EXACT, FLAG_NO_MFMA, FLAG_BF16_V3 = 1, 2, 16
FLAG_SPLIT_QUERY = 32
SP_PO = 3

c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p


class KgeTables(ctypes.Structure):
    _fields_ = [
        ("ent", c_vp), ("rel", c_vp), ("dtype", ctypes.c_int32), ("scorer", ctypes.c_int32),
        ("num_ent", c_i64), ("num_rel", c_i64), ("dim", c_i64), ("rel_dim", c_i64),
        ("ent_ld", c_i64), ("rel_ld", c_i64), ("l_norm", ctypes.c_float),
        ("flags", ctypes.c_int32),
    ]


class KgeIndex(ctypes.Structure):
    _fields_ = [("ptr", c_vp), ("itype", ctypes.c_int32), ("start", ctypes.c_int32),
                ("stride", c_i64)]


class KgeNextQueries(ctypes.Structure):
    _fields_ = [("s", KgeIndex), ("p", KgeIndex), ("o", KgeIndex), ("n", c_i64), ("queries", c_vp),
                ("queries_bytes", c_i64)]


class KgeLabelQueries(ctypes.Structure):
    _fields_ = [("a", KgeIndex), ("p", KgeIndex), ("n", c_i64), ("lbl_rowptr", c_vp), ("lbl_col", c_vp), ("lse", c_vp),
                ("g_rows", c_vp), ("g_scalar", ctypes.c_float), ("g_dev", c_vp)]


LOSS_KL, LOSS_BCE = 0, 1


class KgeAdagradSeg(ctypes.Structure):
    _fields_ = [("param", c_vp), ("grad", c_vp), ("state_sum", c_vp), ("bf16_copy", c_vp), ("count", c_i64),
                ("minus_clr", ctypes.c_float), ("weight_decay", ctypes.c_float), ("eps", ctypes.c_float)]


ADAGRAD_MAX_SEGS = 8


class KgePenaltySeg(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("p", ctypes.c_int32), ("weight", ctypes.c_float), ("row_dim", c_i64),
                ("value", c_vp)]


class KgeRankBand(ctypes.Structure):
    _fields_ = [("table_max_norm", c_vp), ("list", c_vp), ("list_bytes", c_i64), ("status", c_vp)]


class KgeEvalFilter(ctypes.Structure):
    _fields_ = [("sp_keys", c_vp), ("sp_num_keys", c_i64), ("sp_starts", c_vp), ("sp_values", c_vp),
                ("po_keys", c_vp), ("po_num_keys", c_i64), ("po_starts", c_vp), ("po_values", c_vp)]


class KgeFilterQuery(ctypes.Structure):
    _fields_ = [("sorted_keys", c_vp), ("num_keys", c_i64), ("starts", c_vp), ("a", KgeIndex), ("b", KgeIndex),
                ("mult", c_i64), ("begin", c_vp), ("end", c_vp)]


# every symbol include/kge_amd.h declares: name -> (restype, argtypes)
_PT = ctypes.POINTER(KgeTables)
PROTOTYPES = {
    "kge_abi_version": (ctypes.c_int, []),
    "kge_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "kge_device_count": (ctypes.c_int, []),
    "kge_score_spo": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp]),
    "kge_score_workspace_bytes": (c_i64, [_PT, c_i64]),
    "kge_score_sp": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, c_i64, KgeIndex, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "kge_score_po": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, c_i64, KgeIndex, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "kge_score_sp_po": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, KgeIndex, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "kge_queries_bytes": (c_i64, [_PT, ctypes.c_int, c_i64]),
    "kge_build_queries": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_i64, c_vp]),
    "kge_score_queries": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, KgeIndex, c_i64, c_vp, c_i64, c_i64,
                                         ctypes.POINTER(KgeNextQueries), c_vp]),
    "kge_build_queries_multi": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, KgeIndex, c_i64, c_i64, c_vp, c_i64,
                                               c_i64, c_vp]),
    "kge_score_queries_multi": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, c_i64, c_i64, KgeIndex, c_i64, c_vp, c_i64,
                                               c_i64, c_i64, ctypes.POINTER(KgeNextQueries), c_i64, c_vp]),
    "kge_score_neg": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, ctypes.c_int, c_vp,
                                     ctypes.c_int32, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "kge_score_emb": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                     c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "kge_embed": (ctypes.c_int, [_PT, KgeIndex, c_i64, c_vp, c_i64, KgeIndex, c_i64, c_vp, c_i64, c_vp]),
    "kge_ns_bce_loss": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i64, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_vp, c_vp,
                                       c_i64, c_vp]),
    "kge_shard_gather": (ctypes.c_int, [_PT, c_i64, ctypes.POINTER(KgeIndex), ctypes.c_int, c_i64, c_vp, c_i64,
                                        KgeIndex, c_vp, c_i64, c_vp]),
    "kge_shard_pick": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_i64, ctypes.c_int,
                                      ctypes.POINTER(KgeIndex), ctypes.c_int, c_i64, c_vp, c_i64, c_vp]),
    "kge_score_emb_sp_po": (ctypes.c_int, [_PT, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                                           c_vp, c_i64, c_vp, c_i64, c_vp]),
    "kge_score_emb_sp_po_blocks": (ctypes.c_int, [_PT, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                                           c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "kge_rank_counts": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp,
                                       ctypes.c_float, ctypes.c_float, c_vp, c_vp, c_vp]),
    "kge_filter_lookup": (ctypes.c_int, [c_vp, c_i64, c_vp, KgeIndex, KgeIndex, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "kge_filter_lookup_multi": (ctypes.c_int, [ctypes.POINTER(KgeFilterQuery), ctypes.c_int, c_i64, c_vp]),
    "kge_rank_counts_multi": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, ctypes.c_int, c_vp, c_vp, c_vp,
                                             c_i64, c_vp, ctypes.c_float, ctypes.c_float, c_vp, c_vp, c_vp]),
    "kge_score_rank_bits_bytes": (c_i64, [c_i64, c_i64, ctypes.c_int]),
    "kge_score_rank_sp_po": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_i64, c_i64, c_vp, c_vp,
                                            ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_float,
                                            ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                            c_vp]),
    "kge_score_rank_sp_po_band": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_i64, c_i64, c_vp, c_vp,
                                                 ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_float,
                                                 ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                                 c_vp, ctypes.POINTER(KgeRankBand)]),
    "kge_table_max_row_norm": (ctypes.c_int, [_PT, c_i64, c_i64, c_vp, c_vp]),
    "kge_rank_band_list_bytes": (c_i64, [c_i64]),
    "kge_score_rank_emb_sp_po": (ctypes.c_int, [_PT, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, KgeIndex, KgeIndex, c_i64,
                                                c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, ctypes.c_int, c_vp, c_vp, c_vp,
                                                c_vp, c_vp, c_vp, ctypes.c_float, ctypes.c_float, c_vp, c_vp, c_vp,
                                                c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "kge_eval_batch_scratch_bytes": (c_i64, [_PT, c_i64, ctypes.c_int]),
    "kge_eval_batch": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, ctypes.c_int,
                                      ctypes.POINTER(KgeEvalFilter), ctypes.c_float, ctypes.c_float, ctypes.c_int, c_vp,
                                      c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "kge_eval_batch_band": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, ctypes.c_int,
                                           ctypes.POINTER(KgeEvalFilter), ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                           c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp,
                                           ctypes.POINTER(KgeRankBand)]),
    "kge_rank_hist": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_i64, ctypes.c_int, c_vp, c_i64, c_i64, c_vp,
                                     c_vp]),
    "kge_score_bwd_workspace_bytes": (c_i64, [_PT, c_i64, c_i64]),
    "kge_kl_weighted_fwd": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_vp, c_i64, c_vp]),
    "kge_kl_weighted_bwd": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_ce_emb_fwd": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, c_vp, c_i64, KgeIndex, c_i64, c_vp, c_vp,
                                      c_vp, c_i64, c_vp]),
    "kge_ce_emb_bwd": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, c_vp, c_i64, KgeIndex, c_i64, c_vp, c_vp,
                                      ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_kl_weighted_emb_fwd": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp,
                                               c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_kl_weighted_emb_bwd": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp,
                                               c_vp, c_vp, c_vp, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_bce_emb_fwd": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64,
                                       ctypes.c_float, c_vp, c_vp, c_i64, c_vp]),
    "kge_bce_emb_bwd": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64,
                                       ctypes.c_float, c_vp, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_adam_step": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_float, ctypes.c_float, c_vp, c_vp]),
    "kge_adagrad_step_rows": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64,
                                             ctypes.c_float, ctypes.c_float, c_vp, c_i64, c_vp]),
    "kge_score_pairs_bwd": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, c_i64, KgeIndex,
                                           c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp,
                                           c_i64, c_vp]),
    "kge_ce_workspace_bytes": (c_i64, [_PT, c_i64]),
    "kge_ce_fwd": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, c_vp,
                                  c_i64, c_vp]),
    "kge_ce_bwd": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp,
                                  ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_ce_sp_po_workspace_bytes": (c_i64, [_PT, c_i64]),
    "kge_ce_sp_po_fwd": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_ce_sp_po_bwd": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, ctypes.c_float, c_vp,
                                        c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_ce_sp_po_bwd_accum": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, ctypes.c_float,
                                              c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_ce_sp_po_fwd_sum": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, c_vp,
                                            ctypes.c_float, c_vp, c_vp, c_i64, c_vp]),
    "kge_ce_sp_po_bwd_accum_sum": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, c_vp,
                                                  ctypes.c_float, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_adagrad_step_multi": (ctypes.c_int, [ctypes.POINTER(KgeAdagradSeg), ctypes.c_int, c_vp]),
    "kge_adagrad_step_multi_penalty": (ctypes.c_int, [ctypes.POINTER(KgeAdagradSeg), ctypes.POINTER(KgePenaltySeg),
                                                      ctypes.c_int, c_vp]),
    "kge_multilabel2_workspace_bytes": (c_i64, [_PT, c_i64, c_i64]),
    "kge_multilabel2_bwd_accum": (ctypes.c_int, [_PT, ctypes.c_int, ctypes.c_float, ctypes.POINTER(KgeLabelQueries),
                                                 ctypes.POINTER(KgeLabelQueries), c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_kl_fwd": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                  c_i64, c_vp]),
    "kge_kl_bwd": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, c_vp, c_vp,
                                  ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_bce_fwd": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, ctypes.c_float, c_vp,
                                   c_vp, c_i64, c_vp]),
    "kge_bce_bwd": (ctypes.c_int, [_PT, ctypes.c_int, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, ctypes.c_float, c_vp,
                                   ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kge_adagrad_step": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                        c_vp, c_vp]),
    "kge_score_spo_bwd": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp,
                                         c_vp, c_vp, c_vp, c_vp]),
    "kge_score_spo_bwd_accum": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, c_vp, c_vp, c_vp,
                                               c_i64, c_vp, c_i64, c_vp]),
    "kge_score_neg_bwd_accum": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, ctypes.c_int, c_vp,
                                               ctypes.c_int32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64,
                                               c_vp, c_i64, c_vp, c_i64, c_vp]),
    "kge_neg_order": (ctypes.c_int, [c_vp, ctypes.c_int32, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "kge_score_neg_bwd_accum_sorted": (ctypes.c_int, [_PT, KgeIndex, KgeIndex, KgeIndex, c_i64, ctypes.c_int, c_vp,
                                                      ctypes.c_int32, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64,
                                                      c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "kge_score_emb_bwd": (ctypes.c_int, [_PT, ctypes.c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                         c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp,
                                         c_vp]),
}


EXT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_C.so")
_EXT_SRC = os.path.join(_CSRC, "torch_ext.cpp")


def build_extension(force: bool = False) -> str:
    """kge_amd/_C.so: the PyTorch-ROCm C++ extension over the C ABI (csrc/torch_ext.cpp), built in-tree with g++
    against this interpreter's torch headers and linked to libkge_amd.so (rpath $ORIGIN/csrc) -- host code only, the
    kernels are in the library."""
    import sysconfig
    from torch.utils import cpp_extension as ce
    if not force and os.path.exists(EXT_PATH) and os.path.getmtime(EXT_PATH) >= max(
            os.path.getmtime(_EXT_SRC), os.path.getmtime(os.path.join(_CSRC, "..", "..", "include", "kge_amd.h"))):
        return EXT_PATH
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = [sysconfig.get_paths()["include"]] + list(ce.include_paths())
    for extra in ("/opt/rocm/include",):
        if extra not in inc and os.path.isdir(extra):
            inc.append(extra)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-attributes", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_C", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}"]
    cmd += [f"-I{p}" for p in inc]
    cmd += [_EXT_SRC, "-o", EXT_PATH, f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip",
            "-ltorch_python", f"-L{_CSRC}", "-lkge_amd", "-Wl,-rpath,$ORIGIN/csrc", f"-Wl,-rpath,{tlib}"]
    subprocess.check_call(cmd)
    return EXT_PATH


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (cross-compiles without a GPU) and the torch extension over them."""
    cmd = ["make", "-C", _CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    try:
        build_extension(force)
    except (subprocess.CalledProcessError, OSError, ImportError) as exc:
        # the extension is a second binding of the same library: a box without g++ / the torch headers still has the
        # ctypes binding (engine._ext falls back to it), so the library build does not fail on it -- loudly
        import warnings
        if os.path.exists(EXT_PATH):
            os.remove(EXT_PATH)  # never leave a stale one behind
        warnings.warn(f"kge_amd: kge_amd._C (torch extension) was not built ({type(exc).__name__}: {exc}); "
                      "the ctypes binding of libkge_amd.so is used")
    return LIB_PATH


_ext = None


def ext():
    """The torch extension module kge_amd._C (raises if it has not been built)."""
    global _ext
    if _ext is None:
        lib()  # libkge_amd.so first (the same HIP runtime as torch)
        if not os.path.exists(EXT_PATH):
            raise RuntimeError(f"{EXT_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        import importlib.util
        spec = importlib.util.spec_from_file_location("kge_amd._C", EXT_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.abi_version() != 1:
            raise RuntimeError("kge_amd._C: ABI version mismatch with libkge_amd.so")
        _ext = mod
    return _ext


_lib = None


def lib():
    """Load libkge_amd.so (raises if it has not been built: no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (make -C kge_amd/csrc). kge_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.kge_abi_version() != 1:
            raise RuntimeError("libkge_amd.so ABI version mismatch")
        _lib = L
    return _lib


def check(status: int, what: str):
    if status != KGE_OK:
        msg = lib().kge_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (kge_status {status})")


# ---- measurement switches (include/kge_amd_debug.h: kge_debug_set_switch; kge_amd/csrc/switches.hpp lists them) ----
# The library reads no environment variable for kernel selection: tests and tools/ flip a switch through this call.
def set_switch(name: str, value=None) -> None:
    """value None (or < 0) = unset: the library's own choice."""
    L = lib()
    L.kge_debug_set_switch.restype = ctypes.c_int
    L.kge_debug_set_switch.argtypes = [ctypes.c_char_p, c_i64]
    check(L.kge_debug_set_switch(name.encode(), -1 if value is None else int(value)), f"kge_debug_set_switch({name})")


def get_switch(name: str):
    L = lib()
    L.kge_debug_get_switch.restype = c_i64
    L.kge_debug_get_switch.argtypes = [ctypes.c_char_p]
    v = L.kge_debug_get_switch(name.encode())
    if v == -2:
        raise KeyError(name)
    return None if v < 0 else int(v)


class switches:
    """with _lib.switches(V8=0, CE_V3=1): ...   -- restores the previous values on exit (names without KGE_)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_switch(k) for k in self.kw}
        for k, v in self.kw.items():
            set_switch(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_switch(k, v)
        return False


def apply_switch_spec(spec: str) -> None:
    """'V8=0,CE_V3=1' -> set_switch calls; what tools/ scripts take on their command line (--switches) or, for the shell
    wrappers, from KGE_AMD_TOOL_SWITCHES -- read by the TOOL, never by the library or the package."""
    for item in filter(None, (x.strip() for x in spec.split(","))):
        k, _, v = item.partition("=")
        set_switch(k.strip(), int(v))
