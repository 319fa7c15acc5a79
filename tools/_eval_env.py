"""tools/ only: the KGE_EVAL_* environment variables the probes and profiling scripts were written with, mapped onto
kge_amd.eval.EntityRankingEvaluator.OPTIONS (the package itself reads no such variable since round 6).
    import _eval_env; _eval_env.apply()"""
import os


def apply():
    from kge_amd.eval import EntityRankingEvaluator as Ev
    env = os.environ
    if "KGE_EVAL_TWO_STEP" in env:
        Ev.OPTIONS["two_step"] = env["KGE_EVAL_TWO_STEP"] == "1"
    if "KGE_EVAL_LAUNCH_BY_LAUNCH" in env:
        Ev.OPTIONS["launch_by_launch"] = env["KGE_EVAL_LAUNCH_BY_LAUNCH"] == "1"
    if "KGE_EVAL_RESERVE_CUS" in env:
        Ev.OPTIONS["reserve_cus"] = int(env["KGE_EVAL_RESERVE_CUS"])
    if "KGE_EVAL_GRAPH" in env:
        Ev.OPTIONS["hip_graph"] = env["KGE_EVAL_GRAPH"] != "0"
    if "KGE_EVAL_LANES" in env:
        Ev.OPTIONS["lanes"] = int(env["KGE_EVAL_LANES"])
    if "KGE_EVAL_FUSED_EXACT" in env:
        Ev.OPTIONS["fused_exact"] = env["KGE_EVAL_FUSED_EXACT"] == "1"
