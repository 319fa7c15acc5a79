/*
 * kge_amd_debug.h -- measurement hooks of libkge_amd.so.  NOT part of the drop-in boundary (include/kge_amd.h): nothing
 * in kge_amd/ calls them; tools/ (cycle-stamp probes) and two tests do.  They exist in every build of the library so
 * that a profile in profiles/ can be reproduced on the shipped binary.
 *
 * A "stamp buffer" is 64 x uint64 per workgroup of the next launch(es), written with s_memtime by one lane of the
 * kernel at the points its source marks with stamp(); NULL switches stamping off again.  Stamps are stores: with
 * stamping on, a kernel's counted vector-memory waits cover more operations and it runs slower than in production.
 */
#ifndef KGE_AMD_DEBUG_H
#define KGE_AMD_DEBUG_H

#include "kge_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* stamp buffer for the scoring launches of kge_ce_fwd / kge_ce_bwd (tools/ce_phases.py) */
void kge_debug_ce_stamps(unsigned long long* stamps);
/* stamp buffer for the prepared-query scoring / counting launches (pairs_bf16_v6 / v7 / v8 / v8_rank kernels) that are
 * given none of their own (tools/v6_probe.py, tools/v8_probe.py, tools/r4_diag.py, tools/rank8_stamps.py) */
void kge_debug_v6_stamps(unsigned long long* stamps);
/* stamp buffer for gemm16_kernel (tools/gemm16_phases.py) */
void kge_debug_gemm16_stamps(unsigned long long* stamps);
/* one gradient contraction of the mixed-precision backward on its own: which 0 = dQ = G T, 1 = dT = G^T Q; lib 0 =
 * gemm16_kernel, 1 = gemm32_kernel on the widened operands (tests/test_gpu_bwd_gemm16.py, tools/gemm16_probe.py) */
int kge_debug_gemm16(int which, int lib, int d, int64_t rows, int64_t m, const void* x, int64_t ldx, const void* g16,
                     int64_t mp, float* out, float* scratch, int64_t scratch_bytes, void* stream);
/* kge_score_sp of bf16 tables with a stamp buffer of its own (tools/v2_phases.py, tools/prep_probe.py).  ablate 0: the
 * one-call path; 100 / 101: builder launch + scoring launch on prepared / prepared split queries */
int kge_debug_score_sp_bf16_v2(const kge_tables* t, kge_index s, kge_index p, int64_t n, int64_t m, float* out,
                               int64_t ldo, unsigned long long* stamps, int ablate, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* Launches of the persistent kernels issued by this process so far: which 0 = pairs_bf16_v8_kernel (the store kernel:
 * kge_score_queries_multi, and kge_score_sp / _po / _sp_po with >= 1024 rows at d = 512), 1 = pairs_bf16_v8_rank_kernel
 * (the counting kernel); -1 for any other `which`.  Tests use it to prove which kernel a product call reached. */
int kge_debug_launch_count(int which);

/* The matrix pipe alone (bench.py's `matrix_pipe_probe`): one workgroup of eight waves per compute unit, every wave
 * `iters` x 16 v_mfma_f32_32x32x16_bf16 on two independent accumulators from `operands` (8 waves x 4 x 1 KiB of bf16
 * values, 16-byte aligned, loaded once), nothing else.  Returns the launch's flops (time it with events on `stream`), or a
 * negative kge_status.  `sink`: one float nobody writes. */
double kge_debug_mfma_rate(const void* operands, int iters, float* sink, void* stream);

/* The short form of the correctly rounded float square root the RotatE kernels use (common.hpp: sqrt_rn_fast) against
 * the compiler's IEEE sequence, bit for bit, over the `count` bit patterns from `first_bits` on (all 2^32 in two calls):
 * *mismatches (device, zeroed by the caller) += the number that differ, first16 (device) their first 16 patterns. */
int kge_debug_sqrt_check(uint32_t first_bits, uint64_t count, uint64_t* mismatches, uint32_t* first16, void* stream);

/* The library's measurement switches (kge_amd/csrc/switches.hpp lists them with their meanings): which kernel generation a
 * call takes, the cache policy of score stores, probe variants.  The library reads NO environment variable for them
 * (only KGE_ROCTX, for the roctx ranges): a switch is process-local state that only this call changes -- tests flip
 * "V8", "CE_V8", "V5" ... for cross-checks between kernel generations, tools/ for A/B timings.  `name` without the
 * KGE_ prefix; value < 0 = unset (the library's own choice, the state of every switch at load).  Returns KGE_OK, or
 * KGE_ERR_INVALID_ARG for an unknown name.  Not synchronised with calls in flight on other threads. */
int kge_debug_set_switch(const char* name, int64_t value);
/* the switch's value (-1: unset), -2 for an unknown name */
int64_t kge_debug_get_switch(const char* name);

/* 1: this library was built with -DKGE_STALL_INJECT -- in front of every workgroup barrier and every LDS-DMA piece of
 * the hand-synchronised matrix-core kernels a wave sleeps, with probability 1/4, for a pseudo-random 0 .. ~2,000 cycles
 * (common.hpp: kge_stall).  The race / timing pass of SURVEY.md 5: tools/gpu_stall_inject.sh rebuilds the library that
 * way on the GPU box and runs the bit-equality tests under it (profiles/r6_stall_injection.txt).  0: the product build. */
int kge_debug_stall_build(void);

#ifdef __cplusplus
}
#endif
#endif /* KGE_AMD_DEBUG_H */
