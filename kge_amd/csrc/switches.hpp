// switches.hpp -- the library's measurement switches: which kernel generation a call takes, cache policies of the
// score stores, probe variants.  Until round 6 these were ~25 getenv() calls spread over the launchers, read on every
// call: a stray environment variable changed which kernel a production call ran.  Now they are process-local integers
// that ONLY include/kge_amd_debug.h's kge_debug_set_switch() changes (tests and tools/ flip them for A/B runs and
// cross-checks); the library itself reads no environment variable but KGE_ROCTX (roctx ranges, api.hip).  Every
// switch is "unset" (-1) by default = the library's own choice.
#pragma once

namespace kge {

enum Switch : int {
  SW_ONE_CALL_PREPARED = 0, // 0 / 1: query-build launch + prepared kernel for the one-call entries
  SW_ONE_CALL_V8,          // 0: one-call entries with many rows do not take the persistent kernel
  SW_ONE_CALL_V8_MIN_ROWS, // threshold of that route (>= 1024)
  SW_V8_RANK,              // 0: the counting kernel declines everything (pairs_bf16_v4_kernel<V3_RANK>)
  SW_RANK_FUSED_FRONT,     // 0: filter bits and query fragments by separate launches
  SW_BWD_GEMM_LIB,         // 1: the gradient products on gemm32_kernel (cross-check)
  SW_CE_V3,                // 1: fused losses on the single-role kernel
  SW_CE_V8,                // 0: fused losses never on the persistent kernel; 1: whenever its geometry allows
  SW_V4_OWN_BUILD,         // 1: no cooperative query build (every wave builds its own fragments)
  SW_V4_INTERLEAVE,        // 0 / 1: interleaved tiles
  SW_V4_STORE_SC1,         // cache policy of the score stores: 0 plain, 1 sc1, 2 nt, 3 sc1 nt
  SW_V6,                   // 0: the prepared single-batch kernels decline everything
  SW_V7,                   // 0 / 1: the direct-store kernel never / also for unaligned one-sided outputs
  SW_V7_NOSTORE,           // probe
  SW_V7_PROBE,             // probe builds (-DKGE_V7_PROBES)
  SW_V8,                   // 0: the persistent store kernel declines everything; 1: takes single batches too
  SW_V8_VAR,               // probe builds (-DKGE_V8_PROBES)
  SW_V8R_PROBE,            // probe builds (-DKGE_V8_PROBES)
  SW_TRANSE_GENERIC,       // 1: TransE score_sp / score_po on the generic 4 x 4 kernel (the cross-check of pairs_transe_kernel)
  SW_COUNT
};

// value of a switch: -1 = unset
long long sw(Switch s);

}  // namespace kge
