// bwd.hip -- autograd twins of the scoring kernels (reference: implicit torch autograd
// through complex.py/distmult.py/transe.py/rotate.py, triggered at train_1vsAll.py:70,81).
#include "common.hpp"

namespace kge {

int run_pairs_bwd(const kge_tables* t, int dir, const Operand& A, const Operand& R,
                  const Operand& TG, long long n, long long m, const float* gout, long long ldg,
                  float* g_a, float* g_p, float* g_tgt, hipStream_t st) {
  (void)t; (void)dir; (void)A; (void)R; (void)TG; (void)n; (void)m; (void)gout; (void)ldg;
  (void)g_a; (void)g_p; (void)g_tgt; (void)st;
  return KGE_ERR_UNSUPPORTED;
}

int run_spo_bwd(const kge_tables* t, const Operand& S, const Operand& R, const Operand& O,
                long long n, const float* gout, float* g_s, float* g_p, float* g_o,
                hipStream_t st) {
  (void)t; (void)S; (void)R; (void)O; (void)n; (void)gout; (void)g_s; (void)g_p; (void)g_o;
  (void)st;
  return KGE_ERR_UNSUPPORTED;
}

}  // namespace kge
