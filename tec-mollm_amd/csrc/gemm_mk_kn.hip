// Instantiations of the fp32 MFMA GEMM for the A_MK x B_KN operand layouts (see gemm_impl.h).
#include "gemm_impl.h"

int tecm_gemm_dispatch_mk_kn(const TecmGemm& g, int avec, int bvec, bool win, bool drop, hipStream_t st) {
  if (g.N <= 32) return tecm_gemm::dispatch<TECM_A_MK, TECM_B_KN, 32>(g, avec, bvec, win, drop, st);
  if (g.N <= 64) return tecm_gemm::dispatch<TECM_A_MK, TECM_B_KN, 64>(g, avec, bvec, win, drop, st);
  return tecm_gemm::dispatch<TECM_A_MK, TECM_B_KN, 128>(g, avec, bvec, win, drop, st);
}
