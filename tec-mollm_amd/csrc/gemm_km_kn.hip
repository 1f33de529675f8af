// Instantiations of the fp32 MFMA GEMM for the A_KM x B_KN operand layouts (see gemm_impl.h).
#include "gemm_impl.h"

int tecm_gemm_dispatch_km_kn(const TecmGemm& g, int avec, int bvec, bool win, bool drop, hipStream_t st) {
  if (g.N <= 32) return tecm_gemm::dispatch<TECM_A_KM, TECM_B_KN, 32>(g, avec, bvec, win, drop, st);
  if (g.M <= 64 && avec == 4 && bvec == 4) {
    if (g.N <= 64) return tecm_gemm::dispatch_m64<TECM_A_KM, TECM_B_KN, 64>(g, win, drop, st);
    return tecm_gemm::dispatch_m64<TECM_A_KM, TECM_B_KN, 128>(g, win, drop, st);
  }
  if (g.N <= 64) return tecm_gemm::dispatch<TECM_A_KM, TECM_B_KN, 64>(g, avec, bvec, win, drop, st);
  return tecm_gemm::dispatch<TECM_A_KM, TECM_B_KN, 128>(g, avec, bvec, win, drop, st);
}
