// Instantiations of the direct-to-LDS fp32 MFMA GEMM (see gemm_glds_impl.h) for A_MK x {B_NK, B_KN}.
#include "gemm_glds_impl.h"
#include <stdlib.h>

int tecm_gemm_glds_try(const TecmGemm& g, int avec, int bvec, bool win, bool drop, hipStream_t st) {
  static const int mode = [] { const char* e = getenv("TECM_GLDS"); return e ? atoi(e) : 3; }();   // bit0: KN, bit1: NK
  if (g.a_layout != TECM_A_MK || win || drop || avec != 4 || bvec != 4) return 0;
  if (!(mode & (g.b_layout == TECM_B_KN ? 1 : 2))) return 0;
  if (g.K % tecm_gemm::BK != 0 || g.N <= 64 || g._p0 == 0) return 0;
  if (g.b_layout == TECM_B_KN && g.N % 4 != 0) return 0;
  if (g.b_layout == TECM_B_NK) return tecm_gemm::launch_glds<TECM_B_NK>(g, st);
  return tecm_gemm::launch_glds<TECM_B_KN>(g, st);
}
