// Instantiations of the bf16 MFMA GEMM for the A_MK x B_KN operand layouts (see gemm_bf16_impl.h).
#include "gemm_bf16_impl.h"

int tecm_gemm16_dispatch_mk_kn(const TecmGemm& g, bool win, bool drop, hipStream_t st) {
  return tecm_gemm16::dispatch<TECM_A_MK, TECM_B_KN>(g, win, drop, st);
}
