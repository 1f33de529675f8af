// bf16 MFMA GEMM instances of the weight-gradient form (A [k][m], B [k][n]) with ONE operand in HBM as bf16
// (gemm_bf16_impl.h: TStager16): conv dW (A = dy bf16, B = the block input through its window) and the 1x1 conv's dW
// (A = dout, B = gelu(GroupNorm(.)) bf16 through the stride window).
#include "gemm_bf16_impl.h"

int tecm_gemm16_res_a_km_kn(const TecmGemm& g, hipStream_t st) {
  return tecm_gemm16::launch<TECM_A_KM, TECM_B_KN, true, false, 1, 0>(g, st);
}
int tecm_gemm16_res_b_km_kn(const TecmGemm& g, hipStream_t st) {
  return tecm_gemm16::launch<TECM_A_KM, TECM_B_KN, true, false, 0, 1>(g, st);
}
// BOTH operands bf16: conv dW with A = dy and B = the bf16 copy of the block input through its window (round 3)
int tecm_gemm16_res_ab_km_kn(const TecmGemm& g, hipStream_t st) {
  return tecm_gemm16::launch<TECM_A_KM, TECM_B_KN, true, false, 1, 1>(g, st);
}
