// Instantiations of the bf16 MFMA GEMM for the A_KM x B_KN operand layouts (see gemm_bf16_impl.h).
#include "gemm_bf16_impl.h"

int tecm_gemm16_res_a_km_kn(const TecmGemm& g, hipStream_t st);  // gemm_bf16_res_km.hip
int tecm_gemm16_res_b_km_kn(const TecmGemm& g, hipStream_t st);
int tecm_gemm16_res_ab_km_kn(const TecmGemm& g, hipStream_t st);
int tecm_gemm16_tn_try(const TecmGemm& g, hipStream_t st);          // gemm_bf16_tn.hip

int tecm_gemm16_dispatch_km_kn(const TecmGemm& g, bool win, bool drop, hipStream_t st) {
  const bool a16 = g.io_bf16 & TECM_IO_A_BF16, b16 = g.io_bf16 & TECM_IO_B_BF16;
  if (a16 || b16) {
    if (a16 && b16) {
      const int served = tecm_gemm16_tn_try(g, st);                  // natural-orientation LDS-DMA kernel (weight gradients)
      return served != 0 ? served : tecm_gemm16_res_ab_km_kn(g, st);
    }
    return a16 ? tecm_gemm16_res_a_km_kn(g, st) : tecm_gemm16_res_b_km_kn(g, st);
  }
  return tecm_gemm16::dispatch<TECM_A_KM, TECM_B_KN>(g, win, drop, st);
}
