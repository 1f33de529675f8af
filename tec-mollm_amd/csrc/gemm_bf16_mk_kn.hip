// Instantiations of the bf16 MFMA GEMM for the A_MK x B_KN operand layouts (see gemm_bf16_impl.h).
#include "gemm_bf16_impl.h"

int tecm_gemm16_res_a_mk_kn(const TecmGemm& g, hipStream_t st);  // gemm_bf16_res_mk.hip

int tecm_gemm16_dispatch_mk_kn(const TecmGemm& g, bool win, bool drop, hipStream_t st) {
  const bool a16 = g.io_bf16 & TECM_IO_A_BF16, b16 = g.io_bf16 & TECM_IO_B_BF16;
  if (a16 || b16) {
    TECM_REQUIRE(a16 && !b16 && !g.b_win.enabled, TECM_E_ARG, "tecm_gemm_bf16: MK x KN serves a bf16 A only (B fp32, no b_win)");
    return tecm_gemm16_res_a_mk_kn(g, st);
  }
  return tecm_gemm16::dispatch<TECM_A_MK, TECM_B_KN>(g, win, drop, st);
}
