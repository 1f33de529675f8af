// Instantiations of the bf16 MFMA GEMM for the A_MK x B_NK operand layouts (see gemm_bf16_impl.h), including the
// variants whose A and / or B operand already is bf16 in HBM (TecmGemm::io_bf16).
#include "gemm_bf16_impl.h"

int tecm_gemm16_dma_try(const TecmGemm& g, hipStream_t st);     // gemm_bf16_dma.hip

int tecm_gemm16_res_a_mk_nk(const TecmGemm& g, hipStream_t st);  // gemm_bf16_res_mk.hip

int tecm_gemm16_dispatch_mk_nk(const TecmGemm& g, bool win, bool drop, hipStream_t st) {
  const bool a16 = g.io_bf16 & TECM_IO_A_BF16, b16 = g.io_bf16 & TECM_IO_B_BF16;
  if (win && a16 && b16 && !drop && g.a_win.enabled && !g.b_win.enabled) {
    const int served = tecm_gemm16_dma_try(g, st);      // pad-free window views whose taps are whole K-tiles
    if (served != 0) return served;
  }
  if (win && (a16 || b16)) {
    TECM_REQUIRE(a16 && !b16 && !g.b_win.enabled, TECM_E_ARG,
                 "tecm_gemm_bf16: MK x NK with a window serves a bf16 A only (B fp32, no b_win)");
    return tecm_gemm16_res_a_mk_nk(g, st);
  }
  if (a16 && b16) {
    const int served = tecm_gemm16_dma_try(g, st);      // 256 x 256 LDS-DMA kernel when the shape qualifies
    if (served != 0) return served;
  }
  if (a16 && b16) return tecm_gemm16::launch<TECM_A_MK, TECM_B_NK, false, false, 1, 1>(g, st);
  if (a16) return tecm_gemm16::launch<TECM_A_MK, TECM_B_NK, false, false, 1, 0>(g, st);
  if (b16) return tecm_gemm16::launch<TECM_A_MK, TECM_B_NK, false, false, 0, 1>(g, st);
  return tecm_gemm16::dispatch<TECM_A_MK, TECM_B_NK>(g, win, drop, st);
}
