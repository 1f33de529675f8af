// bf16 MFMA GEMM instances whose A operand lives in HBM as bf16 behind a temporal-window view (gemm_bf16_impl.h:
// HStagerW): the strided 1x1 conv of a Multi_Scale_Conv_Block (A = gelu(GroupNorm(.)), modules.py:36) and the conv dX
// GEMMs (A = dy).  One instance per B layout serves windowed and plain A alike.
#include "gemm_bf16_impl.h"

int tecm_gemm16_res_a_mk_nk(const TecmGemm& g, hipStream_t st) {
  return tecm_gemm16::launch<TECM_A_MK, TECM_B_NK, true, false, 1, 0>(g, st);
}
int tecm_gemm16_res_a_mk_kn(const TecmGemm& g, hipStream_t st) {
  return tecm_gemm16::launch<TECM_A_MK, TECM_B_KN, true, false, 1, 0>(g, st);
}
