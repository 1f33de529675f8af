// Instantiation of the bf16x3 (split-bf16, fp32-emulating) GEMM, see gemm_x3_impl.h.
#include "gemm_x3_impl.h"

int tecm_gemm_x3_dispatch(const TecmGemm& g, hipStream_t st) { return tecm_gemm3::launch_x3(g, st); }
