// Instantiations of the split-bf16 (fp32-emulating) GEMM, see gemm_x3_impl.h: x3 = two-way split, three products,
// 256 x 128 x 32 tiles; x6 = three-way split, six products, 256 x 128 x 16 tiles.
#include "gemm_x3_impl.h"

int tecm_gemm_x3_dispatch(const TecmGemm& g, int products, hipStream_t st) {
  if (products == 6) return tecm_gemm3::launch_split<3, 16>(g, st);
  return tecm_gemm3::launch_split<2, 32>(g, st);
}
