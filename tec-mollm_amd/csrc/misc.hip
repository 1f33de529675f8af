// Small kernels around the GEMMs: fused Huber loss + gradient (train.py:372), Conv1d weight
// (Cout,Cin,k) <-> GEMM operand packing, scaled transposes for the K-extended c_attn weight.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void huber_stage1(const float* __restrict__ pred, const float* __restrict__ target,
                                                    float* __restrict__ dpred, float* __restrict__ ws, int64_t n,
                                                    float delta, float gscale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float r = pred[i] - target[i];
    const float a = fabsf(r);
    acc += a <= delta ? 0.5f * r * r : delta * (a - 0.5f * delta);
    if (dpred) dpred[i] = (a <= delta ? r : (r > 0.f ? delta : -delta)) * gscale;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// the same over a logical (B, H, N) index space with separate strides per tensor: the model's output is a permuted view
// (B, L_out, N, 1) of its (B, N, L_out) storage (tec_mollm.py:122-123) and the target arrives in its own layout; the
// gradient is written in the PREDICTION's layout, so the head's backward reads a contiguous tensor
struct HuberStrides { int64_t pb, ph, pn, tb, th, tn; int32_t H, N; };
__global__ __launch_bounds__(256) void huber_stage1_strided(const float* __restrict__ pred, const float* __restrict__ target,
                                                            float* __restrict__ dpred, float* __restrict__ ws, int64_t n,
                                                            float delta, float gscale, const HuberStrides s) {
  __shared__ float red[4];
  float acc = 0.f;
  // storage order of pred: (b, n, h) with h fastest when ph == 1 -- walk the index space so that pred / dpred coalesce
  const int64_t HN = (int64_t)s.H * s.N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HN;
    const int64_t rem = i - b * HN;
    const int32_t nn = (int32_t)(rem / s.H), h = (int32_t)(rem - (int64_t)nn * s.H);
    const int64_t ip = b * s.pb + h * s.ph + nn * s.pn;
    const float r = pred[ip] - target[b * s.tb + h * s.th + nn * s.tn];
    const float a = fabsf(r);
    acc += a <= delta ? 0.5f * r * r : delta * (a - 0.5f * delta);
    if (dpred) dpred[ip] = (a <= delta ? r : (r > 0.f ? delta : -delta)) * gscale;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void huber_stage2(const float* __restrict__ ws, int nb, float* __restrict__ out,
                                                    float inv_n) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += ws[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}

__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ fwd,
                                                        float* __restrict__ bwd, int Cout, int Cin, int k) {
  const int total = Cout * Cin * k;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int tap = i % k;
    const int ci = (i / k) % Cin;
    const int co = i / (k * Cin);
    const float v = w[i];
    if (fwd) fwd[(int64_t)co * (k * Cin) + tap * Cin + ci] = v;
    if (bwd) bwd[((int64_t)(k - 1 - tap) * Cout + co) * Cin + ci] = v;
  }
}
__global__ __launch_bounds__(256) void conv_unpack_kernel(const float* __restrict__ dpack, float* __restrict__ dw,
                                                          int Cout, int Cin, int k) {
  const int total = Cout * Cin * k;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int tap = i % k;
    const int ci = (i / k) % Cin;
    const int co = i / (k * Cin);
    dw[i] = dpack[(int64_t)co * (k * Cin) + tap * Cin + ci];
  }
}

// dst[r][c] = scale * src[c][r] through a 32x33 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void transpose_scale_kernel(const float* __restrict__ src, int64_t lds_,
                                                              float* __restrict__ dst, int64_t ldd, int rows, int cols,
                                                              float scale) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    tile[j][tx] = (c < cols && r < rows) ? src[(int64_t)c * lds_ + r] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) dst[(int64_t)r * ldd + c] = scale * tile[tx][j];
  }
}

// bf16 forms of a trainable fp32 weight W [R][C], one launch: `same` = W rounded ([R][C]), `tr` = W^T rounded ([C][R]).
// A bf16 contraction would round the fp32 weight in its loader to exactly these values; as bf16 tensors they can be the
// [row][k] operand of the LDS-DMA GEMM in the forward (same) and in the input-gradient contraction (tr).
__global__ __launch_bounds__(256) void weight_bf16_kernel(const float* __restrict__ src, int64_t lds_, __bf16* __restrict__ same,
                                                          int64_t ld_same, __bf16* __restrict__ tr, int64_t ld_tr, int R,
                                                          int Cc) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    const float v = (r < R && c < Cc) ? src[(int64_t)r * lds_ + c] : 0.f;
    tile[j][tx] = v;
    if (same && r < R && c < Cc) same[(int64_t)r * ld_same + c] = (__bf16)v;
  }
  __syncthreads();
  if (tr)
    for (int j = ty; j < 32; j += 8) {
      const int c = c0 + j, r = r0 + tx;
      if (r < R && c < Cc) tr[(int64_t)c * ld_tr + r] = (__bf16)tile[tx][j];
    }
}

// dst = dropout(src, drop): element (r, c) keeps iff hash(seed, r*drop.ld + c) passes; float4 streaming copy
// (DUAL: an fp32 result AND its bf16 twin in one pass -- a gradient that column sums read exactly and two bf16
//  contractions read rounded)
template <bool OUT16, bool DUAL = false>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ src, int64_t lds_, void* __restrict__ dst_,
                                                            int64_t ldd, int64_t rows, int32_t cols4, uint64_t seed,
                                                            int64_t dld, uint32_t thresh, float inv,
                                                            const uint64_t* __restrict__ sdev,
                                                            __bf16* __restrict__ dst2 = nullptr, int64_t ldd2 = 0) {
  seed = tecm_seed_now(seed, sdev);
  const int64_t total = rows * cols4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols4;
    const int32_t c = (int32_t)(i - r * cols4) * 4;
    float4 v = *reinterpret_cast<const float4*>(src + r * lds_ + c);
    const uint64_t di = (uint64_t)(r * dld + c);
    v.x *= tecm_drop_mult(seed, di, thresh, inv);
    v.y *= tecm_drop_mult(seed, di + 1, thresh, inv);
    v.z *= tecm_drop_mult(seed, di + 2, thresh, inv);
    v.w *= tecm_drop_mult(seed, di + 3, thresh, inv);
    if constexpr (OUT16) tecm_store_bf16x4(static_cast<__bf16*>(dst_) + r * ldd + c, v.x, v.y, v.z, v.w);
    else *reinterpret_cast<float4*>(static_cast<float*>(dst_) + r * ldd + c) = v;
    if constexpr (DUAL) tecm_store_bf16x4(dst2 + r * ldd2 + c, v.x, v.y, v.z, v.w);
  }
}

}  // namespace

// dst (bf16) [r][c] = round(src [r][c]): a strided block of an fp32 matrix into its bf16 twin (the 32 LoRA columns of
// the K-extended c_attn operand, written by an fp32 GEMM into a matrix whose other columns the LayerNorm already
// emitted as bf16)
namespace {
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, int64_t lds,
                                                        __bf16* __restrict__ dst, int64_t ldd, int64_t rows, int cols4) {
  const int64_t total = rows * cols4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols4;
    const int c = 4 * (int)(i - r * cols4);
    const float4 v = *reinterpret_cast<const float4*>(src + r * lds + c);
    tecm_store_bf16x4(dst + r * ldd + c, v.x, v.y, v.z, v.w);
  }
}
}  // namespace

// ---- LoRA fold: the 32 trainable rows / columns of the K-extended c_attn operands [ W ; s B^T ] ([K+r][n]) and
// [ W^T | s B ] ([n][K+r]) from lora_B (n, r), in fp32 or bf16 -- the frozen part of both is filled once per parameter
// version on the host side (tecmollm/functions.py: _kext), so one tiny launch per layer and step replaces the
// copy / scale / transpose / cast sequence that rebuilt the whole 7 MB operand
namespace {
template <typename TK, typename TN>
__global__ __launch_bounds__(256) void lora_fold_kernel(const float* __restrict__ lB, int n_out, int r, float scale,
                                                        TK* __restrict__ w_kn, int64_t ld_kn, TN* __restrict__ w_nk,
                                                        int64_t ld_nk, int k_off) {
  const int total = n_out * r;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int n = i / r, j = i - n * r;
    const float v = scale * lB[i];
    if (w_kn) w_kn[(int64_t)(k_off + j) * ld_kn + n] = (TK)v;
    if (w_nk) w_nk[(int64_t)n * ld_nk + k_off + j] = (TN)v;
  }
}

// up to 12 small fp32 vectors laid end to end in one buffer (the per-branch bias / gamma / beta of a
// Multi_Scale_Conv_Block, modules.py:27-29, which the conv and norm kernels read as one 3*Cout vector each)
struct PackVecArgs { const float* src[12]; int32_t len[12]; int32_t count; };
__global__ __launch_bounds__(256) void pack_vectors_kernel(const PackVecArgs a, float* __restrict__ dst) {
  int off = 0;
  for (int v = 0; v < a.count; ++v) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.len[v]; i += gridDim.x * 256) dst[off + i] = a.src[v][i];
    off += a.len[v];
  }
}
}  // namespace

extern "C" int tecm_lora_fold(const float* lora_B, int32_t n_out, int32_t r, float scale, void* w_kn, int64_t ld_kn,
                              int32_t kn_bf16, void* w_nk, int64_t ld_nk, int32_t nk_bf16, int32_t k_off, void* stream) {
  TECM_REQUIRE(lora_B && (w_kn || w_nk), TECM_E_ARG, "tecm_lora_fold: null pointer");
  TECM_REQUIRE(n_out > 0 && r > 0 && k_off >= 0 && (!w_kn || ld_kn >= n_out) && (!w_nk || ld_nk >= k_off + r), TECM_E_ARG,
               "tecm_lora_fold: bad shape");
  const int total = n_out * r;
  const dim3 grid((total + 255) / 256), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (kn_bf16 && nk_bf16)
    hipLaunchKernelGGL((lora_fold_kernel<__bf16, __bf16>), grid, blk, 0, st, lora_B, n_out, r, scale, static_cast<__bf16*>(w_kn),
                       ld_kn, static_cast<__bf16*>(w_nk), ld_nk, k_off);
  else if (kn_bf16)
    hipLaunchKernelGGL((lora_fold_kernel<__bf16, float>), grid, blk, 0, st, lora_B, n_out, r, scale, static_cast<__bf16*>(w_kn),
                       ld_kn, static_cast<float*>(w_nk), ld_nk, k_off);
  else if (nk_bf16)
    hipLaunchKernelGGL((lora_fold_kernel<float, __bf16>), grid, blk, 0, st, lora_B, n_out, r, scale, static_cast<float*>(w_kn),
                       ld_kn, static_cast<__bf16*>(w_nk), ld_nk, k_off);
  else
    hipLaunchKernelGGL((lora_fold_kernel<float, float>), grid, blk, 0, st, lora_B, n_out, r, scale, static_cast<float*>(w_kn),
                       ld_kn, static_cast<float*>(w_nk), ld_nk, k_off);
  TECM_CHECK_LAUNCH("tecm_lora_fold");
  return TECM_OK;
}

extern "C" int tecm_pack_vectors(const float* const* srcs, const int32_t* lens, int32_t count, float* dst, void* stream) {
  TECM_REQUIRE(srcs && lens && dst && count > 0 && count <= 12, TECM_E_ARG, "tecm_pack_vectors: 1..12 vectors");
  PackVecArgs a;
  int longest = 0;
  for (int v = 0; v < 12; ++v) {
    a.src[v] = v < count ? srcs[v] : nullptr;
    a.len[v] = v < count ? lens[v] : 0;
    TECM_REQUIRE(v >= count || (srcs[v] && lens[v] > 0), TECM_E_ARG, "tecm_pack_vectors: null / empty vector %d", v);
    if (a.len[v] > longest) longest = a.len[v];
  }
  a.count = count;
  hipLaunchKernelGGL(pack_vectors_kernel, dim3((longest + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, dst);
  TECM_CHECK_LAUNCH("tecm_pack_vectors");
  return TECM_OK;
}

extern "C" int tecm_cast_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int32_t cols,
                              void* stream) {
  TECM_REQUIRE(src && dst && rows > 0 && cols > 0, TECM_E_ARG, "tecm_cast_bf16: bad arguments");
  TECM_REQUIRE(cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && tecm_aligned(src, 16) && tecm_aligned(dst, 8),
               TECM_E_ALIGN, "tecm_cast_bf16: rows must be 16-byte (fp32) / 8-byte (bf16) friendly");
  const int64_t total = rows * (cols / 4);
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                     static_cast<__bf16*>(dst), ld_dst, rows, cols / 4);
  TECM_CHECK_LAUNCH("tecm_cast_bf16");
  return TECM_OK;
}

extern "C" int tecm_weight_bf16(const float* src, int64_t ld_src, void* same_bf16, int64_t ld_same, void* transposed_bf16,
                                int64_t ld_tr, int32_t rows, int32_t cols, void* stream) {
  TECM_REQUIRE(src && (same_bf16 || transposed_bf16), TECM_E_ARG, "tecm_weight_bf16: null pointer");
  TECM_REQUIRE(rows > 0 && cols > 0 && ld_src >= cols && (!same_bf16 || ld_same >= cols) && (!transposed_bf16 || ld_tr >= rows),
               TECM_E_ARG, "tecm_weight_bf16: bad shape");
  hipLaunchKernelGGL(weight_bf16_kernel, dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, (hipStream_t)stream, src,
                     ld_src, static_cast<__bf16*>(same_bf16), ld_same, static_cast<__bf16*>(transposed_bf16), ld_tr, rows, cols);
  TECM_CHECK_LAUNCH("tecm_weight_bf16");
  return TECM_OK;
}

extern "C" int tecm_dropout_apply(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t dst_bf16, void* dst2_bf16,
                                  int64_t ld_dst2, int64_t rows, int32_t cols, const TecmDrop* drop, void* stream) {
  TECM_REQUIRE(src && dst && drop, TECM_E_ARG, "tecm_dropout_apply: null pointer");
  TECM_REQUIRE(rows > 0 && cols > 0 && drop->p > 0.f && drop->p < 1.f, TECM_E_ARG, "tecm_dropout_apply: bad shape / p");
  TECM_REQUIRE(cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && tecm_aligned(src, 16) &&
                   tecm_aligned(dst, dst_bf16 ? 8 : 16),
               TECM_E_ALIGN, "tecm_dropout_apply: rows must be 16-byte (fp32) / 8-byte (bf16) friendly");
  TECM_REQUIRE(!dst2_bf16 || (!dst_bf16 && ld_dst2 % 4 == 0 && tecm_aligned(dst2_bf16, 8)), TECM_E_ARG,
               "tecm_dropout_apply: the second (bf16) output goes with an fp32 first output, rows 8-byte friendly");
  const int64_t total = rows * (cols / 4);
  const int64_t want = (total + 255) / 256;
  const dim3 grid((unsigned)(want < 8192 ? want : 8192));
  const uint32_t th = tecm_drop_thresh(drop->p);
  const float inv = 1.0f / (1.0f - drop->p);
  hipStream_t st = (hipStream_t)stream;
  if (dst2_bf16)
    hipLaunchKernelGGL((dropout_apply_kernel<false, true>), grid, dim3(256), 0, st, src, ld_src, dst, ld_dst, rows, cols / 4,
                       drop->seed, drop->ld, th, inv, drop->seed_dev, static_cast<__bf16*>(dst2_bf16), ld_dst2);
  else if (dst_bf16)
    hipLaunchKernelGGL((dropout_apply_kernel<true, false>), grid, dim3(256), 0, st, src, ld_src, dst, ld_dst, rows, cols / 4,
                       drop->seed, drop->ld, th, inv, drop->seed_dev, (__bf16*)nullptr, (int64_t)0);
  else
    hipLaunchKernelGGL((dropout_apply_kernel<false, false>), grid, dim3(256), 0, st, src, ld_src, dst, ld_dst, rows, cols / 4,
                       drop->seed, drop->ld, th, inv, drop->seed_dev, (__bf16*)nullptr, (int64_t)0);
  TECM_CHECK_LAUNCH("tecm_dropout_apply");
  return TECM_OK;
}

extern "C" int tecm_huber_fwd_bwd(const float* pred, const float* target, float* dpred, float* loss_out, int64_t n,
                                  float delta, float grad_scale, float* workspace, void* stream) {
  TECM_REQUIRE(pred && target && loss_out && workspace, TECM_E_ARG, "tecm_huber_fwd_bwd: null pointer");
  TECM_REQUIRE(n > 0 && delta > 0.f, TECM_E_ARG, "tecm_huber_fwd_bwd: bad n/delta");
  const int64_t want = (n + 255) / 256;
  const int nb = (int)(want < 1024 ? want : 1024);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(huber_stage1, dim3(nb), dim3(256), 0, st, pred, target, dpred, workspace, n, delta,
                     grad_scale / (float)n);
  TECM_CHECK_LAUNCH("tecm_huber_fwd_bwd/stage1");
  hipLaunchKernelGGL(huber_stage2, dim3(1), dim3(256), 0, st, workspace, nb, loss_out, 1.0f / (float)n);
  TECM_CHECK_LAUNCH("tecm_huber_fwd_bwd/stage2");
  return TECM_OK;
}

extern "C" int tecm_huber_fwd_bwd_strided(const float* pred, const int64_t* pred_strides, const float* target,
                                          const int64_t* target_strides, float* dpred, float* loss_out, int32_t B, int32_t H,
                                          int32_t N, float delta, float grad_scale, float* workspace, void* stream) {
  TECM_REQUIRE(pred && target && loss_out && workspace && pred_strides && target_strides, TECM_E_ARG,
               "tecm_huber_fwd_bwd_strided: null pointer");
  TECM_REQUIRE(B > 0 && H > 0 && N > 0 && delta > 0.f, TECM_E_ARG, "tecm_huber_fwd_bwd_strided: bad shape / delta");
  const int64_t n = (int64_t)B * H * N;
  HuberStrides s{pred_strides[0], pred_strides[1], pred_strides[2], target_strides[0], target_strides[1], target_strides[2], H, N};
  const int64_t want = (n + 255) / 256;
  const int nb = (int)(want < 1024 ? want : 1024);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(huber_stage1_strided, dim3(nb), dim3(256), 0, st, pred, target, dpred, workspace, n, delta,
                     grad_scale / (float)n, s);
  TECM_CHECK_LAUNCH("tecm_huber_fwd_bwd_strided/stage1");
  hipLaunchKernelGGL(huber_stage2, dim3(1), dim3(256), 0, st, workspace, nb, loss_out, 1.0f / (float)n);
  TECM_CHECK_LAUNCH("tecm_huber_fwd_bwd_strided/stage2");
  return TECM_OK;
}

extern "C" int tecm_conv_weight_pack(const float* w, float* fwd_pack, float* bwd_pack, int32_t Cout, int32_t Cin,
                                     int32_t k, void* stream) {
  TECM_REQUIRE(w && (fwd_pack || bwd_pack), TECM_E_ARG, "tecm_conv_weight_pack: null pointer");
  TECM_REQUIRE(Cout > 0 && Cin > 0 && k > 0, TECM_E_ARG, "tecm_conv_weight_pack: bad shape");
  const int total = Cout * Cin * k;
  hipLaunchKernelGGL(conv_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, fwd_pack,
                     bwd_pack, Cout, Cin, k);
  TECM_CHECK_LAUNCH("tecm_conv_weight_pack");
  return TECM_OK;
}

extern "C" int tecm_conv_weight_unpack(const float* dpack, float* dw, int32_t Cout, int32_t Cin, int32_t k,
                                       void* stream) {
  TECM_REQUIRE(dpack && dw, TECM_E_ARG, "tecm_conv_weight_unpack: null pointer");
  TECM_REQUIRE(Cout > 0 && Cin > 0 && k > 0, TECM_E_ARG, "tecm_conv_weight_unpack: bad shape");
  const int total = Cout * Cin * k;
  hipLaunchKernelGGL(conv_unpack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, dpack, dw, Cout,
                     Cin, k);
  TECM_CHECK_LAUNCH("tecm_conv_weight_unpack");
  return TECM_OK;
}

extern "C" int tecm_transpose_scale(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t rows, int32_t cols,
                                    float scale, void* stream) {
  TECM_REQUIRE(src && dst, TECM_E_ARG, "tecm_transpose_scale: null pointer");
  TECM_REQUIRE(rows > 0 && cols > 0, TECM_E_ARG, "tecm_transpose_scale: bad shape");
  hipLaunchKernelGGL(transpose_scale_kernel, dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0,
                     (hipStream_t)stream, src, lds, dst, ldd, rows, cols, scale);
  TECM_CHECK_LAUNCH("tecm_transpose_scale");
  return TECM_OK;
}
