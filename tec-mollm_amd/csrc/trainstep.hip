// The shell around the model step: fused global-norm clip + AdamW on flat
// buffers (train.py:92-109, :358-366), evaluation statistics on device (src/evaluation/metrics.py),
// sliding-window batch assembly (src/data/dataset.py:65-99).  All three are HBM-bound streaming kernels.
#include "common.h"

namespace {

__device__ __forceinline__ double block_sum_f64(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------ clip + AdamW
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ partials) {
  __shared__ double red[4];
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = g4[i];
    acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    acc += (double)(v * v);
  }
  acc = block_sum_f64(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

struct AdamConsts {
  float decay, b1, b2, omb1, omb2, step_size, inv_sqrt_bc2, eps;
};

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float gs, const AdamConsts& c) {
  const float gg = g * gs;
  p *= c.decay;
  m = c.b1 * m + c.omb1 * gg;
  v = c.b2 * v + c.omb2 * gg * gg;
  const float denom = sqrtf(v) * c.inv_sqrt_bc2 + c.eps;
  p -= c.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, const double* __restrict__ partials,
                                                    int n_partials, float* __restrict__ norm_out, float grad_scale,
                                                    float max_norm, AdamConsts c, int zero_grad) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += 256) acc += partials[i];
  acc = block_sum_f64(acc, red);
  const float norm = (float)sqrt(acc) * fabsf(grad_scale);
  float coef = 1.0f;
  if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
  const float gs = grad_scale * coef;

  const int64_t n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, gs, c);
    adam_one(pp.y, gg.y, mm.y, vv.y, gs, c);
    adam_one(pp.z, gg.z, mm.z, vv.z, gs, c);
    adam_one(pp.w, gg.w, mm.w, vv.w, gs, c);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
    if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    adam_one(pp, gg, mm, vv, gs, c);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (zero_grad) g[i] = 0.f;
  }
}

// ------------------------------------------------------------------ evaluation statistics
__device__ __forceinline__ float unscale_f32(float y, double mean, double scale) {
  // sklearn StandardScaler.inverse_transform on a float32 array: X *= scale_ ; X += mean_ (two f32 roundings)
  const float a = (float)((double)y * scale);
  return (float)((double)a + mean);
}
__device__ __forceinline__ float nan_to_num_tec(float v) {
  if (v != v) return 0.f;
  if (isinf(v)) return v > 0.f ? 100.f : 0.f;
  return v;
}

__global__ __launch_bounds__(256) void metrics_kernel(TecmMetrics q, int chunks) {
  __shared__ double red[4];
  const int h = blockIdx.y;
  const int64_t total = q.S * q.I;
  double a[TECM_METRIC_STATS];
#pragma unroll
  for (int k = 0; k < TECM_METRIC_STATS; ++k) a[k] = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)chunks * 256) {
    const int64_t s = e / q.I, i = e - s * q.I;
    float pv = q.pred[s * q.p_stride_s + h * q.p_stride_h + i * q.p_stride_i];
    float tv = q.target[s * q.t_stride_s + h * q.t_stride_h + i * q.t_stride_i];
    if (!isfinite(pv)) pv = 0.f;                                   // metrics.py:139-145
    pv = nan_to_num_tec(unscale_f32(pv, q.mean, q.scale));         // :36-46
    tv = nan_to_num_tec(unscale_f32(tv, q.mean, q.scale));
    if (q.clip) pv = fminf(fmaxf(pv, q.clip_lo), q.clip_hi);       // :50-51
    const double t = tv, p = pv, d = t - p;
    a[0] += 1.0; a[1] += t; a[2] += p; a[3] += t * t; a[4] += p * p; a[5] += t * p; a[6] += fabs(d); a[7] += d * d;
  }
#pragma unroll
  for (int k = 0; k < TECM_METRIC_STATS; ++k) {
    const double v = block_sum_f64(a[k], red);
    if (threadIdx.x == 0) atomicAdd(&q.stats[(int64_t)h * TECM_METRIC_STATS + k], v);
  }
}

// ------------------------------------------------------------------ sliding-window batch assembly
// grid.x covers float4 chunks of one (b, t) row of x; grid.y = B*L_in rows.  Rows are contiguous
// N*C-float runs on both sides => pure streaming copy.
__global__ __launch_bounds__(256) void window_x_kernel(TecmWindowBatch w, int vec) {
  const int bt = blockIdx.y;
  const int b = bt / w.L_in, t = bt - b * w.L_in;
  const int64_t a = w.starts[b];
  const float* src = w.X + (a + t) * w.row;
  float* dst = w.x_out + (int64_t)bt * w.row;
  if (vec) {
    const int64_t n4 = w.row >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) d4[i] = s4[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < w.row; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
  }
  if (blockIdx.x == 0 && w.tf_out && threadIdx.x < w.F_t)
    w.tf_out[(int64_t)bt * w.F_t + threadIdx.x] = w.TF[(a + t) * w.F_t + threadIdx.x];
}
// y_out[b][h][i] = Y[a+L_in-1][i][h] through LDS so that both sides stay coalesced (L_out is small:
// a 64-node slab of Y is 64*L_out contiguous floats).
__global__ __launch_bounds__(256) void window_y_kernel(TecmWindowBatch w) {
  extern __shared__ float slab[];                       // 64 * (L_out + 1)
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * 64;
  const int cnt = min(64, w.N - i0);
  const int64_t a = w.starts[b] + w.L_in - 1;
  const float* src = w.Y + (a * w.N + i0) * w.L_out;
  const int ld = w.L_out + 1;
  for (int e = threadIdx.x; e < cnt * w.L_out; e += 256) {
    const int i = e / w.L_out, h = e - i * w.L_out;
    slab[i * ld + h] = src[e];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * w.L_out; e += 256) {
    const int h = e >> 6, i = e & 63;
    if (i < cnt) w.y_out[((int64_t)b * w.L_out + h) * w.N + i0 + i] = slab[i * ld + h];
  }
}

// ------------------------------------------------------------------ data-parallel parameter checksum (train.py:353-354)
// Every rank writes an f64 checksum of its parameters, split into two floats, into ITS two slots of the tail of the flat
// gradient buffer before the step's one all-reduce; after the SUM every rank holds everybody's and compares.
constexpr int CSUM_BLOCKS = 256;
__global__ __launch_bounds__(256) void checksum_partial_kernel(const float* __restrict__ p, int64_t n, double* __restrict__ ws) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += (double)p[i];
  acc = block_sum_f64(acc, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void checksum_tail_kernel(const double* __restrict__ ws, int nws, float* __restrict__ tail,
                                                            int world, int rank) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nws; i += 256) acc += ws[i];
  acc = block_sum_f64(acc, red);
  const float hi = (float)acc, lo = (float)(acc - (double)hi);
  for (int i = threadIdx.x; i < 2 * world; i += 256) tail[i] = i == 2 * rank ? hi : (i == 2 * rank + 1 ? lo : 0.f);
}
__global__ void checksum_verify_kernel(const float* __restrict__ tail, int world, int32_t* err_word, int32_t bit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < world && (tail[2 * i] != tail[0] || tail[2 * i + 1] != tail[1])) atomicOr(err_word, bit);   // NaN != NaN: flagged
}

}  // namespace

extern "C" int tecm_checksum_tail(const float* param, int64_t n, float* tail, int32_t world, int32_t rank, double* ws,
                                  void* stream) {
  TECM_REQUIRE(param && tail && ws && n > 0 && world >= 1 && rank >= 0 && rank < world, TECM_E_ARG,
               "tecm_checksum_tail: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)((n + 255) / 256 < CSUM_BLOCKS ? (n + 255) / 256 : CSUM_BLOCKS);
  hipLaunchKernelGGL(checksum_partial_kernel, dim3(nb), dim3(256), 0, st, param, n, ws);
  TECM_CHECK_LAUNCH("tecm_checksum_tail/partial");
  hipLaunchKernelGGL(checksum_tail_kernel, dim3(1), dim3(256), 0, st, ws, nb, tail, world, rank);
  TECM_CHECK_LAUNCH("tecm_checksum_tail/tail");
  return TECM_OK;
}

extern "C" int tecm_checksum_verify(const float* tail, int32_t world, int32_t* err_word, int32_t bit, void* stream) {
  TECM_REQUIRE(tail && err_word && world >= 1, TECM_E_ARG, "tecm_checksum_verify: bad arguments");
  hipLaunchKernelGGL(checksum_verify_kernel, dim3((world + 63) / 64), dim3(64), 0, (hipStream_t)stream, tail, world, err_word, bit);
  TECM_CHECK_LAUNCH("tecm_checksum_verify");
  return TECM_OK;
}

extern "C" int tecm_adamw_clip_step(const TecmAdamW* a, void* stream) {
  TECM_REQUIRE(a, TECM_E_ARG, "tecm_adamw_clip_step: null descriptor");
  TECM_REQUIRE(a->param && a->grad && a->exp_avg && a->exp_avg_sq && a->partials, TECM_E_ARG,
               "tecm_adamw_clip_step: null pointer");
  TECM_REQUIRE(a->n > 0 && a->step >= 1, TECM_E_ARG, "tecm_adamw_clip_step: n and step must be >= 1");
  TECM_REQUIRE(tecm_aligned(a->param, 16) && tecm_aligned(a->grad, 16) && tecm_aligned(a->exp_avg, 16) &&
                   tecm_aligned(a->exp_avg_sq, 16) && tecm_aligned(a->partials, 8),
               TECM_E_ALIGN, "tecm_adamw_clip_step: flat buffers must be 16-byte aligned");
  TECM_REQUIRE(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps >= 0.f && a->lr >= 0.f,
               TECM_E_ARG, "tecm_adamw_clip_step: bad hyper-parameters");
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = ((a->n >> 2) + 255) / 256;
  const int nb = (int)(want < 1 ? 1 : (want < TECM_NORM_BLOCKS ? want : TECM_NORM_BLOCKS));
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, a->grad, a->n, a->partials);
  TECM_CHECK_LAUNCH("tecm_adamw_clip_step/sumsq");
  AdamConsts c;
  const double bc1 = 1.0 - pow((double)a->beta1, (double)a->step);
  const double bc2 = 1.0 - pow((double)a->beta2, (double)a->step);
  c.decay = 1.0f - a->lr * a->weight_decay;
  c.b1 = a->beta1; c.b2 = a->beta2; c.omb1 = 1.0f - a->beta1; c.omb2 = 1.0f - a->beta2;
  c.step_size = (float)((double)a->lr / bc1);
  c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  c.eps = a->eps;
  const int64_t blocks = want < 1 ? 1 : (want < 2048 ? want : 2048);
  hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, st, a->param, a->grad, a->exp_avg, a->exp_avg_sq,
                     a->n, a->partials, nb, a->total_norm_out, a->grad_scale, a->max_norm, c, a->zero_grad);
  TECM_CHECK_LAUNCH("tecm_adamw_clip_step/adamw");
  return TECM_OK;
}

extern "C" int tecm_metrics_accumulate(const TecmMetrics* m, void* stream) {
  TECM_REQUIRE(m, TECM_E_ARG, "tecm_metrics_accumulate: null descriptor");
  TECM_REQUIRE(m->pred && m->target && m->stats, TECM_E_ARG, "tecm_metrics_accumulate: null pointer");
  TECM_REQUIRE(m->S > 0 && m->H > 0 && m->H <= 65535 && m->I > 0, TECM_E_ARG, "tecm_metrics_accumulate: bad shape");
  TECM_REQUIRE(m->scale != 0.0, TECM_E_ARG, "tecm_metrics_accumulate: scale must be non-zero");
  TECM_REQUIRE(tecm_aligned(m->stats, 8), TECM_E_ALIGN, "tecm_metrics_accumulate: stats must be 8-byte aligned");
  const int64_t total = m->S * m->I;
  const int64_t want = (total + 256 * 8 - 1) / (256 * 8);
  const int chunks = (int)(want < 1 ? 1 : (want < 256 ? want : 256));
  hipLaunchKernelGGL(metrics_kernel, dim3(chunks, m->H), dim3(256), 0, (hipStream_t)stream, *m, chunks);
  TECM_CHECK_LAUNCH("tecm_metrics_accumulate");
  return TECM_OK;
}

namespace {
__global__ void seed_advance_kernel(uint64_t* word, uint64_t inc) { *word += inc; }
}  // namespace
extern "C" int tecm_seed_advance(uint64_t* word, uint64_t inc, void* stream) {
  TECM_REQUIRE(word && tecm_aligned(word, 8), TECM_E_ARG, "tecm_seed_advance: need an 8-byte aligned device word");
  hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, word, inc);
  TECM_CHECK_LAUNCH("tecm_seed_advance");
  return TECM_OK;
}

extern "C" int tecm_window_batch(const TecmWindowBatch* w, void* stream) {
  TECM_REQUIRE(w, TECM_E_ARG, "tecm_window_batch: null descriptor");
  TECM_REQUIRE(w->X && w->starts && w->x_out, TECM_E_ARG, "tecm_window_batch: null pointer");
  TECM_REQUIRE(w->T > 0 && w->row > 0 && w->N > 0 && w->L_in > 0 && w->B > 0 && w->B * (int64_t)w->L_in <= 65535,
               TECM_E_ARG, "tecm_window_batch: bad shape (B*L_in must be <= 65535)");
  TECM_REQUIRE(!w->tf_out || (w->TF && w->F_t > 0 && w->F_t <= 256), TECM_E_ARG,
               "tecm_window_batch: time features need TF and 0 < F_t <= 256");
  TECM_REQUIRE(!w->y_out || (w->Y && w->L_out > 0 && w->L_out <= 512), TECM_E_ARG,
               "tecm_window_batch: targets need Y and 0 < L_out <= 512");
  if (w->starts_host_check) {
    for (int b = 0; b < w->B; ++b) {
      const int64_t a = w->starts_host_check[b];
      TECM_REQUIRE(a >= 0 && a + w->L_in <= w->T, TECM_E_ARG,
                   "tecm_window_batch: window %d starts at %lld, outside [0, T - L_in]", b, (long long)a);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const int vec = (w->row % 4 == 0) && tecm_aligned(w->X, 16) && tecm_aligned(w->x_out, 16);
  const int64_t per_row = vec ? (w->row >> 2) : w->row;
  int64_t gx = (per_row + 256 * 4 - 1) / (256 * 4);
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  hipLaunchKernelGGL(window_x_kernel, dim3((int)gx, w->B * w->L_in), dim3(256), 0, st, *w, vec);
  TECM_CHECK_LAUNCH("tecm_window_batch/x");
  if (w->y_out) {
    hipLaunchKernelGGL(window_y_kernel, dim3((w->N + 63) / 64, w->B), dim3(256), 64 * (w->L_out + 1) * sizeof(float), st,
                       *w);
    TECM_CHECK_LAUNCH("tecm_window_batch/y");
  }
  return TECM_OK;
}
