// Bandwidth-bound normalisation kernels of the TEC-MoLLM path (gfx950, wave = 64):
//   LayerNorm fwd/bwd     -- GPT2Block.ln_1 / ln_2 / ln_f   (modeling_gpt2.py:262-310, :620)
//   GroupNorm(1)+GELU     -- Multi_Scale_Conv_Block branches (modules.py:28-29), 3 branches at once
//   column sums           -- bias / gamma / beta / wpe gradients
// One wave owns one row (LayerNorm) or one sequence (GroupNorm); statistics are wave reductions,
// per-channel parameter gradients are kept in registers across the rows a wave visits and leave
// the kernel as one partial row per block (summed by tecm_colsum) -- no atomics, deterministic.
#include "common.h"

namespace {

constexpr int LN_MAXCH = 4;   // float4 chunks per lane: D <= 1024

struct DropCtxN {
  uint64_t seed;
  int64_t ld;
  uint32_t thresh;
  float inv;
  const uint64_t* sdev;        // TecmDrop::seed_dev: added to seed when the kernel starts (seed_now below)
};
__device__ __forceinline__ void seed_now(DropCtxN& c) { c.seed = tecm_seed_now(c.seed, c.sdev); }
__host__ __device__ inline DropCtxN make_dropn(const TecmDrop* d) {
  DropCtxN c;
  c.seed = d ? d->seed : 0;
  c.sdev = d ? d->seed_dev : nullptr;
  c.ld = d ? d->ld : 0;
  c.thresh = (d && d->p > 0.f) ? tecm_drop_thresh(d->p) : 0u;
  c.inv = (d && d->p > 0.f) ? 1.0f / (1.0f - d->p) : 1.0f;
  return c;
}

// ------------------------------------------------------------------------------ LayerNorm
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            int64_t ldy, void* __restrict__ y16, int64_t ldy16,
                                                            void* __restrict__ y16d, int64_t ldy16d, DropCtxN dd,
                                                            float* __restrict__ stats, int64_t M, int D, float eps,
                                                            int permT, int permN) {
  seed_now(dd);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= M) return;
  // y16d may be written SEQUENCE-major: time-major row (b, t, n) -> row (b, n, t), i.e. PredictionHead's view(batch, -1)
  // (modules.py:307) of the (B*N, T, D) hidden state as a plain [B*N][T*D] matrix (the mask index stays the time-major one)
  int64_t drow = row;
  if (permT > 0) {
    const int64_t tn = (int64_t)permT * permN;
    const int64_t b = row / tn, rem = row - b * tn;
    const int64_t t = rem / permN, n = rem - t * permN;
    drow = (b * permN + n) * permT + t;
  }
  const float* xr = x + row * ldx;
  float4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = 4 * (lane + 64 * i);
    v[i] = c < D ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = 4 * (lane + 64 * i);
    if (c < D) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = 4 * (lane + 64 * i);
    if (c < D) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (y) *reinterpret_cast<float4*>(y + row * ldy + c) = o;
      // bf16 copy for a bf16 matrix-core GEMM that only ever reads the rounded value (rounded once here)
      if (y16) tecm_store_bf16x4(static_cast<__bf16*>(y16) + row * ldy16 + c, o.x, o.y, o.z, o.w);
      // ... and bf16(dropout(o)): the LoRA branch's input (peft lora_dropout in front of lora_A, modules.py:181), the
      // cast autocast applies to the DROPPED fp32 value; read by the LoRA-A GEMM and by its weight gradient
      if (y16d) {
        const uint64_t di = (uint64_t)(row * dd.ld + c);
        tecm_store_bf16x4(static_cast<__bf16*>(y16d) + drow * ldy16d + c,
                          o.x * tecm_drop_mult(dd.seed, di, dd.thresh, dd.inv),
                          o.y * tecm_drop_mult(dd.seed, di + 1, dd.thresh, dd.inv),
                          o.z * tecm_drop_mult(dd.seed, di + 2, dd.thresh, dd.inv),
                          o.w * tecm_drop_mult(dd.seed, di + 3, dd.thresh, dd.inv));
      }
    }
  }
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

// Optional SECOND gradient stream of the same tensor: dy2 (M, D), fp32 or bf16, times an optional dropout mask -- the gradient
// the LoRA branch of peft's c_attn returns for ITS input (lora_A's d-input GEMM, modules.py:177-186), which reaches the
// LayerNorm output through lora_dropout's backward (modules.py:181):     dy[row][c] += keep(row, c) / (1 - p) * dy2[row][c].
// (Round 4 first folded the rank-32 product itself into this kernel -- lora_A in LDS, one 1024-thread block per CU -- and
// measured 370 us per launch against 148 + 129 for this kernel plus the K = 32 GEMM: the LDS image costs the occupancy a
// bandwidth kernel lives on.  The product stays a GEMM, its read-modify-write of the M x D gradient is what went away.)
struct LnAdd {
  const void* dy2;
  int64_t ld;
  int32_t bf16, _pad;
  DropCtxN drop;
};

// DY16: dy is a bf16 matrix (bf16 mode: the gradient a bf16 Linear hands back for its input, train.py:68)
// DYMAP (with DY16): that matrix is SEQUENCE-major -- row (b, n, t) for the time-major row (b, t, n) this kernel walks -- and
// still in front of a dropout: dy[row][c] = keep(row, c) / (1 - p) * dy16[(b, n, t)][c].  The gradient the head's first
// Linear returns for F.dropout(hidden) (tec_mollm.py:115, modules.py:307), consumed by ln_f's backward without a pass of
// its own for the mask or the layout.
struct LnDyMap {
  int T, N;
  DropCtxN drop;
};
template <int NCH, bool ADD, bool DY16 = false, bool DYMAP = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ dres, float* __restrict__ dx,
                                                            float* __restrict__ dxm, int dxm_bf16, DropCtxN odc,
                                                            float* __restrict__ partials, int64_t M, int D, LnAdd ad,
                                                            LnDyMap dm = LnDyMap{}) {
  seed_now(odc);
  seed_now(ad.drop);
  seed_now(dm.drop);
  __shared__ float red[4][2 * 4 * 64 * NCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 dg[NCH], db[NCH], gm[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = 4 * (lane + 64 * i);
    gm[i] = c < D ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invD = 1.0f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float4 xh[NCH], g[NCH];
    float s1 = 0.f, s2 = 0.f;
    int64_t yrow = row;
    if constexpr (DYMAP) {
      const int64_t tn = (int64_t)dm.T * dm.N;
      const int64_t b = row / tn, rem = row - b * tn;
      const int64_t t = rem / dm.N, n = rem - t * dm.N;
      yrow = (b * dm.N + n) * dm.T + t;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < D) {
        const float4 xv = *reinterpret_cast<const float4*>(x + row * ldx + c);
        float4 d;
        if constexpr (DY16) {
          const tecm_bf16x4 h = *reinterpret_cast<const tecm_bf16x4*>(reinterpret_cast<const __bf16*>(dy) + yrow * lddy + c);
          d = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
          if constexpr (DYMAP) {
            if (dm.drop.thresh) {
              const uint64_t di = (uint64_t)(row * dm.drop.ld + c);
              d.x *= tecm_drop_mult(dm.drop.seed, di, dm.drop.thresh, dm.drop.inv);
              d.y *= tecm_drop_mult(dm.drop.seed, di + 1, dm.drop.thresh, dm.drop.inv);
              d.z *= tecm_drop_mult(dm.drop.seed, di + 2, dm.drop.thresh, dm.drop.inv);
              d.w *= tecm_drop_mult(dm.drop.seed, di + 3, dm.drop.thresh, dm.drop.inv);
            }
          }
        } else {
          d = *reinterpret_cast<const float4*>(dy + row * lddy + c);
        }
        if constexpr (ADD) {
          float4 e;
          if (ad.bf16) {
            const tecm_bf16x4 h = *reinterpret_cast<const tecm_bf16x4*>(reinterpret_cast<const __bf16*>(ad.dy2) + row * ad.ld + c);
            e = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
          } else {
            e = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ad.dy2) + row * ad.ld + c);
          }
          if (ad.drop.thresh) {
            const uint64_t di = (uint64_t)(row * ad.drop.ld + c);
            e.x *= tecm_drop_mult(ad.drop.seed, di, ad.drop.thresh, ad.drop.inv);
            e.y *= tecm_drop_mult(ad.drop.seed, di + 1, ad.drop.thresh, ad.drop.inv);
            e.z *= tecm_drop_mult(ad.drop.seed, di + 2, ad.drop.thresh, ad.drop.inv);
            e.w *= tecm_drop_mult(ad.drop.seed, di + 3, ad.drop.thresh, ad.drop.inv);
          }
          d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
        }
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
        dg[i].x += d.x * xh[i].x; dg[i].y += d.y * xh[i].y; dg[i].z += d.z * xh[i].z; dg[i].w += d.w * xh[i].w;
        db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      } else {
        xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        g[i] = xh[i];
      }
    }
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < D) {
        float o[4] = {rstd * (g[i].x - s1 - xh[i].x * s2), rstd * (g[i].y - s1 - xh[i].y * s2),
                      rstd * (g[i].z - s1 - xh[i].z * s2), rstd * (g[i].w - s1 - xh[i].w * s2)};
        if (dres) {
          const float4 r = *reinterpret_cast<const float4*>(dres + row * (int64_t)D + c);
          o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
        *reinterpret_cast<float4*>(dx + row * (int64_t)D + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (dxm) {          // second output: dropout(dx) for the GEMM that consumes the masked gradient
          if (odc.thresh) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              o[e] *= tecm_drop_mult(odc.seed, (uint64_t)(row * odc.ld + c + e), odc.thresh, odc.inv);
          }
          if (dxm_bf16)
            tecm_store_bf16x4(reinterpret_cast<__bf16*>(dxm) + row * (int64_t)D + c, o[0], o[1], o[2], o[3]);
          else
            *reinterpret_cast<float4*>(dxm + row * (int64_t)D + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
  // block reduction of the per-lane parameter gradients -> one partial row per block
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = 4 * (lane + 64 * i);
    float* r0 = &red[wave][c];
    float* r1 = &red[wave][4 * 64 * NCH + c];
    r0[0] = dg[i].x; r0[1] = dg[i].y; r0[2] = dg[i].z; r0[3] = dg[i].w;
    r1[0] = db[i].x; r1[1] = db[i].y; r1[2] = db[i].z; r1[3] = db[i].w;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    const int src = c < D ? c : (4 * 64 * NCH + (c - D));
    partials[(int64_t)blockIdx.x * 2 * D + c] = (red[0][src] + red[1][src]) + (red[2][src] + red[3][src]);
  }
}

// ------------------------------------------------------------------------------ GroupNorm(1) + GELU
// CPB = Cout / 64 (channel chunks of 64 per branch); lane owns channels lane + 64*i.
template <int CPB>
__global__ __launch_bounds__(256) void gn_gelu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ act,
                                                          float* __restrict__ stats, int B, int L, int N, float eps) {
  constexpr int CT = 3 * CPB * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t sidx = (int64_t)blockIdx.x * 4 + wave;
  if (sidx >= (int64_t)B * N) return;
  const int b = (int)(sidx / N), n = (int)(sidx - (int64_t)b * N);
  const float inv_cnt = 1.0f / (float)(L * CPB * 64);
  const int64_t tstride = (int64_t)N * CT;
  const float* base = y + ((int64_t)b * L * N + n) * CT + lane;
  float mean[3], rstd[3];
  {
    float s[3] = {0.f, 0.f, 0.f};
    for (int t = 0; t < L; ++t) {
      const float* p = base + t * tstride;
#pragma unroll
      for (int br = 0; br < 3; ++br)
#pragma unroll
        for (int k = 0; k < CPB; ++k) s[br] += p[64 * (br * CPB + k)];
    }
#pragma unroll
    for (int br = 0; br < 3; ++br) mean[br] = wave_sum(s[br]) * inv_cnt;
  }
  {
    float s[3] = {0.f, 0.f, 0.f};
    for (int t = 0; t < L; ++t) {
      const float* p = base + t * tstride;
#pragma unroll
      for (int br = 0; br < 3; ++br)
#pragma unroll
        for (int k = 0; k < CPB; ++k) {
          const float d = p[64 * (br * CPB + k)] - mean[br];
          s[br] += d * d;
        }
    }
#pragma unroll
    for (int br = 0; br < 3; ++br) rstd[br] = 1.0f / sqrtf(wave_sum(s[br]) * inv_cnt + eps);
  }
  float gm[3 * CPB], bt[3 * CPB];
#pragma unroll
  for (int i = 0; i < 3 * CPB; ++i) {
    gm[i] = gamma[lane + 64 * i];
    bt[i] = beta[lane + 64 * i];
  }
  float* obase = act + ((int64_t)b * L * N + n) * CT + lane;
  for (int t = 0; t < L; ++t) {
    const float* p = base + t * tstride;
    float* o = obase + t * tstride;
#pragma unroll
    for (int br = 0; br < 3; ++br)
#pragma unroll
      for (int k = 0; k < CPB; ++k) {
        const int i = br * CPB + k;
        o[64 * i] = gelu_erf((p[64 * i] - mean[br]) * rstd[br] * gm[i] + bt[i]);
      }
  }
  if (lane < 3) {
    stats[(sidx * 3 + lane) * 2] = mean[lane];
    stats[(sidx * 3 + lane) * 2 + 1] = rstd[lane];
  }
}

template <int CPB>
__global__ __launch_bounds__(256) void gn_gelu_bwd_kernel(const float* __restrict__ dact, int dstride, int L2,
                                                          const float* __restrict__ y,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ stats, float* __restrict__ dy,
                                                          float* __restrict__ partials, int B, int L, int N) {
  constexpr int NCHK = 3 * CPB;
  constexpr int CT = NCHK * 64;
  __shared__ float red[4][3 * CT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_cnt = 1.0f / (float)(L * CPB * 64);
  const int64_t tstride = (int64_t)N * CT;
  float gm[NCHK], bt[NCHK], dgm[NCHK], dbt[NCHK], dys[NCHK];
#pragma unroll
  for (int i = 0; i < NCHK; ++i) {
    gm[i] = gamma[lane + 64 * i];
    bt[i] = beta[lane + 64 * i];
    dgm[i] = 0.f;
    dbt[i] = 0.f;
    dys[i] = 0.f;
  }
  for (int64_t sidx = (int64_t)blockIdx.x * 4 + wave; sidx < (int64_t)B * N; sidx += (int64_t)gridDim.x * 4) {
    const int b = (int)(sidx / N), n = (int)(sidx - (int64_t)b * N);
    float mean[3], rstd[3];
#pragma unroll
    for (int br = 0; br < 3; ++br) {
      mean[br] = stats[(sidx * 3 + br) * 2];
      rstd[br] = stats[(sidx * 3 + br) * 2 + 1];
    }
    const float* ybase = y + ((int64_t)b * L * N + n) * CT + lane;
    const float* dbase = dact + ((int64_t)b * L2 * N + n) * CT + lane;
    float* obase = dy + ((int64_t)b * L * N + n) * CT + lane;
    float s1[3] = {0.f, 0.f, 0.f}, s2[3] = {0.f, 0.f, 0.f};
    for (int t = 0; t < L; ++t) {
      const bool has = (t % dstride) == 0;
      const float* p = ybase + t * tstride;
      const float* dp = dbase + (t / dstride) * tstride;
#pragma unroll
      for (int br = 0; br < 3; ++br)
#pragma unroll
        for (int k = 0; k < CPB; ++k) {
          const int i = br * CPB + k;
          if (has) {
            const float yh = (p[64 * i] - mean[br]) * rstd[br];
            const float g = dp[64 * i] * dgelu_erf(yh * gm[i] + bt[i]);
            dgm[i] += g * yh;
            dbt[i] += g;
            const float dyh = g * gm[i];
            s1[br] += dyh;
            s2[br] += dyh * yh;
          }
        }
    }
#pragma unroll
    for (int br = 0; br < 3; ++br) {
      s1[br] = wave_sum(s1[br]) * inv_cnt;
      s2[br] = wave_sum(s2[br]) * inv_cnt;
    }
    for (int t = 0; t < L; ++t) {
      const bool has = (t % dstride) == 0;
      const float* p = ybase + t * tstride;
      const float* dp = dbase + (t / dstride) * tstride;
      float* o = obase + t * tstride;
#pragma unroll
      for (int br = 0; br < 3; ++br)
#pragma unroll
        for (int k = 0; k < CPB; ++k) {
          const int i = br * CPB + k;
          const float yh = (p[64 * i] - mean[br]) * rstd[br];
          float dyh = 0.f;
          if (has) dyh = dp[64 * i] * dgelu_erf(yh * gm[i] + bt[i]) * gm[i];
          const float dyv = rstd[br] * (dyh - s1[br] - yh * s2[br]);
          o[64 * i] = dyv;
          dys[i] += dyv;                    // column sum of dy = gradient of the Conv1d bias in front of this norm
        }
    }
  }
#pragma unroll
  for (int i = 0; i < NCHK; ++i) {
    red[wave][lane + 64 * i] = dgm[i];
    red[wave][CT + lane + 64 * i] = dbt[i];
    red[wave][2 * CT + lane + 64 * i] = dys[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 3 * CT; c += 256)
    partials[(int64_t)blockIdx.x * 3 * CT + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}


// ------------------------------------------------------------------------------ GroupNorm(1) + GELU, register-resident
// Fast path for sequences that fit in registers: WPS waves (LPS = 64*WPS lanes) own one sequence, a lane holds
// NP <= NPMAX float4 "pairs" (t, channel quad) of it, so y (and dact) are read from HBM exactly once with
// 16-byte loads and every statistic is a register pass.  Pair j = l2 + LPS*i of lane l2 (i = 0..NP-1) is row
// t = j / QPR, quad = j % QPR (QPR = CT/4 quads per row).  3*LPS is a multiple of QPR for CT in {192, 384, 768},
// so a lane only ever meets three different channel quads (slot = i % 3): three gamma/beta quads and three
// parameter-gradient accumulators per lane.  Requires L*QPR % LPS == 0 and L*QPR/LPS <= NPMAX.
//   <WPS 4, NPMAX 9>: the default shapes (L 48 x 192 ch, L 24 x 384 ch): 9 pairs, ~150 VGPRs in the backward
//   <WPS 8, NPMAX 9>: twice the sequence length (L_in = 96), one 512-thread block per sequence
//   <WPS 2, NPMAX 18>: forward only, short sequences whose quad count is not a multiple of 256

// A quad of the (B, L, N, 3*Cout) conv tensors: fp32, or -- the kernels' OUTPUTS in bf16 mode (BASELINE configs[2]) --
// bf16: act and dy only feed bf16 contractions, which would round them in their loaders.  The input y stays fp32: these
// kernels are bound by requests in flight, not bytes, and 8-byte loads made them 20 % slower (measured).
// Offsets are in elements.
template <bool h16>
__device__ __forceinline__ float4 gn_ld4(const void* p, int64_t off) {
  if constexpr (h16) {
    const tecm_bf16x4 v = *reinterpret_cast<const tecm_bf16x4*>(reinterpret_cast<const __bf16*>(p) + off);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + off);
}
template <bool h16>
__device__ __forceinline__ void gn_st4(void* p, int64_t off, const float4& v) {
  if constexpr (h16) tecm_store_bf16x4(reinterpret_cast<__bf16*>(p) + off, v.x, v.y, v.z, v.w);
  else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + off) = v;
}

template <int CPB, int WPS>
struct GnGeom {
  static constexpr int CT = 192 * CPB;
  static constexpr int QPR = CT / 4;        // quads per row
  static constexpr int QPB = QPR / 3;       // quads per branch
  static constexpr int LPS = 64 * WPS;      // lanes per sequence
  static constexpr int NTHR = WPS > 4 ? 64 * WPS : 256;   // threads per block
  static constexpr int SPB = NTHR / LPS;    // sequences per block
  static constexpr int DT = LPS / QPR, DQ = LPS % QPR;
  static_assert((3 * LPS) % QPR == 0, "slot period");
};

__device__ __forceinline__ float sel3(int k, float a, float b, float c) { return k == 0 ? a : (k == 1 ? b : c); }
__device__ __forceinline__ float sum4(const float4& v) { return (v.x + v.y) + (v.z + v.w); }

// sums over the WPS waves of a sequence: xch[wave][3], waves of one sequence are consecutive
template <int WPS>
__device__ __forceinline__ void seq_reduce3(float (&v)[3], float (*xch)[3], int wave, int lane) {
#pragma unroll
  for (int b = 0; b < 3; ++b) v[b] = wave_sum(v[b]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int b = 0; b < 3; ++b) xch[wave][b] = v[b];
  }
  __syncthreads();
  const int w0 = wave / WPS * WPS;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    float t = xch[w0][b];
#pragma unroll
    for (int w = 1; w < WPS; ++w) t += xch[w0 + w][b];
    v[b] = t;
  }
}

template <int CPB, int WPS, int GN_NPMAX, bool IO16>
__global__ __launch_bounds__((GnGeom<CPB, WPS>::NTHR)) void gn_gelu_fwd_reg(const void* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, void* __restrict__ act,
                                                       float* __restrict__ stats, int B, int L, int N, float eps, int NP,
                                                       int astride) {
  // astride > 1: only the time steps t % astride == 0 are written, into a COMPACT (B, ceil(L / astride), N, CT) tensor --
  // the strided 1x1 conv behind the block (modules.py:36-41) reads nothing else, while the statistics need every step
  using G = GnGeom<CPB, WPS>;
  __shared__ float xch[G::NTHR / 64][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sp = wave / WPS, half = wave % WPS, l2 = half * 64 + lane;
  int64_t sidx = (int64_t)blockIdx.x * G::SPB + sp;
  const bool live = sidx < (int64_t)B * N;
  if (!live) sidx = (int64_t)B * N - 1;                  // keep the wave in the barriers; it stores nothing
  const int b = (int)(sidx / N), n = (int)(sidx - (int64_t)b * N);
  const int64_t tstride = (int64_t)N * G::CT;
  const int64_t base = ((int64_t)b * L * N + n) * G::CT;
  const float inv_cnt = 1.0f / (float)(L * CPB * 64);

  // the lane's three channel quads
  int qs[3], brs[3];
  {
    int q = l2 % G::QPR;
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      qs[s_] = q;
      brs[s_] = q / G::QPB;
      q += G::DQ;
      if (q >= G::QPR) q -= G::QPR;
    }
  }
  float4 v[GN_NPMAX];
  int32_t off[GN_NPMAX];                                 // relative to the sequence base (host checks L*N*CT < 2^31)
  int32_t aoff[GN_NPMAX];                                // the same in the (compact) act tensor; -1: this step is not written
  const int La = (L + astride - 1) / astride;
  const int64_t abase = ((int64_t)b * La * N + n) * G::CT;
  {
    int t = l2 / G::QPR, q = l2 % G::QPR;
#pragma unroll
    for (int i = 0; i < GN_NPMAX; ++i) {
      off[i] = t * (int32_t)tstride + q * 4;
      aoff[i] = (t % astride) == 0 ? (t / astride) * (int32_t)tstride + q * 4 : -1;
      if (i < NP) v[i] = gn_ld4<false>(y, base + off[i]);      // y stays fp32: 8-byte loads make the kernel slower, not faster
      t += G::DT;
      q += G::DQ;
      if (q >= G::QPR) { q -= G::QPR; ++t; }
    }
  }
  float4 gm[3], bt[3];
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_) {
    gm[s_] = *reinterpret_cast<const float4*>(gamma + qs[s_] * 4);
    bt[s_] = *reinterpret_cast<const float4*>(beta + qs[s_] * 4);
  }
  // mean
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GN_NPMAX; ++i)
    if (i < NP) acc[i % 3] += sum4(v[i]);
  float mean[3];
#pragma unroll
  for (int bb = 0; bb < 3; ++bb)
    mean[bb] = (brs[0] == bb ? acc[0] : 0.f) + (brs[1] == bb ? acc[1] : 0.f) + (brs[2] == bb ? acc[2] : 0.f);
  seq_reduce3<WPS>(mean, xch, wave, lane);
#pragma unroll
  for (int bb = 0; bb < 3; ++bb) mean[bb] *= inv_cnt;
  float ms[3];
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_) ms[s_] = sel3(brs[s_], mean[0], mean[1], mean[2]);
  // variance (two-pass form, like nn.GroupNorm)
  acc[0] = acc[1] = acc[2] = 0.f;
#pragma unroll
  for (int i = 0; i < GN_NPMAX; ++i)
    if (i < NP) {
      const float m = ms[i % 3];
      const float dx = v[i].x - m, dy_ = v[i].y - m, dz = v[i].z - m, dw = v[i].w - m;
      acc[i % 3] += (dx * dx + dy_ * dy_) + (dz * dz + dw * dw);
    }
  float rstd[3];
#pragma unroll
  for (int bb = 0; bb < 3; ++bb)
    rstd[bb] = (brs[0] == bb ? acc[0] : 0.f) + (brs[1] == bb ? acc[1] : 0.f) + (brs[2] == bb ? acc[2] : 0.f);
  seq_reduce3<WPS>(rstd, xch, wave, lane);
#pragma unroll
  for (int bb = 0; bb < 3; ++bb) rstd[bb] = 1.0f / sqrtf(rstd[bb] * inv_cnt + eps);
  float rs[3];
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_) rs[s_] = sel3(brs[s_], rstd[0], rstd[1], rstd[2]);
  if (live) {
#pragma unroll
    for (int i = 0; i < GN_NPMAX; ++i)
      if (i < NP && aoff[i] >= 0) {
        const int s_ = i % 3;
        float4 o;
        o.x = gelu_erf_fast((v[i].x - ms[s_]) * rs[s_] * gm[s_].x + bt[s_].x);
        o.y = gelu_erf_fast((v[i].y - ms[s_]) * rs[s_] * gm[s_].y + bt[s_].y);
        o.z = gelu_erf_fast((v[i].z - ms[s_]) * rs[s_] * gm[s_].z + bt[s_].z);
        o.w = gelu_erf_fast((v[i].w - ms[s_]) * rs[s_] * gm[s_].w + bt[s_].w);
        gn_st4<IO16>(act, abase + aoff[i], o);
      }
    if (half == 0 && lane < 3) {
      stats[(sidx * 3 + lane) * 2] = sel3(lane, mean[0], mean[1], mean[2]);
      stats[(sidx * 3 + lane) * 2 + 1] = sel3(lane, rstd[0], rstd[1], rstd[2]);
    }
  }
}

// Y16: y itself is bf16 (round 4; 8-byte loads: the same number of requests as the fp32 form, half the bytes -- the
// 8-channel kernels below issue a third of the requests per lane and run slower in this direction)
#ifdef GN_ABLATE_DGELU                                    // diagnostics (tools/build_variant.py): no transcendental in the backward
#define GN_DGELU(v) (0.5f + 0.1f * (v))
#else
#define GN_DGELU(v) dgelu_erf_fast(v)
#endif
template <int CPB, int WPS, int GN_NPMAX, bool IO16, bool D16 = false, bool Y16 = false>
__global__ __launch_bounds__((GnGeom<CPB, WPS>::NTHR)) void gn_gelu_bwd_reg(const void* __restrict__ dact, int dstride, int L2,
                                                       const void* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ stats,
                                                       void* __restrict__ dy, float* __restrict__ partials, int B,
                                                       int L, int N, int NP) {
  using G = GnGeom<CPB, WPS>;
  __shared__ float xch[G::NTHR / 64][3];
  __shared__ float red[3 * G::CT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sp = wave / WPS, half = wave % WPS, l2 = half * 64 + lane;
  const int64_t tstride = (int64_t)N * G::CT;
  const float inv_cnt = 1.0f / (float)(L * CPB * 64);
  for (int c = threadIdx.x; c < 3 * G::CT; c += G::NTHR) red[c] = 0.f;

  int qs[3], brs[3];
  {
    int q = l2 % G::QPR;
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      qs[s_] = q;
      brs[s_] = q / G::QPB;
      q += G::DQ;
      if (q >= G::QPR) q -= G::QPR;
    }
  }
  float4 gm[3], bt[3], dgm[3], dbt[3], dys[3];
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_) {
    gm[s_] = *reinterpret_cast<const float4*>(gamma + qs[s_] * 4);
    bt[s_] = *reinterpret_cast<const float4*>(beta + qs[s_] * 4);
    dgm[s_] = dbt[s_] = dys[s_] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t S = (int64_t)B * N;
  for (int64_t s0 = (int64_t)blockIdx.x * G::SPB; s0 < S; s0 += (int64_t)gridDim.x * G::SPB) {
    int64_t sidx = s0 + sp;
    const bool live = sidx < S;
    if (!live) sidx = S - 1;
    const int b = (int)(sidx / N), n = (int)(sidx - (int64_t)b * N);
    const int64_t ybase = ((int64_t)b * L * N + n) * G::CT;
    const int64_t dbase = ((int64_t)b * L2 * N + n) * G::CT;
    float mean[3], rstd[3];
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      mean[bb] = stats[(sidx * 3 + bb) * 2];
      rstd[bb] = stats[(sidx * 3 + bb) * 2 + 1];
    }
    float ms[3], rs[3];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      ms[s_] = sel3(brs[s_], mean[0], mean[1], mean[2]);
      rs[s_] = sel3(brs[s_], rstd[0], rstd[1], rstd[2]);
    }
    float4 yh[GN_NPMAX], gd[GN_NPMAX];                    // y (then y_hat) and dact (then d y_hat)
    int32_t off[GN_NPMAX];
    {
      int t = l2 / G::QPR, q = l2 % G::QPR;
#pragma unroll
      for (int i = 0; i < GN_NPMAX; ++i) {
        off[i] = t * (int32_t)tstride + q * 4;
        if (i < NP) {
          yh[i] = gn_ld4<Y16>(y, ybase + off[i]);
          const bool has = (t % dstride) == 0;
          // (D16: the gradient of the strided 1x1 conv's input arrives as the bf16 tensor its bf16 GEMM wrote)
          gd[i] = has ? gn_ld4<D16>(dact, dbase + (t / dstride) * (int32_t)tstride + q * 4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        t += G::DT;
        q += G::DQ;
        if (q >= G::QPR) { q -= G::QPR; ++t; }
      }
    }
    float a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < GN_NPMAX; ++i)
      if (i < NP) {
        const int s_ = i % 3;
#define GN_ELEM(c)                                                               \
  {                                                                              \
    const float h_ = (yh[i].c - ms[s_]) * rs[s_];                                \
    const float g_ = gd[i].c * GN_DGELU(h_ * gm[s_].c + bt[s_].c);                    \
    if (live) { dgm[s_].c += g_ * h_; dbt[s_].c += g_; }                         \
    const float d_ = g_ * gm[s_].c;                                              \
    a1[s_] += d_;                                                                \
    a2[s_] += d_ * h_;                                                           \
    yh[i].c = h_;                                                                \
    gd[i].c = d_;                                                                \
  }
        GN_ELEM(x) GN_ELEM(y) GN_ELEM(z) GN_ELEM(w)
#undef GN_ELEM
      }
    float s1[3], s2[3];
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      s1[bb] = (brs[0] == bb ? a1[0] : 0.f) + (brs[1] == bb ? a1[1] : 0.f) + (brs[2] == bb ? a1[2] : 0.f);
      s2[bb] = (brs[0] == bb ? a2[0] : 0.f) + (brs[1] == bb ? a2[1] : 0.f) + (brs[2] == bb ? a2[2] : 0.f);
    }
    seq_reduce3<WPS>(s1, xch, wave, lane);
    seq_reduce3<WPS>(s2, xch, wave, lane);
    float m1[3], m2[3];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      m1[s_] = sel3(brs[s_], s1[0], s1[1], s1[2]) * inv_cnt;
      m2[s_] = sel3(brs[s_], s2[0], s2[1], s2[2]) * inv_cnt;
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < GN_NPMAX; ++i)
        if (i < NP) {
          const int s_ = i % 3;
          float4 o;
          o.x = rs[s_] * (gd[i].x - m1[s_] - yh[i].x * m2[s_]);
          o.y = rs[s_] * (gd[i].y - m1[s_] - yh[i].y * m2[s_]);
          o.z = rs[s_] * (gd[i].z - m1[s_] - yh[i].z * m2[s_]);
          o.w = rs[s_] * (gd[i].w - m1[s_] - yh[i].w * m2[s_]);
          gn_st4<IO16>(dy, ybase + off[i], o);
          dys[s_].x += o.x; dys[s_].y += o.y; dys[s_].z += o.z; dys[s_].w += o.w;   // conv-bias gradient
        }
    }
  }
  // block reduction of the per-lane parameter gradients (each channel quad is held by NTHR*3/QPR (lane, slot) pairs)
  // -> one partial row.  In a FIXED order, so that the gradients repeat bit for bit: for a given slot the holders of
  // one quad are the lanes with the same l2 % QPR, ranked by (sequence of the block, l2 / QPR); in round (slot, rank)
  // every quad has at most one writer, which adds with a plain read-modify-write.  (LDS float atomics here made
  // d gamma / d beta / d conv-bias order-dependent in the last bits.)
  {
    constexpr int RMAX = (G::LPS + G::QPR - 1) / G::QPR;
    const int rank = sp * RMAX + l2 / G::QPR;
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      const int c = qs[s_] * 4;
      for (int r = 0; r < G::SPB * RMAX; ++r) {
        __syncthreads();
        if (r == rank) {
          red[c + 0] += dgm[s_].x; red[c + 1] += dgm[s_].y; red[c + 2] += dgm[s_].z; red[c + 3] += dgm[s_].w;
          red[G::CT + c + 0] += dbt[s_].x; red[G::CT + c + 1] += dbt[s_].y;
          red[G::CT + c + 2] += dbt[s_].z; red[G::CT + c + 3] += dbt[s_].w;
          red[2 * G::CT + c + 0] += dys[s_].x; red[2 * G::CT + c + 1] += dys[s_].y;
          red[2 * G::CT + c + 2] += dys[s_].z; red[2 * G::CT + c + 3] += dys[s_].w;
        }
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 3 * G::CT; c += G::NTHR) partials[(int64_t)blockIdx.x * 3 * G::CT + c] = red[c];
}

// ------------------------------------------------------------------------------ GroupNorm(1) + GELU on a bf16 y (round 4)
// bf16 mode: the conv output y is the bf16 tensor a bf16 Conv1d hands to the fp32 GroupNorm under autocast (train.py:68;
// conv_fwd_seq writes it as such), and every tensor these kernels touch is bf16 (y, act, dact, dy).  The register-resident
// kernels above are bound by memory requests in flight, not bytes -- fed 8-byte quads they got slower, not faster -- so here
// a lane's unit is an OCT (8 consecutive channels = one 16-byte load / store): the same number of requests per lane, twice
// the elements.  (Geometry: GnGeom8 below.)  The sequence need not fill the last round of lanes: rows past L are skipped.
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x8 gn_ld8(const void* p, int64_t off) {
  const bf16x8n v = *reinterpret_cast<const bf16x8n*>(reinterpret_cast<const __bf16*>(p) + off);
  f32x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (float)v[e];
  return r;
}
__device__ __forceinline__ void gn_st8(void* p, int64_t off, const f32x8& v) {
  bf16x8n h;
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
  *reinterpret_cast<bf16x8n*>(reinterpret_cast<__bf16*>(p) + off) = h;
}
__device__ __forceinline__ f32x8 gn_ld8f(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f32x8 r = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return r;
}
__device__ __forceinline__ float sum8(const f32x8& v) { return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); }

// Geometry: SIX waves (384 lanes) own one sequence and a lane keeps ONE channel oct for the whole kernel (384 is a multiple
// of OPR = 24, 48, 96): lane l2 holds oct l2 % OPR of the rows t = l2 / OPR + k * (384 / OPR), k < NG.  One gamma / beta oct
// and one set of parameter-gradient accumulators per lane (the three-slot form of the fp32-y kernels needs 72 + 48
// registers for them at 8 channels per unit and ran at one wave per SIMD).
template <int CPB>
struct GnGeom8 {
  static constexpr int CT = 192 * CPB;
  static constexpr int OPR = CT / 8;        // octs per row
  static constexpr int OPB = OPR / 3;       // octs per branch
  static constexpr int LPS = 384;           // lanes per sequence = threads per block
  static constexpr int DT = LPS / OPR;      // rows between a lane's consecutive octs
  static_assert(LPS % OPR == 0, "a lane keeps its channel oct");
};

// sums over the six waves of the block; branch br of this lane's contribution goes to slot br
__device__ __forceinline__ void seq_reduce3x6(float (&v)[3], float (*xch)[3], int wave, int lane) {
#pragma unroll
  for (int b = 0; b < 3; ++b) v[b] = wave_sum(v[b]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int b = 0; b < 3; ++b) xch[wave][b] = v[b];
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 3; ++b) v[b] = ((xch[0][b] + xch[1][b]) + (xch[2][b] + xch[3][b])) + (xch[4][b] + xch[5][b]);
}

template <int CPB, int NG>
__global__ __launch_bounds__(384) void gn_gelu_fwd_reg16(const void* __restrict__ y, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, void* __restrict__ act,
                                                         float* __restrict__ stats, int B, int L, int N, float eps,
                                                         int astride) {
  using G = GnGeom8<CPB>;
  __shared__ float xch[6][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l2 = threadIdx.x;
  const int64_t sidx = blockIdx.x;                        // one sequence per block
  const int b = (int)(sidx / N), n = (int)(sidx - (int64_t)b * N);
  const int64_t tstride = (int64_t)N * G::CT;
  const int64_t base = ((int64_t)b * L * N + n) * G::CT;
  const int La = (L + astride - 1) / astride;
  const int64_t abase = ((int64_t)b * La * N + n) * G::CT;
  const float inv_cnt = 1.0f / (float)(L * CPB * 64);
  const int q = l2 % G::OPR, t0 = l2 / G::OPR, br = q / G::OPB;
  f32x8 v[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    const int t = t0 + k * G::DT;
    const int tc = t < L ? t : L - 1;                      // clamped: an unconditional load, the value is not used
    v[k] = gn_ld8(y, base + tc * (int32_t)tstride + q * 8);
    if (t >= L)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
  }
  const f32x8 gm = gn_ld8f(gamma + q * 8), bt = gn_ld8f(beta + q * 8);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < NG; ++k) acc += sum8(v[k]);          // absent octs hold zeros
  float mean[3] = {br == 0 ? acc : 0.f, br == 1 ? acc : 0.f, br == 2 ? acc : 0.f};
  seq_reduce3x6(mean, xch, wave, lane);
#pragma unroll
  for (int bb = 0; bb < 3; ++bb) mean[bb] *= inv_cnt;
  const float m = sel3(br, mean[0], mean[1], mean[2]);
  acc = 0.f;
#pragma unroll
  for (int k = 0; k < NG; ++k)
    if (t0 + k * G::DT < L) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[k][e] - m;
        acc += d * d;
      }
    }
  float rstd[3] = {br == 0 ? acc : 0.f, br == 1 ? acc : 0.f, br == 2 ? acc : 0.f};
  seq_reduce3x6(rstd, xch, wave, lane);
#pragma unroll
  for (int bb = 0; bb < 3; ++bb) rstd[bb] = 1.0f / sqrtf(rstd[bb] * inv_cnt + eps);
  const float rs = sel3(br, rstd[0], rstd[1], rstd[2]);
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    const int t = t0 + k * G::DT;
    if (t < L && (t % astride) == 0) {
      f32x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = gelu_erf_fast((v[k][e] - m) * rs * gm[e] + bt[e]);
      gn_st8(act, abase + (t / astride) * (int32_t)tstride + q * 8, o);
    }
  }
  if (l2 < 3) {
    stats[(sidx * 3 + l2) * 2] = sel3(l2, mean[0], mean[1], mean[2]);
    stats[(sidx * 3 + l2) * 2 + 1] = sel3(l2, rstd[0], rstd[1], rstd[2]);
  }
}

template <int CPB, int NG>
#ifndef GN16_OCC
#define GN16_OCC 3       // waves per SIMD the 3-oct backward is compiled for (168 registers, ~8 spilled; 2 = 176, none)
#endif
__global__ __launch_bounds__(384, (NG == 3 ? GN16_OCC : 2)) void gn_gelu_bwd_reg16(const void* __restrict__ dact, int dstride, int L2,
                                                         const void* __restrict__ y, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ stats,
                                                         void* __restrict__ dy, float* __restrict__ partials, int B,
                                                         int L, int N) {
  using G = GnGeom8<CPB>;
  __shared__ float xch[6][3];
  __shared__ float red[3 * G::CT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l2 = threadIdx.x;
  const int64_t tstride = (int64_t)N * G::CT;
  const float inv_cnt = 1.0f / (float)(L * CPB * 64);
  const int q = l2 % G::OPR, t0 = l2 / G::OPR, br = q / G::OPB;
  const f32x8 gm = gn_ld8f(gamma + q * 8), bt = gn_ld8f(beta + q * 8);
  f32x8 dgm, dbt, dys;
#pragma unroll
  for (int e = 0; e < 8; ++e) dgm[e] = dbt[e] = dys[e] = 0.f;
  const int64_t S = (int64_t)B * N;
  for (int64_t sidx = blockIdx.x; sidx < S; sidx += gridDim.x) {
    const int b = (int)(sidx / N), n = (int)(sidx - (int64_t)b * N);
    const int64_t ybase = ((int64_t)b * L * N + n) * G::CT;
    const int64_t dbase = ((int64_t)b * L2 * N + n) * G::CT;
    const float m = stats[(sidx * 3 + br) * 2], rs = stats[(sidx * 3 + br) * 2 + 1];
    f32x8 yh[NG], gd[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int t = t0 + k * G::DT;
      const int tc = t < L ? t : L - 1;
      yh[k] = gn_ld8(y, ybase + tc * (int32_t)tstride + q * 8);
      gd[k] = gn_ld8(dact, dbase + (tc / dstride) * (int32_t)tstride + q * 8);     // unconditional (clamped), masked below
      if (t >= L || (tc % dstride) != 0)
#pragma unroll
        for (int e = 0; e < 8; ++e) gd[k][e] = 0.f;
    }
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const bool has = t0 + k * G::DT < L;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float h_ = (yh[k][e] - m) * rs;
        const float g_ = gd[k][e] * dgelu_erf_fast(h_ * gm[e] + bt[e]);            // gd = 0 where there is no gradient
        dgm[e] += g_ * h_;
        dbt[e] += g_;
        const float d_ = g_ * gm[e];
        a1 += d_;
        a2 += d_ * h_;
        yh[k][e] = has ? h_ : 0.f;
        gd[k][e] = d_;
      }
    }
    float s1[3] = {br == 0 ? a1 : 0.f, br == 1 ? a1 : 0.f, br == 2 ? a1 : 0.f};
    float s2[3] = {br == 0 ? a2 : 0.f, br == 1 ? a2 : 0.f, br == 2 ? a2 : 0.f};
    seq_reduce3x6(s1, xch, wave, lane);
    seq_reduce3x6(s2, xch, wave, lane);
    const float m1 = sel3(br, s1[0], s1[1], s1[2]) * inv_cnt, m2 = sel3(br, s2[0], s2[1], s2[2]) * inv_cnt;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int t = t0 + k * G::DT;
      if (t < L) {
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = rs * (gd[k][e] - m1 - yh[k][e] * m2);
          dys[e] += o[e];                                  // conv-bias gradient
        }
        gn_st8(dy, ybase + t * (int32_t)tstride + q * 8, o);
      }
    }
  }
  // block reduction in a FIXED order: the holders of one oct are the lanes l2 = q + OPR * j, j = 0 .. DT-1, taken in turn
  for (int c = threadIdx.x; c < 3 * G::CT; c += 384) red[c] = 0.f;
  for (int j = 0; j < G::DT; ++j) {
    __syncthreads();
    if (t0 == j) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[q * 8 + e] += dgm[e];
        red[G::CT + q * 8 + e] += dbt[e];
        red[2 * G::CT + q * 8 + e] += dys[e];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 3 * G::CT; c += 384) partials[(int64_t)blockIdx.x * 3 * G::CT + c] = red[c];
}

// ------------------------------------------------------------------------------ column sums
// stage 1: grid (colblocks, RB, nseg).  Lane = column, the 4 waves stride the block's row chunk.
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ in, int64_t ld, int64_t outer,
                                                     int64_t inner, int nseg, int C, DropCtxN idc,
                                                     float* __restrict__ ws, int64_t chunk) {
  seed_now(idc);
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int rb = blockIdx.y, s = blockIdx.z;
  const int64_t total = outer * inner;
  const int64_t beg = (int64_t)rb * chunk;
  const int64_t end = beg + chunk < total ? beg + chunk : total;
  float acc = 0.f;
  if (c < C) {
    float a4[4] = {0.f, 0.f, 0.f, 0.f};          // four rows in flight per wave: the loop is latency-bound otherwise
    for (int64_t r0 = beg + wave; r0 < end; r0 += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t ri = r0 + 4 * u;
        if (ri < end) {
          const int64_t o = ri / inner, j = ri - o * inner;
          const int64_t row = (o * nseg + s) * inner + j;
          float v = in[row * ld + c];
          if (idc.thresh) v *= tecm_drop_mult(idc.seed, (uint64_t)(row * idc.ld + c), idc.thresh, idc.inv);
          a4[u] += v;
        }
      }
    }
    acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < C)
    ws[((int64_t)s * gridDim.y + rb) * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// stage 1, float4 form for 16-byte friendly inputs: a block owns a chunk of rows and ALL C columns,
// thread = (row slot, column quad), so a wave reads whole 1 KiB runs of consecutive rows instead of 256-byte
// slivers (narrow matrices -- 24 .. 128 columns -- ran at 0.3 .. 1 TB/s through the lane-per-column kernel).
__global__ __launch_bounds__(256) void colsum_stage1_v4(const float* __restrict__ in, int64_t ld, int64_t outer,
                                                        int64_t inner, int nseg, int C, DropCtxN idc,
                                                        float* __restrict__ ws, int64_t chunk,
                                                        __bf16* __restrict__ twin = nullptr, int64_t ld_twin = 0) {
  seed_now(idc);
  __shared__ float4 red[256];
  const int Q = C >> 2;                       // quads per row (host: Q <= 256)
  const int RPB = 256 / Q;                    // row slots per pass
  const int q = threadIdx.x % Q, rs = threadIdx.x / Q;
  const int rb = blockIdx.x, s = blockIdx.y;
  const int64_t total = outer * inner;
  const int64_t beg = (int64_t)rb * chunk;
  const int64_t end = beg + chunk < total ? beg + chunk : total;
  float4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rs < RPB) {
    for (int64_t r0 = beg + rs; r0 < end; r0 += 4 * RPB) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t ri = r0 + (int64_t)u * RPB;
        if (ri < end) {
          int64_t row = ri;
          if (nseg > 1) {
            const int64_t o = ri / inner, j = ri - o * inner;
            row = (o * nseg + s) * inner + j;
          }
          float4 v = *reinterpret_cast<const float4*>(in + row * ld + 4 * q);
          if (idc.thresh) {
            const uint64_t di = (uint64_t)(row * idc.ld + 4 * q);
            v.x *= tecm_drop_mult(idc.seed, di, idc.thresh, idc.inv);
            v.y *= tecm_drop_mult(idc.seed, di + 1, idc.thresh, idc.inv);
            v.z *= tecm_drop_mult(idc.seed, di + 2, idc.thresh, idc.inv);
            v.w *= tecm_drop_mult(idc.seed, di + 3, idc.thresh, idc.inv);
          }
          // every row is visited exactly once: the (masked) values can leave as a bf16 twin on the way (tecm_colsum_twin)
          if (twin) tecm_store_bf16x4(twin + row * ld_twin + 4 * q, v.x, v.y, v.z, v.w);
          a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
        }
      }
    }
  }
  float4 t;
  t.x = (a[0].x + a[1].x) + (a[2].x + a[3].x);
  t.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
  t.z = (a[0].z + a[1].z) + (a[2].z + a[3].z);
  t.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
  red[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x < Q) {
    float4 acc = red[threadIdx.x];
    for (int r = 1; r < RPB; ++r) {
      const float4 v = red[r * Q + threadIdx.x];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(ws + ((int64_t)s * gridDim.x + rb) * C + 4 * threadIdx.x) = acc;
  }
}

// stage 2: 512 threads = 8 waves per 64 columns, 32 loads in flight per lane: 1024 partial rows are four rounds of
// memory latency (the first version -- 4 waves, 16 in flight -- took sixteen: 23 us per launch, 68 launches a step).
// WAVES = 16 (round 5): the narrow matrices (C <= 128: one or two column blocks for up to 1024 partial rows) take ONE round
// of latency instead of four -- 17-23 us -> see DESIGN_HISTORY B.11; the summation order differs between the two forms
// only in the association of the partial rows, fixed per (RB, WAVES): results stay run-to-run identical.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void colsum_stage2(const float* __restrict__ ws, int RB, int nseg, int C,
                                                            float* __restrict__ out, int64_t ldo, int accumulate,
                                                            float scale) {
  constexpr int UNR = WAVES == 16 ? 64 : 32;            // loads in flight per lane
  __shared__ float red[WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, s = blockIdx.y;
  float acc = 0.f;
  if (c < C) {
    float a32[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) a32[u] = 0.f;
    for (int rb = wave; rb < RB; rb += WAVES * UNR) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {                   // clamped row + select, not `if`: the bound is wave-uniform, a branch
        const int r = rb + WAVES * u;                   // per load would put a wait between every two of them
        const float v = ws[((int64_t)s * RB + (r < RB ? r : RB - 1)) * C + c];
        a32[u] += r < RB ? v : 0.f;
      }
    }
#pragma unroll
    for (int u = UNR / 2; u > 0; u >>= 1)
#pragma unroll
      for (int v = 0; v < u; ++v) a32[v] += a32[v + u];
    acc = a32[0];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < C) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; w += 4) v += (red[w][lane] + red[w + 1][lane]) + (red[w + 2][lane] + red[w + 3][lane]);
    v *= scale;
    float* o = out + (int64_t)s * ldo + c;
    *o = accumulate ? *o + v : v;
  }
}
void launch_colsum_stage2(const float* ws, int RB, int nseg, int C, float* out, int64_t ldo, int accumulate, float scale,
                          int colblocks, hipStream_t st) {
  if (colblocks * nseg <= 4 && RB > 256)
    hipLaunchKernelGGL(colsum_stage2<16>, dim3(colblocks, nseg), dim3(1024), 0, st, ws, RB, nseg, C, out, ldo, accumulate, scale);
  else
    hipLaunchKernelGGL(colsum_stage2<8>, dim3(colblocks, nseg), dim3(512), 0, st, ws, RB, nseg, C, out, ldo, accumulate, scale);
}

int ln_blocks(int64_t M) {
  const int64_t want = (M + 3) / 4;
  return (int)(want < 1024 ? want : 1024);
}
// pairs per lane of the register-resident GroupNorm path with `wps` waves per sequence, 0 when the sequence does
// not fit / is not 16-byte friendly
int gn_reg_pairs(int L, int N, int Cout, int wps, int npmax, const void* a, const void* b) {
  const int64_t quads = (int64_t)L * (3 * Cout / 4);
  const int lps = 64 * wps;
  if ((int64_t)L * N * 3 * Cout >= (1ll << 31)) return 0;      // 32-bit offsets inside one sample
  if (quads % lps != 0 || quads / lps > npmax) return 0;
  if (!tecm_aligned(a, 16) || !tecm_aligned(b, 16)) return 0;
  return (int)(quads / lps);
}
// octs per lane of the all-bf16 kernels (3 or 6; 384 lanes per sequence), 0 when the sequence does not fit
int gn16_ng(int L, int N, int Cout, const void* a, const void* b) {
  if (Cout != 64 && Cout != 128 && Cout != 256) return 0;
  const int64_t octs = (int64_t)L * (3 * Cout / 8);
  if ((int64_t)L * N * 3 * Cout >= (1ll << 31)) return 0;
  if ((a && !tecm_aligned(a, 16)) || (b && !tecm_aligned(b, 16))) return 0;
  if (octs <= 384 * 3) return 3;
  if (octs <= 384 * 6) return 6;
  return 0;
}
// GroupNorm(1) + GELU forward with the statistics GIVEN (conv_fwd_seq computes them from the y it holds in registers): what is
// left is elementwise and needs only the time steps the strided 1x1 conv reads -- one oct (8 channels) per lane per visit,
// every access a 16-byte one, no sequence held anywhere, no synchronisation.  act is the compact (B, L / stride, N, CT) tensor.
__global__ __launch_bounds__(256) void gn_apply_fwd16_kernel(const void* __restrict__ y, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, void* __restrict__ act,
                                                             const float* __restrict__ stats, int B, int L, int N, int Cout,
                                                             int stride) {
  const int CT = 3 * Cout, OPR = CT / 8;                       // octs per row
  const int Lo = (L + stride - 1) / stride;
  const int64_t total = (int64_t)B * Lo * N * OPR;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int o = (int)(i % OPR);
    const int64_t row = i / OPR;                               // (b, to, n)
    const int n = (int)(row % N);
    const int64_t bt = row / N;
    const int to = (int)(bt % Lo), b = (int)(bt / Lo);
    const int c = o * 8, br = c / Cout;
    const float2 ms = *reinterpret_cast<const float2*>(stats + (((int64_t)b * N + n) * 3 + br) * 2);
    const f32x8 v = gn_ld8(y, (((int64_t)b * L + (int64_t)to * stride) * N + n) * CT + c);
    const f32x8 g = gn_ld8f(gamma + c), be = gn_ld8f(beta + c);
    f32x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = gelu_erf_fast((v[e] - ms.x) * ms.y * g[e] + be[e]);
    gn_st8(act, row * CT + c, r);
  }
}

// ------------------------------------------------------------------ GroupNorm(1) + GELU backward on bf16 tensors, split (round 4)
// The sequence-resident backward (a block holds one sequence in registers: load, reduce, barrier, reduce, barrier, write)
// runs at 2.3 TB/s of its 1.07 GB -- two waves per SIMD, every phase exposed.  The same arithmetic as TWO streaming kernels
// with no resident sequence:
//   sums   a block owns NPB nodes of one sample and walks the time steps the strided 1x1 conv read (the only ones with a
//          gradient); a lane keeps one (node, channel oct): sum d and sum d * y_hat per (sequence, branch), d gamma / d beta
//          per channel in registers across the block's items.  Reads y (those steps) + dact: 0.43 GB.
//   apply  elementwise: dy = rstd * (d - mean(d) - y_hat * mean(d y_hat)), d recomputed (GELU' a second time is cheaper than
//          a resident sequence); a lane keeps its channel oct, so the conv-bias gradient (column sums of dy) stays in
//          registers.  Reads y + dact, writes dy: 1.07 GB.
// Both deterministic (fixed lane -> channel map, per-block partial rows summed by tecm_colsum).
template <int CPB>
__global__ __launch_bounds__(256, 4) void gn_bwd_sums16_kernel(const void* __restrict__ dact, int dstride, int L2,
                                                            const void* __restrict__ y, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ stats,
                                                            float* __restrict__ sums, float* __restrict__ partials, int B,
                                                            int L, int N, int active) {
  constexpr int CT = 192 * CPB, OPR = CT / 8, OPB = OPR / 3, NPB = 256 / OPR;
  __shared__ float red[NPB][OPR][2];
  __shared__ float pg[2][NPB][CT];
  const int nl = threadIdx.x / OPR, o = threadIdx.x - nl * OPR;
  const bool lane_on = nl < NPB;
  const int c = o * 8, br = o / OPB;
  const f32x8 gm = lane_on ? gn_ld8f(gamma + c) : f32x8{}, bt = lane_on ? gn_ld8f(beta + c) : f32x8{};
  f32x8 dg = {}, db = {};
  const int nblk = (N + NPB - 1) / NPB;
  const int items = B * nblk;
  // `active` blocks share the items evenly (the rest of the grid only writes its zero partial row)
  for (int item = blockIdx.x < active ? blockIdx.x : items; item < items; item += active) {
    const int b = item / nblk, n = (item - b * nblk) * NPB + nl;
    const bool on = lane_on && n < N;
    const int nn = on ? n : 0;
    const float2 ms = *reinterpret_cast<const float2*>(stats + (((int64_t)b * N + nn) * 3 + (lane_on ? br : 0)) * 2);
    float a1 = 0.f, a2 = 0.f;
    if (on) {
      const int64_t yb = ((int64_t)b * L * N + nn) * CT + c, dbs = ((int64_t)b * L2 * N + nn) * CT + c;
      const int64_t ystep = (int64_t)dstride * N * CT, dstep = (int64_t)N * CT;
#pragma unroll 4
      for (int t2 = 0; t2 < L2; ++t2) {
        const f32x8 yv = gn_ld8(y, yb + t2 * ystep), dv = gn_ld8(dact, dbs + t2 * dstep);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float h_ = (yv[e] - ms.x) * ms.y;
          const float g_ = dv[e] * GN_DGELU(h_ * gm[e] + bt[e]);
          dg[e] += g_ * h_;
          db[e] += g_;
          const float d_ = g_ * gm[e];
          a1 += d_;
          a2 += d_ * h_;
        }
      }
    }
    __syncthreads();                                          // red is free again
    if (lane_on) { red[nl][o][0] = a1; red[nl][o][1] = a2; }
    __syncthreads();
    if (threadIdx.x < NPB * 3) {                              // (node, branch): OPB octs in a fixed order
      const int q = threadIdx.x / 3, bb = threadIdx.x - q * 3;
      const int n2 = (item - b * nblk) * NPB + q;
      float t1 = 0.f, t2 = 0.f;
      for (int k = 0; k < OPB; ++k) { t1 += red[q][bb * OPB + k][0]; t2 += red[q][bb * OPB + k][1]; }
      if (n2 < N) {
        float* so = sums + (((int64_t)b * N + n2) * 3 + bb) * 2;
        so[0] = t1;
        so[1] = t2;
      }
    }
  }
  __syncthreads();
  if (lane_on) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { pg[0][nl][c + e] = dg[e]; pg[1][nl][c + e] = db[e]; }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * CT; k += 256) {
    const int w = k / CT, cc = k - w * CT;
    float t = 0.f;
    for (int q = 0; q < NPB; ++q) t += pg[w][q][cc];
    partials[(int64_t)blockIdx.x * 3 * CT + k] = t;
  }
}

template <int CPB>
__global__ __launch_bounds__(256) void gn_bwd_apply16_kernel(const void* __restrict__ dact, int dstride, int L2,
                                                             const void* __restrict__ y, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ stats,
                                                             const float* __restrict__ sums, void* __restrict__ dy,
                                                             float* __restrict__ partials, int B, int L, int N) {
  constexpr int CT = 192 * CPB, OPR = CT / 8, OPB = OPR / 3, RPB = 256 / OPR;
  __shared__ float pg[RPB][CT];
  const int rl = threadIdx.x / OPR, o = threadIdx.x - rl * OPR;
  const bool lane_on = rl < RPB;
  const int c = o * 8, br = o / OPB;
  const f32x8 gm = lane_on ? gn_ld8f(gamma + c) : f32x8{}, bt = lane_on ? gn_ld8f(beta + c) : f32x8{};
  const float inv_cnt = 1.0f / (float)(L * CPB * 64);
  f32x8 dys = {};
  const int64_t rows = (int64_t)B * L * N;
  if (lane_on)
    for (int64_t row = (int64_t)blockIdx.x * RPB + rl; row < rows; row += (int64_t)gridDim.x * RPB) {
      const int n = (int)(row % N);
      const int64_t bt_ = row / N;
      const int t = (int)(bt_ % L), b = (int)(bt_ / L);
      const int64_t si = (((int64_t)b * N + n) * 3 + br) * 2;
      const float2 ms = *reinterpret_cast<const float2*>(stats + si);
      const float2 sm = *reinterpret_cast<const float2*>(sums + si);
      const float m1 = sm.x * inv_cnt, m2 = sm.y * inv_cnt;
      const f32x8 yv = gn_ld8(y, row * CT + c);
      const bool has = (t % dstride) == 0;
      f32x8 dv = {};
      if (has) dv = gn_ld8(dact, (((int64_t)b * L2 + t / dstride) * N + n) * CT + c);
      f32x8 ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float h_ = (yv[e] - ms.x) * ms.y;
        const float d_ = has ? dv[e] * GN_DGELU(h_ * gm[e] + bt[e]) * gm[e] : 0.f;
        ov[e] = ms.y * (d_ - m1 - h_ * m2);
        dys[e] += ov[e];
      }
      gn_st8(dy, row * CT + c, ov);
    }
  if (lane_on) {
#pragma unroll
    for (int e = 0; e < 8; ++e) pg[rl][c + e] = dys[e];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < CT; k += 256) {
    float t = 0.f;
    for (int q = 0; q < RPB; ++q) t += pg[q][k];
    partials[(int64_t)blockIdx.x * 3 * CT + 2 * CT + k] = t;
  }
}

int gn_blocks(int64_t S) {
  const int64_t want = (S + 3) / 4;
  return (int)(want < 1024 ? want : 1024);
}

}  // namespace

// 1 when the all-bf16 GroupNorm kernels (TECM_GN_Y_BF16) serve sequences of L steps x 3*Cout channels, else 0
extern "C" int tecm_gn_y16_supported(int32_t L, int32_t N, int32_t Cout) { return gn16_ng(L, N, Cout, nullptr, nullptr) > 0 ? 1 : 0; }

extern "C" int tecm_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                                  int64_t ldy, void* y16, int64_t ldy16, void* y16d, int64_t ldy16d, const TecmDrop* drop,
                                  int32_t y16d_seq_T, int32_t y16d_seq_N, float* stats, int64_t M, int32_t D, float eps,
                                  void* stream) {
  TECM_REQUIRE(x && gamma && beta && (y || y16 || y16d) && stats, TECM_E_ARG, "tecm_layernorm_fwd: null pointer");
  TECM_REQUIRE(y16d_seq_T == 0 || (y16d && y16d_seq_T > 0 && y16d_seq_N > 0 && M % ((int64_t)y16d_seq_T * y16d_seq_N) == 0),
               TECM_E_ARG, "tecm_layernorm_fwd: the sequence-major form needs y16d and M = B * T * N");
  TECM_REQUIRE(!y16d || (tecm_aligned(y16d, 8) && ldy16d % 4 == 0 && ldy16d >= D), TECM_E_ALIGN,
               "tecm_layernorm_fwd: the dropped bf16 output must be 8-byte aligned with a leading dimension multiple of 4");
  TECM_REQUIRE(!y16d || (drop && drop->p >= 0.f && drop->p < 1.f), TECM_E_ARG, "tecm_layernorm_fwd: y16d needs its dropout spec");
  const DropCtxN dd = make_dropn(y16d ? drop : nullptr);
  TECM_REQUIRE(!y16 || (tecm_aligned(y16, 8) && ldy16 % 4 == 0 && ldy16 >= D), TECM_E_ALIGN,
               "tecm_layernorm_fwd: the bf16 output must be 8-byte aligned with a leading dimension multiple of 4");
  TECM_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAXCH, TECM_E_ARG,
               "tecm_layernorm_fwd: need D %% 4 == 0 and D <= %d (got %d)", 256 * LN_MAXCH, D);
  TECM_REQUIRE(ldx % 4 == 0 && (!y || (ldy % 4 == 0 && tecm_aligned(y, 16))) && tecm_aligned(x, 16) &&
                   tecm_aligned(gamma, 16) && tecm_aligned(beta, 16),
               TECM_E_ALIGN, "tecm_layernorm_fwd: 16-byte alignment required");
  const dim3 grid((unsigned)((M + 3) / 4));
  const int nch = (D + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
#define LN_FWD(NCH) \
  hipLaunchKernelGGL((layernorm_fwd_kernel<NCH>), grid, dim3(256), 0, st, x, ldx, gamma, beta, y, ldy, y16, ldy16, y16d, ldy16d, \
                     dd, stats, M, D, eps, (int)y16d_seq_T, (int)y16d_seq_N)
  switch (nch) {
    case 1: LN_FWD(1); break;
    case 2: LN_FWD(2); break;
    case 3: LN_FWD(3); break;
    default: LN_FWD(4); break;
  }
#undef LN_FWD
  TECM_CHECK_LAUNCH("tecm_layernorm_fwd");
  return TECM_OK;
}

extern "C" int tecm_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                  const float* stats, const float* dres, float* dx, void* dx_masked,
                                  int32_t masked_bf16, const TecmDrop* mask_drop, float* dgb_partials,
                                  int32_t* num_blocks, int64_t M, int32_t D, const TecmLnAdd* add, int32_t dy_bf16,
                                  const TecmLnDyMap* dymap, void* stream) {
  TECM_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAXCH, TECM_E_ARG, "tecm_layernorm_bwd: bad M/D");
  const int nb = ln_blocks(M);
  if (num_blocks) *num_blocks = nb;
  if (dx == nullptr) return TECM_OK;   // query mode
  TECM_REQUIRE(dy && x && gamma && stats && dgb_partials, TECM_E_ARG, "tecm_layernorm_bwd: null pointer");
  TECM_REQUIRE(lddy % 4 == 0 && ldx % 4 == 0 && tecm_aligned(dy, dy_bf16 ? 8 : 16) && tecm_aligned(x, 16) &&
                   tecm_aligned(dx, 16) && (!dres || tecm_aligned(dres, 16)) &&
                   (!dx_masked || tecm_aligned(dx_masked, 16)),
               TECM_E_ALIGN, "tecm_layernorm_bwd: 16-byte alignment required");
  const DropCtxN odc = make_dropn(mask_drop);
  const int nch = (D + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  LnAdd ad{};
  const bool has_add = add != nullptr && add->dy2 != nullptr;
  if (has_add) {
    TECM_REQUIRE(add->ld % 4 == 0 && add->ld >= D && tecm_aligned(add->dy2, add->bf16 ? 8 : 16), TECM_E_ALIGN,
                 "tecm_layernorm_bwd: dy2 must be 16-byte (fp32) / 8-byte (bf16) friendly");
    ad.dy2 = add->dy2; ad.ld = add->ld; ad.bf16 = add->bf16; ad.drop = make_dropn(&add->drop);
  }
 LnDyMap dm{};
  const bool has_map = dymap != nullptr && dymap->T > 0;
  if (has_map) {
    TECM_REQUIRE(dy_bf16 && !has_add && dymap->N > 0 && M % ((int64_t)dymap->T * dymap->N) == 0, TECM_E_ARG,
                 "tecm_layernorm_bwd: the sequence-major dy is a bf16 matrix of M = B * T * N rows, without a second stream");
    dm.T = dymap->T; dm.N = dymap->N; dm.drop = make_dropn(&dymap->drop);
  }
#define LN_BWD(NCH)                                                                                                       \
  do {                                                                                                                    \
    if (has_map)                                                                                                          \
      hipLaunchKernelGGL((layernorm_bwd_kernel<NCH, false, true, true>), dim3(nb), dim3(256), 0, st, dy, lddy, x, ldx, gamma, \
                         stats, dres, dx, static_cast<float*>(dx_masked), (int)masked_bf16, odc, dgb_partials, M, D, ad, dm); \
    else if (has_add && dy_bf16)                                                                                               \
      hipLaunchKernelGGL((layernorm_bwd_kernel<NCH, true, true>), dim3(nb), dim3(256), 0, st, dy, lddy, x, ldx, gamma, stats, \
                         dres, dx, static_cast<float*>(dx_masked), (int)masked_bf16, odc, dgb_partials, M, D, ad);        \
    else if (has_add)                                                                                                     \
      hipLaunchKernelGGL((layernorm_bwd_kernel<NCH, true, false>), dim3(nb), dim3(256), 0, st, dy, lddy, x, ldx, gamma, stats, \
                         dres, dx, static_cast<float*>(dx_masked), (int)masked_bf16, odc, dgb_partials, M, D, ad);        \
    else if (dy_bf16)                                                                                                     \
      hipLaunchKernelGGL((layernorm_bwd_kernel<NCH, false, true>), dim3(nb), dim3(256), 0, st, dy, lddy, x, ldx, gamma, stats, \
                         dres, dx, static_cast<float*>(dx_masked), (int)masked_bf16, odc, dgb_partials, M, D, ad);        \
    else                                                                                                                  \
      hipLaunchKernelGGL((layernorm_bwd_kernel<NCH, false, false>), dim3(nb), dim3(256), 0, st, dy, lddy, x, ldx, gamma, stats, \
                         dres, dx, static_cast<float*>(dx_masked), (int)masked_bf16, odc, dgb_partials, M, D, ad);        \
  } while (0)
  switch (nch) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    default: LN_BWD(4); break;
  }
#undef LN_BWD
  TECM_CHECK_LAUNCH("tecm_layernorm_bwd");
  return TECM_OK;
}

extern "C" int tecm_groupnorm_gelu_fwd(const void* y_, const float* gamma, const float* beta, void* act_,
                                       float* stats, int32_t B, int32_t L, int32_t N, int32_t Cout, float eps,
                                       int32_t io_bf16, int32_t act_stride, void* stream) {
  const float* y = reinterpret_cast<const float*>(y_);
  float* act = reinterpret_cast<float*>(act_);
  TECM_REQUIRE(io_bf16 == 0 || io_bf16 == TECM_GN_OUT_BF16 || io_bf16 == (TECM_GN_OUT_BF16 | TECM_GN_Y_BF16) ||
                   io_bf16 == (TECM_GN_OUT_BF16 | TECM_GN_Y_BF16 | TECM_GN_STATS_GIVEN),
               TECM_E_ARG, "tecm_groupnorm_gelu_fwd: io_bf16 is 0, OUT_BF16, OUT_BF16 | Y_BF16 or the latter | STATS_GIVEN");
  TECM_REQUIRE(act_stride >= 1, TECM_E_ARG, "tecm_groupnorm_gelu_fwd: act_stride must be at least 1");
  if (io_bf16 & TECM_GN_STATS_GIVEN) {                   // elementwise: the statistics were computed by tecm_conv_fwd_bf16
    TECM_REQUIRE(y_ && gamma && beta && act_ && stats, TECM_E_ARG, "tecm_groupnorm_gelu_fwd: null pointer");
    TECM_REQUIRE(B > 0 && L > 0 && N > 0 && Cout > 0 && Cout % 8 == 0 && tecm_aligned(y_, 16) && tecm_aligned(act_, 16) &&
                     tecm_aligned(gamma, 16) && tecm_aligned(beta, 16) && tecm_aligned(stats, 8),
                 TECM_E_ARG, "tecm_groupnorm_gelu_fwd: the elementwise form needs Cout %% 8 == 0 and 16-byte aligned tensors");
    const int64_t octs = (int64_t)B * ((L + act_stride - 1) / act_stride) * N * (3 * Cout / 8);
    const int64_t want = (octs + 255) / 256;
    hipLaunchKernelGGL(gn_apply_fwd16_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, y_,
                       gamma, beta, act_, stats, B, L, N, Cout, (int)act_stride);
    TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_fwd/apply16");
    return TECM_OK;
  }
  const bool io16 = (io_bf16 & TECM_GN_OUT_BF16) != 0;
  if (io_bf16 & TECM_GN_Y_BF16) {                        // every tensor bf16: the oct kernels
    TECM_REQUIRE(y_ && gamma && beta && act_ && stats, TECM_E_ARG, "tecm_groupnorm_gelu_fwd: null pointer");
    TECM_REQUIRE(B > 0 && L > 0 && N > 0, TECM_E_ARG, "tecm_groupnorm_gelu_fwd: bad shape");
    const int ng = gn16_ng(L, N, Cout, y_, act_);
    TECM_REQUIRE(ng > 0, TECM_E_ARG, "tecm_groupnorm_gelu_fwd: a bf16 y needs L * 3*Cout/8 <= 2304 and Cout in {64, 128, 256} "
                 "(got L = %d, Cout = %d)", L, Cout);
    const dim3 g1((unsigned)((int64_t)B * N));
    hipStream_t st16 = (hipStream_t)stream;
#define GN_FWD16(CPB, NG) \
  hipLaunchKernelGGL((gn_gelu_fwd_reg16<CPB, NG>), g1, dim3(384), 0, st16, y_, gamma, beta, act_, stats, B, L, N, eps, (int)act_stride)
    if (ng == 3) {
      if (Cout == 64) GN_FWD16(1, 3); else if (Cout == 128) GN_FWD16(2, 3); else GN_FWD16(4, 3);
    } else {
      if (Cout == 64) GN_FWD16(1, 6); else if (Cout == 128) GN_FWD16(2, 6); else GN_FWD16(4, 6);
    }
#undef GN_FWD16
    TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_fwd/reg16");
    return TECM_OK;
  }
  TECM_REQUIRE(y && gamma && beta && act && stats, TECM_E_ARG, "tecm_groupnorm_gelu_fwd: null pointer");
  TECM_REQUIRE(B > 0 && L > 0 && N > 0, TECM_E_ARG, "tecm_groupnorm_gelu_fwd: bad shape");
  TECM_REQUIRE(Cout == 64 || Cout == 128 || Cout == 256, TECM_E_ARG,
               "tecm_groupnorm_gelu_fwd: Cout must be 64, 128 or 256 (got %d)", Cout);
  const int64_t S = (int64_t)B * N;
  const dim3 grid((unsigned)((S + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  // register-resident paths: one HBM read of y, float4 accesses
#define GN_FWD_REG(CPB, WPS, NPM, GRID) \
  do {                                                                                                            \
    if (io16)                                                                                                     \
      hipLaunchKernelGGL((gn_gelu_fwd_reg<CPB, WPS, NPM, true>), GRID, dim3(WPS > 4 ? 64 * WPS : 256), 0, st, y_, gamma, \
                         beta, act_, stats, B, L, N, eps, np, (int)act_stride);                                   \
    else                                                                                                          \
      hipLaunchKernelGGL((gn_gelu_fwd_reg<CPB, WPS, NPM, false>), GRID, dim3(WPS > 4 ? 64 * WPS : 256), 0, st, y_, gamma, \
                         beta, act_, stats, B, L, N, eps, np, (int)act_stride);                                   \
  } while (0)
  int np = gn_reg_pairs(L, N, Cout, 4, 9, y, act);
  if (np > 0) {
    const dim3 g4((unsigned)S);
    if (Cout == 64) GN_FWD_REG(1, 4, 9, g4);
    else if (Cout == 128) GN_FWD_REG(2, 4, 9, g4);
    else GN_FWD_REG(4, 4, 9, g4);
    TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_fwd/reg4");
    return TECM_OK;
  }
  np = gn_reg_pairs(L, N, Cout, 8, 9, y, act);
  if (np > 0) {
    const dim3 g8((unsigned)S);
    if (Cout == 64) GN_FWD_REG(1, 8, 9, g8);
    else if (Cout == 128) GN_FWD_REG(2, 8, 9, g8);
    else GN_FWD_REG(4, 8, 9, g8);
    TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_fwd/reg8");
    return TECM_OK;
  }
  np = gn_reg_pairs(L, N, Cout, 2, 18, y, act);
  if (np > 0) {
    const dim3 g2((unsigned)((S + 1) / 2));
    if (Cout == 64) GN_FWD_REG(1, 2, 18, g2);
    else if (Cout == 128) GN_FWD_REG(2, 2, 18, g2);
    else GN_FWD_REG(4, 2, 18, g2);
    TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_fwd/reg2");
    return TECM_OK;
  }
#undef GN_FWD_REG
  TECM_REQUIRE(!io16 && act_stride == 1, TECM_E_ARG,
               "tecm_groupnorm_gelu_fwd: bf16 tensors and a compact strided act are served by the register-resident kernels only "
               "(L * Cout too large)");
  if (Cout == 64)
    hipLaunchKernelGGL((gn_gelu_fwd_kernel<1>), grid, dim3(256), 0, st, y, gamma, beta, act, stats, B, L, N, eps);
  else if (Cout == 128)
    hipLaunchKernelGGL((gn_gelu_fwd_kernel<2>), grid, dim3(256), 0, st, y, gamma, beta, act, stats, B, L, N, eps);
  else
    hipLaunchKernelGGL((gn_gelu_fwd_kernel<4>), grid, dim3(256), 0, st, y, gamma, beta, act, stats, B, L, N, eps);
  TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_fwd");
  return TECM_OK;
}

extern "C" int tecm_groupnorm_gelu_bwd(const void* dact_, int32_t dstride, const void* y_, const float* gamma,
                                       const float* beta, const float* stats, void* dy_, float* dgb_partials,
                                       int32_t* num_blocks, int32_t B, int32_t L, int32_t N, int32_t Cout,
                                       int32_t io_bf16, float* seq_sums, void* stream) {
  const float* y = reinterpret_cast<const float*>(y_);
  float* dy = reinterpret_cast<float*>(dy_);
  TECM_REQUIRE((io_bf16 & ~(TECM_GN_OUT_BF16 | TECM_GN_DACT_BF16 | TECM_GN_Y_BF16)) == 0 &&
                   (!(io_bf16 & TECM_GN_DACT_BF16) || (io_bf16 & TECM_GN_OUT_BF16)) &&
                   (!(io_bf16 & TECM_GN_Y_BF16) || (io_bf16 & (TECM_GN_OUT_BF16 | TECM_GN_DACT_BF16)) == (TECM_GN_OUT_BF16 | TECM_GN_DACT_BF16)),
               TECM_E_ARG, "tecm_groupnorm_gelu_bwd: io_bf16 is 0, OUT_BF16, OUT_BF16 | DACT_BF16 or all three with Y_BF16");
  const bool io16 = (io_bf16 & TECM_GN_OUT_BF16) != 0, d16 = (io_bf16 & TECM_GN_DACT_BF16) != 0;
  const float* dact = reinterpret_cast<const float*>(dact_);
  TECM_REQUIRE(B > 0 && L > 0 && N > 0 && dstride > 0, TECM_E_ARG, "tecm_groupnorm_gelu_bwd: bad shape");
  TECM_REQUIRE(Cout == 64 || Cout == 128 || Cout == 256, TECM_E_ARG,
               "tecm_groupnorm_gelu_bwd: Cout must be 64, 128 or 256 (got %d)", Cout);
  const int nb = gn_blocks((int64_t)B * N);
  if (num_blocks) *num_blocks = nb;
  if (dy == nullptr) return TECM_OK;   // query mode
  TECM_REQUIRE(dact && y && gamma && beta && stats && dgb_partials, TECM_E_ARG, "tecm_groupnorm_gelu_bwd: null pointer");
  const int L2 = (L + dstride - 1) / dstride;
  hipStream_t st = (hipStream_t)stream;
  if ((io_bf16 & TECM_GN_Y_BF16) && seq_sums && (Cout == 64 || Cout == 128) && tecm_aligned(dact_, 16) && tecm_aligned(y_, 16) &&
      tecm_aligned(dy_, 16)) {
    // every tensor bf16 and a workspace for the per-sequence sums: the two streaming kernels (see gn_bwd_sums16_kernel)
    const char* sp = std::getenv("TECM_GN_BWD_SPLIT");
    if (!(sp && sp[0] == '0')) {
      // items = (sample, node group)
      const int npb = Cout == 64 ? 10 : 5;
      const int items = B * ((N + npb - 1) / npb);
      const int active = items < nb ? items : nb;       // (an even deal -- 779 blocks of 3 items instead of 1024 of 2-3 --
                                                        //  measured 145 us against 121: fewer waves in flight cost more)
      if (Cout == 64) {
        hipLaunchKernelGGL(gn_bwd_sums16_kernel<1>, dim3(nb), dim3(256), 0, st, dact_, dstride, L2, y_, gamma, beta, stats, seq_sums,
                           dgb_partials, B, L, N, active);
        hipLaunchKernelGGL(gn_bwd_apply16_kernel<1>, dim3(nb), dim3(256), 0, st, dact_, dstride, L2, y_, gamma, beta, stats, seq_sums,
                           dy_, dgb_partials, B, L, N);
      } else {
        hipLaunchKernelGGL(gn_bwd_sums16_kernel<2>, dim3(nb), dim3(256), 0, st, dact_, dstride, L2, y_, gamma, beta, stats, seq_sums,
                           dgb_partials, B, L, N, active);
        hipLaunchKernelGGL(gn_bwd_apply16_kernel<2>, dim3(nb), dim3(256), 0, st, dact_, dstride, L2, y_, gamma, beta, stats, seq_sums,
                           dy_, dgb_partials, B, L, N);
      }
      TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_bwd/split16");
      return TECM_OK;
    }
  }
  if (io_bf16 & TECM_GN_Y_BF16) {                        // every tensor bf16
    // the four-channel register kernel with 8-byte y / dact loads where its geometry serves the sequence (measured at B = 8:
    // 1.3x faster than the 8-channel kernel in this direction: more requests in flight per lane), else the 8-channel one
    const char* wps_env = std::getenv("TECM_GN_BWD_WPS");        // "8": force the 8-wave geometry (A/B diagnostics)
    const int npq = (tecm_aligned(dact_, 8) && !(wps_env && wps_env[0] == '8')) ? gn_reg_pairs(L, N, Cout, 4, 9, y_, dy_) : 0;
    if (npq > 0) {
#define GN_BWD_Q16(CPB) \
  hipLaunchKernelGGL((gn_gelu_bwd_reg<CPB, 4, 9, true, true, true>), dim3(nb), dim3(256), 0, st, dact_, dstride, L2, y_, gamma, \
                     beta, stats, dy_, dgb_partials, B, L, N, npq)
      if (Cout == 64) GN_BWD_Q16(1); else if (Cout == 128) GN_BWD_Q16(2); else GN_BWD_Q16(4);
#undef GN_BWD_Q16
      TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_bwd/reg-y16");
      return TECM_OK;
    }
    const int npq8 = tecm_aligned(dact_, 8) ? gn_reg_pairs(L, N, Cout, 8, 9, y_, dy_) : 0;
    if (npq8 > 0) {
#define GN_BWD_Q16(CPB) \
  hipLaunchKernelGGL((gn_gelu_bwd_reg<CPB, 8, 9, true, true, true>), dim3(nb), dim3(512), 0, st, dact_, dstride, L2, y_, gamma, \
                     beta, stats, dy_, dgb_partials, B, L, N, npq8)
      if (Cout == 64) GN_BWD_Q16(1); else if (Cout == 128) GN_BWD_Q16(2); else GN_BWD_Q16(4);
#undef GN_BWD_Q16
      TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_bwd/reg8-y16");
      return TECM_OK;
    }
    const int ng = gn16_ng(L, N, Cout, y_, dy_);
    TECM_REQUIRE(ng > 0 && tecm_aligned(dact_, 16), TECM_E_ARG,
                 "tecm_groupnorm_gelu_bwd: a bf16 y needs L * 3*Cout/8 <= 2304, Cout in {64, 128, 256}, 16-byte aligned tensors");
#define GN_BWD16(CPB, NG) \
  hipLaunchKernelGGL((gn_gelu_bwd_reg16<CPB, NG>), dim3(nb), dim3(384), 0, st, dact_, dstride, L2, y_, gamma, beta, stats, dy_, \
                     dgb_partials, B, L, N)
    if (ng == 3) {
      if (Cout == 64) GN_BWD16(1, 3); else if (Cout == 128) GN_BWD16(2, 3); else GN_BWD16(4, 3);
    } else {
      if (Cout == 64) GN_BWD16(1, 6); else if (Cout == 128) GN_BWD16(2, 6); else GN_BWD16(4, 6);
    }
#undef GN_BWD16
    TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_bwd/reg16");
    return TECM_OK;
  }
#define GN_BWD_REG(CPB, WPS, NT, NPV)                                                                              \
  do {                                                                                                            \
    if (io16 && d16)                                                                                              \
      hipLaunchKernelGGL((gn_gelu_bwd_reg<CPB, WPS, 9, true, true>), dim3(nb), dim3(NT), 0, st, dact_, dstride, L2, y_, gamma, \
                         beta, stats, dy_, dgb_partials, B, L, N, NPV);                                           \
    else if (io16)                                                                                                \
      hipLaunchKernelGGL((gn_gelu_bwd_reg<CPB, WPS, 9, true>), dim3(nb), dim3(NT), 0, st, dact_, dstride, L2, y_, gamma, beta, \
                         stats, dy_, dgb_partials, B, L, N, NPV);                                                 \
    else                                                                                                          \
      hipLaunchKernelGGL((gn_gelu_bwd_reg<CPB, WPS, 9, false>), dim3(nb), dim3(NT), 0, st, dact_, dstride, L2, y_, gamma, beta, \
                         stats, dy_, dgb_partials, B, L, N, NPV);                                                 \
  } while (0)
  const int np = tecm_aligned(dact, 16) ? gn_reg_pairs(L, N, Cout, 4, 9, y, dy) : 0;
  if (np > 0) {
    if (Cout == 64)
      GN_BWD_REG(1, 4, 256, np);
    else if (Cout == 128)
      GN_BWD_REG(2, 4, 256, np);
    else
      GN_BWD_REG(4, 4, 256, np);
    TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_bwd/reg");
    return TECM_OK;
  }
  const int np8 = tecm_aligned(dact, 16) ? gn_reg_pairs(L, N, Cout, 8, 9, y, dy) : 0;
  if (np8 > 0) {
    if (Cout == 64)
      GN_BWD_REG(1, 8, 512, np8);
    else if (Cout == 128)
      GN_BWD_REG(2, 8, 512, np8);
    else
      GN_BWD_REG(4, 8, 512, np8);
    TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_bwd/reg8");
    return TECM_OK;
  }
#undef GN_BWD_REG
  TECM_REQUIRE(!io16 && !d16, TECM_E_ARG,
               "tecm_groupnorm_gelu_bwd: bf16 tensors are served by the register-resident kernels only (L * Cout too large)");
  if (Cout == 64)
    hipLaunchKernelGGL((gn_gelu_bwd_kernel<1>), dim3(nb), dim3(256), 0, st, dact, dstride, L2, y, gamma, beta, stats,
                       dy, dgb_partials, B, L, N);
  else if (Cout == 128)
    hipLaunchKernelGGL((gn_gelu_bwd_kernel<2>), dim3(nb), dim3(256), 0, st, dact, dstride, L2, y, gamma, beta, stats,
                       dy, dgb_partials, B, L, N);
  else
    hipLaunchKernelGGL((gn_gelu_bwd_kernel<4>), dim3(nb), dim3(256), 0, st, dact, dstride, L2, y, gamma, beta, stats,
                       dy, dgb_partials, B, L, N);
  TECM_CHECK_LAUNCH("tecm_groupnorm_gelu_bwd");
  return TECM_OK;
}

static int colsum_impl(const float* in, int64_t ld, int64_t outer, int64_t inner, int32_t nseg, int32_t C, float* out,
                       int64_t ldo, int32_t accumulate, float scale, const TecmDrop* in_drop, float* workspace, void* twin,
                       int64_t ld_twin, void* stream) {
  TECM_REQUIRE(in && out && workspace, TECM_E_ARG, "tecm_colsum: null pointer");
  TECM_REQUIRE(outer > 0 && inner > 0 && nseg > 0 && C > 0, TECM_E_ARG, "tecm_colsum: bad shape");
  const int colblocks = (C + 63) / 64;
  const int64_t total = outer * inner;
  int64_t rb = 1024 / ((int64_t)colblocks * nseg);
  if (rb > 256) rb = 256;
  if (rb > (total + 15) / 16) rb = (total + 15) / 16;
  if (rb < 1) rb = 1;
  const int64_t chunk = (total + rb - 1) / rb;
  rb = (total + chunk - 1) / chunk;
  hipStream_t st = (hipStream_t)stream;
  const DropCtxN idc = make_dropn(in_drop);
  if (C % 4 == 0 && C <= 1024 && ld % 4 == 0 && tecm_aligned(in, 16) && tecm_aligned(workspace, 16)) {
    int64_t rb4 = 1024 / nseg;                          // workspace contract: >= 1024 * nseg * C floats
    // at least 16 rows per block.  (Fewer, larger blocks to shrink the second stage were measured at B = 2 -- where the two
    // stages of the 17 column sums are 0.35 ms of a 5.2 ms step: 512 rows per block 5.20 -> 5.85 ms, 2048 -> 7.9 ms, 16 instead
    // of 64 -0.03 ms: the first stage lives on its block count.)  TECM_COLSUM_MINROWS overrides (diagnostics).
    static const int64_t min_rows = [] { const char* e = std::getenv("TECM_COLSUM_MINROWS"); return e ? std::atoll(e) : 16ll; }();
    if (rb4 > (total + min_rows - 1) / min_rows) rb4 = (total + min_rows - 1) / min_rows;
    if (rb4 < 1) rb4 = 1;
    const int64_t chunk4 = (total + rb4 - 1) / rb4;
    rb4 = (total + chunk4 - 1) / chunk4;
    hipLaunchKernelGGL(colsum_stage1_v4, dim3((unsigned)rb4, nseg), dim3(256), 0, st, in, ld, outer, inner, nseg, C,
                       idc, workspace, chunk4, static_cast<__bf16*>(twin), ld_twin);
    TECM_CHECK_LAUNCH("tecm_colsum/stage1_v4");
    launch_colsum_stage2(workspace, (int)rb4, nseg, C, out, ldo, accumulate, scale, colblocks, st);
    TECM_CHECK_LAUNCH("tecm_colsum/stage2");
    return TECM_OK;
  }
  TECM_REQUIRE(!twin, TECM_E_ALIGN, "tecm_colsum_twin: needs C %% 4 == 0, C <= 1024 and 16-byte friendly rows");
  hipLaunchKernelGGL(colsum_stage1, dim3(colblocks, (unsigned)rb, nseg), dim3(256), 0, st, in, ld, outer, inner, nseg,
                     C, idc, workspace, chunk);
  TECM_CHECK_LAUNCH("tecm_colsum/stage1");
  launch_colsum_stage2(workspace, (int)rb, nseg, C, out, ldo, accumulate, scale, colblocks, st);
  TECM_CHECK_LAUNCH("tecm_colsum/stage2");
  return TECM_OK;
}

extern "C" int tecm_colsum(const float* in, int64_t ld, int64_t outer, int64_t inner, int32_t nseg, int32_t C,
                           float* out, int64_t ldo, int32_t accumulate, float scale, const TecmDrop* in_drop,
                           float* workspace, void* stream) {
  return colsum_impl(in, ld, outer, inner, nseg, C, out, ldo, accumulate, scale, in_drop, workspace, nullptr, 0, stream);
}

// The same pass also writes the (masked) values as a bf16 matrix twin[row][ld_twin]: the operand two bf16 contractions
// read next (their loaders would round the fp32 values to exactly these), at no extra read of `in`.
extern "C" int tecm_colsum_twin(const float* in, int64_t ld, int64_t outer, int64_t inner, int32_t nseg, int32_t C,
                                float* out, int64_t ldo, int32_t accumulate, float scale, const TecmDrop* in_drop,
                                float* workspace, void* twin_bf16, int64_t ld_twin, void* stream) {
  TECM_REQUIRE(twin_bf16 && ld_twin >= C && ld_twin % 4 == 0 && tecm_aligned(twin_bf16, 8), TECM_E_ARG,
               "tecm_colsum_twin: the bf16 twin needs 8-byte friendly rows of at least C values");
  return colsum_impl(in, ld, outer, inner, nseg, C, out, ldo, accumulate, scale, in_drop, workspace, twin_bf16, ld_twin, stream);
}
