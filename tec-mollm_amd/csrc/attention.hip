// Causal multi-head attention over the T (<= 32) latent-patch tokens of one (b, n) sequence --
// GPT2Attention with an all-ones mask (modeling_gpt2.py:54-73, :144-226; tec_mollm.py:111).
//
// T is tiny (3 for L_in=48, 6 for 96, 21 for 336) while there are B*N*12 independent (sequence, head)
// problems, so this is a bandwidth kernel, not an MFMA one: 16 lanes own one (sequence, head), each
// lane holds 4 of the 64 head dims (one float4 = a 256-byte coalesced segment per 16 lanes), dot
// products are 16-lane shuffle reductions, softmax is per-lane scalar work replicated in the group.
// Rows are time-major: token p of sequence (b, n) is row (b*T + p)*N + n of the (B,T,N,3D) qkv buffer.
// Dropout on the probabilities uses idx = (((b*N + n)*H + h)*T + i)*T + j.
#include "common.h"

namespace {

struct DropA {
  uint64_t seed;
  uint32_t thresh;
  float inv;
  const uint64_t* sdev;        // TecmDrop::seed_dev: added to seed when the kernel starts
};

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
}
__device__ __forceinline__ void fma4(float4& acc, float s, const float4& v) {
  acc.x += s * v.x; acc.y += s * v.y; acc.z += s * v.z; acc.w += s * v.w;
}

// qkv is fp32, or bf16 (template Q16) when the c_attn GEMM stored it that way (bf16 mode: the Linear's output is a bf16
// tensor under autocast): four consecutive elements at element offset `off` as floats
template <bool Q16>
__device__ __forceinline__ float4 ldq(const float* __restrict__ qkv, int64_t off) {
  if constexpr (Q16) {
    const tecm_bf16x4 h = *reinterpret_cast<const tecm_bf16x4*>(reinterpret_cast<const __bf16*>(qkv) + off);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
  } else {
    return *reinterpret_cast<const float4*>(qkv + off);
  }
}

// ctx is fp32, or bf16 when its only reader is a bf16 matrix-core GEMM (rounded once here instead of in that loader)
__device__ __forceinline__ void store_ctx(float* ctx, int ctx_bf16, int64_t off, const float4& o) {
  if (ctx_bf16)
    tecm_store_bf16x4(reinterpret_cast<__bf16*>(ctx) + off, o.x, o.y, o.z, o.w);
  else
    *reinterpret_cast<float4*>(ctx + off) = o;
}

// TT > 0: compile-time T with q/k/v held in registers; TT == 0: runtime T <= 32, k/v re-read (L1/L2).
template <int TT, bool Q16>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                            int ctx_bf16, int B, int Trt, int N, int H, int D, DropA dr) {
  dr.seed = tecm_seed_now(dr.seed, dr.sdev);
  const int T = TT > 0 ? TT : Trt;
  const int sub = threadIdx.x & 15;
  const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t items = (int64_t)B * N * H;
  if (item >= items) return;                      // whole 16-lane groups leave together
  const int64_t seq = item / H;
  const int h = (int)(item - seq * H);
  const int b = (int)(seq / N), n = (int)(seq - (int64_t)b * N);
  const int64_t ld = 3 * (int64_t)D;
  const float scale = 0.125f;                     // 1/sqrt(64)
  const int64_t row0 = ((int64_t)b * T) * N + n;  // row of token 0; token p at row0 + p*N
  const int col = h * 64 + sub * 4;

  if constexpr (TT > 0) {
    float4 q[TT], k[TT], v[TT];
#pragma unroll
    for (int p = 0; p < TT; ++p) {
      const int64_t r = (row0 + (int64_t)p * N) * ld + col;
      q[p] = ldq<Q16>(qkv, r);
      k[p] = ldq<Q16>(qkv, r + D);
      v[p] = ldq<Q16>(qkv, r + 2 * D);
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      float s[TT];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        s[j] = group16_sum(dot4(q[i], k[j])) * scale;
        mx = fmaxf(mx, s[j]);
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        s[j] = expf(s[j] - mx);
        den += s[j];
      }
      const float inv = 1.0f / den;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        float p = s[j] * inv;
        if (dr.thresh) p *= tecm_drop_mult(dr.seed, (uint64_t)(((item * T + i) * T) + j), dr.thresh, dr.inv);
        fma4(o, p, v[j]);
      }
      store_ctx(ctx, ctx_bf16, (row0 + (int64_t)i * N) * D + col, o);
    }
  } else {
    for (int i = 0; i < T; ++i) {
      const float4 qi = ldq<Q16>(qkv, (row0 + (int64_t)i * N) * ld + col);
      float s[32];
      float mx = -INFINITY;
      for (int j = 0; j <= i; ++j) {
        const float4 kj = ldq<Q16>(qkv, (row0 + (int64_t)j * N) * ld + D + col);
        s[j] = group16_sum(dot4(qi, kj)) * scale;
        mx = fmaxf(mx, s[j]);
      }
      float den = 0.f;
      for (int j = 0; j <= i; ++j) {
        s[j] = expf(s[j] - mx);
        den += s[j];
      }
      const float inv = 1.0f / den;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j <= i; ++j) {
        const float4 vj = ldq<Q16>(qkv, (row0 + (int64_t)j * N) * ld + 2 * D + col);
        float p = s[j] * inv;
        if (dr.thresh) p *= tecm_drop_mult(dr.seed, (uint64_t)(((item * T + i) * T) + j), dr.thresh, dr.inv);
        fma4(o, p, vj);
      }
      store_ctx(ctx, ctx_bf16, (row0 + (int64_t)i * N) * D + col, o);
    }
  }
}

// Forward for 12 < T <= TM: k and v register-resident (2 x 24 float4), q_i loaded per query row; loops unrolled to TM
// behind wave-uniform `i < T` guards (see attention_bwd_kernel_qstream).  The runtime-T path above re-reads k_j / v_j
// from L1/L2 for every (i, j).
template <int TM, bool Q16>
__global__ __launch_bounds__(256) void attention_fwd_kernel_kv(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                               int ctx_bf16, int B, int T, int N, int H, int D, DropA dr) {
  dr.seed = tecm_seed_now(dr.seed, dr.sdev);
  const int sub = threadIdx.x & 15;
  const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t items = (int64_t)B * N * H;
  if (item >= items) return;
  const int64_t seq = item / H;
  const int h = (int)(item - seq * H);
  const int b = (int)(seq / N), n = (int)(seq - (int64_t)b * N);
  const int64_t ld = 3 * (int64_t)D;
  const float scale = 0.125f;
  const int64_t row0 = ((int64_t)b * T) * N + n;
  const int col = h * 64 + sub * 4;
  float4 k[TM], v[TM];
#pragma unroll
  for (int p = 0; p < TM; ++p) {
    k[p] = v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < T) {
      const int64_t r = (row0 + (int64_t)p * N) * ld + col;
      k[p] = ldq<Q16>(qkv, r + D);
      v[p] = ldq<Q16>(qkv, r + 2 * D);
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i < T) {
      const float4 qi = ldq<Q16>(qkv, (row0 + (int64_t)i * N) * ld + col);
      float s[TM];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        s[j] = group16_sum(dot4(qi, k[j])) * scale;
        mx = fmaxf(mx, s[j]);
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        s[j] = expf(s[j] - mx);
        den += s[j];
      }
      const float inv = 1.0f / den;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        float p = s[j] * inv;
        if (dr.thresh) p *= tecm_drop_mult(dr.seed, (uint64_t)(((item * T + i) * T) + j), dr.thresh, dr.inv);
        fma4(o, p, v[j]);
      }
      store_ctx(ctx, ctx_bf16, (row0 + (int64_t)i * N) * D + col, o);
    }
  }
}

// Backward: recompute the probabilities, then
//   dP~_ij = <dctx_i, v_j>;  dV_j += P~_ij dctx_i;  dP_ij = dP~_ij * keep/(1-p);
//   dS_ij = P_ij (dP_ij - sum_k P_ik dP_ik);  dQ_i += dS_ij K_j / 8;  dK_j += dS_ij Q_i / 8.
// D16: dctx is the bf16 tensor the attn.c_proj d-input GEMM wrote (bf16 mode: the gradient of a bf16 Linear's input)
template <int TT, bool Q16, bool D16 = false>
__device__ __forceinline__ void attention_bwd_body(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                   float* __restrict__ dqkv, int dqkv_bf16, int B, int Trt, int N, int H,
                                                   int D, DropA dr) {
  dr.seed = tecm_seed_now(dr.seed, dr.sdev);
  const int T = TT > 0 ? TT : Trt;
  const int sub = threadIdx.x & 15;
  const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t items = (int64_t)B * N * H;
  if (item >= items) return;
  const int64_t seq = item / H;
  const int h = (int)(item - seq * H);
  const int b = (int)(seq / N), n = (int)(seq - (int64_t)b * N);
  const int64_t ld = 3 * (int64_t)D;
  const float scale = 0.125f;
  const int64_t row0 = ((int64_t)b * T) * N + n;
  const int col = h * 64 + sub * 4;
  constexpr int TM = TT > 0 ? TT : 32;

  float4 q[TM], k[TM], v[TM], dq[TM], dk[TM], dv[TM];
#pragma unroll
  for (int p = 0; p < T; ++p) {
    const int64_t r = (row0 + (int64_t)p * N) * ld + col;
    q[p] = ldq<Q16>(qkv, r);
    k[p] = ldq<Q16>(qkv, r + D);
    v[p] = ldq<Q16>(qkv, r + 2 * D);
    dq[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    dk[p] = dq[p];
    dv[p] = dq[p];
  }
#pragma unroll
  for (int i = 0; i < T; ++i) {
    const float4 go = ldq<D16>(dctx, (row0 + (int64_t)i * N) * D + col);
    float pr[TM], dp[TM];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      pr[j] = group16_sum(dot4(q[i], k[j])) * scale;
      mx = fmaxf(mx, pr[j]);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      pr[j] = expf(pr[j] - mx);
      den += pr[j];
    }
    const float inv = 1.0f / den;
    float dotp = 0.f;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      pr[j] *= inv;
      float m = 1.0f;
      if (dr.thresh) m = tecm_drop_mult(dr.seed, (uint64_t)(((item * T + i) * T) + j), dr.thresh, dr.inv);
      const float dpt = group16_sum(dot4(go, v[j]));
      fma4(dv[j], pr[j] * m, go);
      dp[j] = dpt * m;
      dotp += pr[j] * dp[j];
    }
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const float ds = pr[j] * (dp[j] - dotp) * scale;
      fma4(dq[i], ds, k[j]);
      fma4(dk[j], ds, q[i]);
    }
  }
#pragma unroll
  for (int p = 0; p < T; ++p) {
    const int64_t off = (row0 + (int64_t)p * N) * ld + col;       // dqkv: fp32, or bf16 for the two bf16 GEMMs behind it
    store_ctx(dqkv, dqkv_bf16, off, dq[p]);
    store_ctx(dqkv, dqkv_bf16, off + D, dk[p]);
    store_ctx(dqkv, dqkv_bf16, off + 2 * D, dv[p]);
  }
}

// 12 < T <= 24 (the reference's default L_in = 336 gives T = 21): k, v, dk, dv stay in registers (4 x 24 float4 = 384 of
// the 512-entry file, one wave per SIMD), q_i is loaded and dq_i stored per query row.  Loops are unrolled to TM = 24
// behind wave-uniform `i < T` guards: registers stay statically indexed (the runtime-T body above indexes float4[32]
// arrays with runtime bounds, i.e. scratch: 6.0 ms per launch at B = 2, L_in = 336 against 1.2 ms for the forward), and
// the guards keep the scheduler from interleaving query rows (a guard-free T = 21 instance hoisted the loads of all
// rows and spilled 856 VGPRs).
// The mask of this kernel is drawn through an out-of-line call.  With the round-5 hash inlined (16 VALU instructions instead of
// 70) hipcc 7.2 allocates this body differently -- 114 spilled VGPRs instead of 81 for the fp32 instance -- and the kernel
// then returns NON-REPRODUCIBLE garbage from T = 16 on, dropout on or off (tools/scratch/att_debug.py: three launches,
// three answers; T = 13 fine), while the identical source with the hash behind a call is exact and bit-reproducible at every
// T.  The body sits at the edge of the 512-register file by design (see above); which side of that edge the allocator
// lands on must not depend on an inlining decision, so the call is spelled out.  test_attention_fwd_bwd[13..32] is the guard.
__device__ __attribute__((noinline)) float att_drop_mult_call(uint64_t seed, uint64_t idx, uint32_t thresh, float inv_keep) {
  return tecm_drop_mult(seed, idx, thresh, inv_keep);
}

template <int TM, bool Q16, bool D16 = false>
__global__ __launch_bounds__(256, 1) void attention_bwd_kernel_qstream(const float* __restrict__ qkv,
                                                                       const float* __restrict__ dctx,
                                                                       float* __restrict__ dqkv, int dqkv_bf16, int B, int T,
                                                                       int N, int H, int D, DropA dr) {
  dr.seed = tecm_seed_now(dr.seed, dr.sdev);
  const int sub = threadIdx.x & 15;
  const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t items = (int64_t)B * N * H;
  if (item >= items) return;
  const int64_t seq = item / H;
  const int h = (int)(item - seq * H);
  const int b = (int)(seq / N), n = (int)(seq - (int64_t)b * N);
  const int64_t ld = 3 * (int64_t)D;
  const float scale = 0.125f;
  const int64_t row0 = ((int64_t)b * T) * N + n;
  const int col = h * 64 + sub * 4;

  float4 k[TM], v[TM], dk[TM], dv[TM];
#pragma unroll
  for (int p = 0; p < TM; ++p) {
    k[p] = v[p] = dk[p] = dv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < T) {
      const int64_t r = (row0 + (int64_t)p * N) * ld + col;
      k[p] = ldq<Q16>(qkv, r + D);
      v[p] = ldq<Q16>(qkv, r + 2 * D);
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i < T) {
      const int64_t roff = row0 + (int64_t)i * N;
      const float4 qi = ldq<Q16>(qkv, roff * ld + col);
      const float4 go = ldq<D16>(dctx, roff * D + col);
      float4 dqi = make_float4(0.f, 0.f, 0.f, 0.f);
      float pr[TM], dp[TM];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        pr[j] = group16_sum(dot4(qi, k[j])) * scale;
        mx = fmaxf(mx, pr[j]);
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        pr[j] = expf(pr[j] - mx);
        den += pr[j];
      }
      const float inv = 1.0f / den;
      float dotp = 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        pr[j] *= inv;
        float m = 1.0f;
        if (dr.thresh) m = att_drop_mult_call(dr.seed, (uint64_t)(((item * T + i) * T) + j), dr.thresh, dr.inv);
        const float dpt = group16_sum(dot4(go, v[j]));
        fma4(dv[j], pr[j] * m, go);
        dp[j] = dpt * m;
        dotp += pr[j] * dp[j];
      }
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const float ds = pr[j] * (dp[j] - dotp) * scale;
        fma4(dqi, ds, k[j]);
        fma4(dk[j], ds, qi);
      }
      store_ctx(dqkv, dqkv_bf16, roff * ld + col, dqi);
    }
  }
#pragma unroll
  for (int p = 0; p < TM; ++p) {
    if (p < T) {
      const int64_t off = (row0 + (int64_t)p * N) * ld + col;
      store_ctx(dqkv, dqkv_bf16, off + D, dk[p]);
      store_ctx(dqkv, dqkv_bf16, off + 2 * D, dv[p]);
    }
  }
}

template <int TT, bool Q16, bool D16 = false>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ dctx, float* __restrict__ dqkv,
                                                            int dqkv_bf16, int B, int Trt, int N, int H, int D, DropA dr) {
  attention_bwd_body<TT, Q16, D16>(qkv, dctx, dqkv, dqkv_bf16, B, Trt, N, H, D, dr);
}
// T = 8, 12: six float4[T] register arrays need more than 256 VGPRs -- one wave per SIMD, the whole 512-entry file
template <int TT, bool Q16, bool D16 = false>
__global__ __launch_bounds__(256, 1) void attention_bwd_kernel_wide(const float* __restrict__ qkv,
                                                                    const float* __restrict__ dctx,
                                                                    float* __restrict__ dqkv, int dqkv_bf16, int B, int Trt,
                                                                    int N, int H, int D, DropA dr) {
  attention_bwd_body<TT, Q16, D16>(qkv, dctx, dqkv, dqkv_bf16, B, Trt, N, H, D, dr);
}

int check(const char* who, const void* a, const void* b, const void* c, int B, int T, int N, int heads, int D) {
  TECM_REQUIRE(a && b && c, TECM_E_ARG, "%s: null pointer", who);
  TECM_REQUIRE(B > 0 && N > 0 && T > 0 && T <= 32, TECM_E_ARG, "%s: need 1 <= T <= 32 (got %d)", who, T);
  TECM_REQUIRE(heads > 0 && D == heads * 64, TECM_E_ARG, "%s: head_dim must be 64 (D=%d heads=%d)", who, D, heads);
  TECM_REQUIRE(tecm_aligned(a, 16) && tecm_aligned(b, 16) && tecm_aligned(c, 16), TECM_E_ALIGN,
               "%s: 16-byte alignment required", who);
  return TECM_OK;
}

DropA make_dropa(const TecmDrop* d) {
  DropA r;
  r.seed = d ? d->seed : 0;
  r.sdev = d ? d->seed_dev : nullptr;
  r.thresh = (d && d->p > 0.f) ? tecm_drop_thresh(d->p) : 0u;
  r.inv = (d && d->p > 0.f) ? 1.0f / (1.0f - d->p) : 1.0f;
  return r;
}

}  // namespace

extern "C" int tecm_attention_fwd(const float* qkv, void* ctxv, int32_t io_bf16, int32_t B, int32_t T, int32_t N,
                                  int32_t heads, int32_t D, const TecmDrop* prob_drop, void* stream) {
  float* ctx = static_cast<float*>(ctxv);
  const int rc = check("tecm_attention_fwd", qkv, ctx, ctx, B, T, N, heads, D);
  if (rc) return rc;
  TECM_REQUIRE((io_bf16 & ~3) == 0, TECM_E_ARG, "tecm_attention_fwd: io_bf16 is a mask of TECM_ATT_OUT_BF16 | TECM_ATT_QKV_BF16");
  const int ctx_bf16 = io_bf16 & TECM_ATT_OUT_BF16;
  const bool q16 = (io_bf16 & TECM_ATT_QKV_BF16) != 0;
  const int64_t items = (int64_t)B * N * heads;
  const dim3 grid((unsigned)((items + 15) / 16));
  const DropA dr = make_dropa(prob_drop);
  hipStream_t st = (hipStream_t)stream;
#define ATT_FWD(TT)                                                                                                        \
  do {                                                                                                                     \
    if (q16) hipLaunchKernelGGL((attention_fwd_kernel<TT, true>), grid, dim3(256), 0, st, qkv, ctx, ctx_bf16, B, T, N, heads, D, dr); \
    else hipLaunchKernelGGL((attention_fwd_kernel<TT, false>), grid, dim3(256), 0, st, qkv, ctx, ctx_bf16, B, T, N, heads, D, dr);    \
  } while (0)
  switch (T) {
    case 1: ATT_FWD(1); break;
    case 2: ATT_FWD(2); break;
    case 3: ATT_FWD(3); break;
    case 4: ATT_FWD(4); break;
    case 6: ATT_FWD(6); break;
    case 8: ATT_FWD(8); break;
    case 12: ATT_FWD(12); break;       // L_in = 192 with patch_len 4
    default:
      if (T > 12 && T <= 24) {           // L_in = 336 -> 21 tokens (the reference's default)
        if (q16) hipLaunchKernelGGL((attention_fwd_kernel_kv<24, true>), grid, dim3(256), 0, st, qkv, ctx, ctx_bf16, B, T, N, heads, D, dr);
        else hipLaunchKernelGGL((attention_fwd_kernel_kv<24, false>), grid, dim3(256), 0, st, qkv, ctx, ctx_bf16, B, T, N, heads, D, dr);
      } else {
        ATT_FWD(0);                      // runtime T <= 32: k / v re-read from L1/L2
      }
      break;
  }
#undef ATT_FWD
  TECM_CHECK_LAUNCH("tecm_attention_fwd");
  return TECM_OK;
}

extern "C" int tecm_attention_bwd(const float* qkv, const float* dctx, void* dqkv_, int32_t io_bf16, int32_t B, int32_t T,
                                  int32_t N, int32_t heads, int32_t D, const TecmDrop* prob_drop, void* stream) {
  float* dqkv = reinterpret_cast<float*>(dqkv_);
  const int rc = check("tecm_attention_bwd", qkv, dctx, dqkv, B, T, N, heads, D);
  if (rc) return rc;
  TECM_REQUIRE((io_bf16 & ~7) == 0 && (!(io_bf16 & TECM_ATT_DCTX_BF16) || (io_bf16 & TECM_ATT_QKV_BF16)), TECM_E_ARG,
               "tecm_attention_bwd: io_bf16 is a mask of TECM_ATT_OUT_BF16 | TECM_ATT_QKV_BF16 | TECM_ATT_DCTX_BF16 (the last "
               "only with a bf16 qkv)");
  const int dqkv_bf16 = io_bf16 & TECM_ATT_OUT_BF16;
  const bool q16 = (io_bf16 & TECM_ATT_QKV_BF16) != 0, d16 = (io_bf16 & TECM_ATT_DCTX_BF16) != 0;
  const int64_t items = (int64_t)B * N * heads;
  const dim3 grid((unsigned)((items + 15) / 16));
  const DropA dr = make_dropa(prob_drop);
  hipStream_t st = (hipStream_t)stream;
#define ATT_LAUNCH(KERNEL, TT)                                                                                             \
  do {                                                                                                                     \
    if (q16 && d16) hipLaunchKernelGGL((KERNEL<TT, true, true>), grid, dim3(256), 0, st, qkv, dctx, dqkv, dqkv_bf16, B, T, N, heads, D, dr); \
    else if (q16) hipLaunchKernelGGL((KERNEL<TT, true>), grid, dim3(256), 0, st, qkv, dctx, dqkv, dqkv_bf16, B, T, N, heads, D, dr); \
    else hipLaunchKernelGGL((KERNEL<TT, false>), grid, dim3(256), 0, st, qkv, dctx, dqkv, dqkv_bf16, B, T, N, heads, D, dr);    \
  } while (0)
  switch (T) {
    case 1: ATT_LAUNCH(attention_bwd_kernel, 1); break;
    case 2: ATT_LAUNCH(attention_bwd_kernel, 2); break;
    case 3: ATT_LAUNCH(attention_bwd_kernel, 3); break;
    case 4: ATT_LAUNCH(attention_bwd_kernel, 4); break;
    case 6: ATT_LAUNCH(attention_bwd_kernel, 6); break;
    case 8: ATT_LAUNCH(attention_bwd_kernel_wide, 8); break;
    case 12: ATT_LAUNCH(attention_bwd_kernel_wide, 12); break;
    default:
      if (T > 12 && T <= 24) ATT_LAUNCH(attention_bwd_kernel_qstream, 24);
      else ATT_LAUNCH(attention_bwd_kernel, 0);
      break;
  }
#undef ATT_LAUNCH
  TECM_CHECK_LAUNCH("tecm_attention_bwd");
  return TECM_OK;
}
