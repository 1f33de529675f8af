// Shared device/host helpers for libtecmollm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>

#include "../../include/tecmollm.h"

#define TECM_WAVE 64

// ------------------------------------------------------------------ host-side error plumbing
void tecm_set_error(const char* fmt, ...);

#define TECM_REQUIRE(cond, code, ...)            \
  do {                                           \
    if (!(cond)) {                               \
      tecm_set_error(__VA_ARGS__);               \
      return (code);                             \
    }                                            \
  } while (0)

#define TECM_CHECK_LAUNCH(name)                                                     \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      tecm_set_error("%s: launch failed: %s", (name), hipGetErrorString(e_));       \
      return TECM_E_LAUNCH;                                                         \
    }                                                                               \
  } while (0)

static inline bool tecm_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ------------------------------------------------------------------ counter-based dropout mask
// keep(seed, idx): a 32-bit avalanche mixer ("lowbias32": xorshift 16 / * 0x7feb352d / xorshift 15 / * 0x846ca68b /
// xorshift 16) of  lo(idx) + lo(seed) + hi(seed) * 0x9E3779B1 + (hi(idx) & 0xffffff) * 0x9E3779  (mod 2^32: the high
// words move the stream by odd multiples, the seed's part is loop-invariant, the index's is a full-rate 24-bit multiply
// that is zero below 2^32 elements); top 24 bits compared with p*2^24.  Pure function => forward and backward agree
// without storing masks.  Mirrored bit-for-bit by tecmollm/rng.py for the parity tests with dropout enabled.
// (Round 5: this replaces the splitmix64 finaliser of rounds 1-4 -- three 64-bit multiplications = twelve quarter-rate
//  32-bit multiplies per element, about 70 VALU issue slots against 16 now; in the GEMM epilogues that draw a mask per
//  output element the hash cost more than the GELU.  The seeds stay splitmix64-derived per site, tecmollm/ops.py.)
__host__ __device__ __forceinline__ uint32_t tecm_hash24(uint64_t seed, uint64_t idx) {
  uint32_t x = (uint32_t)idx + ((uint32_t)(idx >> 32) & 0xffffffu) * 0x9E3779u;
  x += (uint32_t)seed + (uint32_t)(seed >> 32) * 0x9E3779B1u;
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x >> 8;
}
// the seed a kernel draws with: the recorded one plus the step's device word (TecmDrop::seed_dev, tecmollm.h)
__device__ __forceinline__ uint64_t tecm_seed_now(uint64_t seed, const uint64_t* seed_dev) {
  return seed_dev ? seed + *seed_dev : seed;
}
__host__ __device__ __forceinline__ uint32_t tecm_drop_thresh(float p) { return (uint32_t)(p * 16777216.0f); }
// returns the multiplier to apply: 0 or 1/(1-p)
__device__ __forceinline__ float tecm_drop_mult(uint64_t seed, uint64_t idx, uint32_t thresh, float inv_keep) {
  return tecm_hash24(seed, idx) >= thresh ? inv_keep : 0.0f;
}

// ------------------------------------------------------------------ bf16 stores (RNE; a plain cast is NaN-safe)
typedef __bf16 tecm_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tecm_store_bf16x4(void* dst, float a, float b, float c, float d) {
  tecm_bf16x4 v;
  v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
  *reinterpret_cast<tecm_bf16x4*>(dst) = v;                  // 8 bytes
}

// ------------------------------------------------------------------ activations
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// erf-GELU and its derivative from ONE exponential: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far
// inside the 1e-3 parity bar) whose e^{-z^2}, z = x/sqrt(2), is also the Gaussian pdf of the derivative.  A dozen
// instructions instead of erff's two-branch polynomial: for kernels that unroll the activation many times.
// (v_rcp_f32 / v_exp_f32 directly: `__frcp_rn` and `__fdividef` compile to the 11-instruction IEEE division sequence on this
// toolchain, which made the activation half of the GroupNorm kernels' instruction count and cost the c_fc GEMM 160 us of
// epilogue per launch; the hardware reciprocal is good to 1 ulp)
__device__ __forceinline__ void gelu_erf_parts(float x, float& cdf, float& e) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);            // e^{-x^2/2}
  const float poly =
      t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
  float cdf, e;
  gelu_erf_parts(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float dgelu_erf_fast(float x) {
  float cdf, e;
  gelu_erf_parts(x, cdf, e);
  return cdf + x * 0.39894228040143267794f * e;
}
// tanh(u) = 1 - 2/(1 + e^{2u}) on the hardware exp/rcp units (abs error ~1e-7 .. 1e-6, saturates cleanly)
__device__ __forceinline__ float fast_tanh(float u) {
  const float e = __builtin_amdgcn_exp2f(2.88539008177792681472f * u);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
// tanh-GELU (transformers "gelu_new", the GPT-2 MLP) in its sigmoid form: with u = sqrt(2/pi) (x + 0.044715 x^3),
//   0.5 x (1 + tanh u) = x s,   s = E / (1 + E),  E = e^{2u};      1 - s = 1 / (1 + E) = r
//   d/dx = s + x s r (2 du/dx),                                     2 du/dx = 2 sqrt(2/pi) (1 + 3 * 0.044715 x^2)
// E = 2^a is clamped at 2^100 so that s = E * r is 1 (not inf * 0) for large x and exactly x * 0 for very negative x;
// s = E * r has no cancellation on either side.  7 / 12 instructions (two of them transcendental) against 23 / 26.
__device__ __forceinline__ void gelu_tanh_parts(float x, float x2, float& s, float& r) {
  float a = x * fmaf(x2, 0.10294324f, 2.3022082f);     // 2 u log2(e): 2 * 0.7978845608 * 1.4426950409 * (1 + 0.044715 x^2)
  a = fminf(a, 100.0f);
  const float E = __builtin_amdgcn_exp2f(a);
  r = __builtin_amdgcn_rcpf(1.0f + E);
  s = E * r;
}
__device__ __forceinline__ float gelu_tanh(float x) {
  float s, r;
  gelu_tanh_parts(x, x * x, s, r);
  return x * s;
}
__device__ __forceinline__ float dgelu_tanh(float x) {
  const float x2 = x * x;
  float s, r;
  gelu_tanh_parts(x, x2, s, r);
  const float du2 = fmaf(x2, 0.21406445f, 1.5957691f);  // 2 sqrt(2/pi) (1 + 0.134145 x^2)
  return fmaf(x * du2 * r, s, s);
}
__device__ __forceinline__ float apply_act(int act, float v) {
  return act == TECM_ACT_GELU_ERF ? gelu_erf(v) : (act == TECM_ACT_GELU_TANH ? gelu_tanh(v) : v);
}
__device__ __forceinline__ float apply_dact(int act, float v) {
  return act == TECM_ACT_GELU_ERF ? dgelu_erf(v) : (act == TECM_ACT_GELU_TANH ? dgelu_tanh(v) : 1.0f);
}

// ------------------------------------------------------------------ wave / block reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over aligned groups of 16 lanes (result in every lane of the group)
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------ temporal-window row view
// Resolved once per (thread, row): everything the inner loop needs to address element kk.
// Kept to 3 registers because the narrow-vector GEMM variants hold up to 16 of them per operand.
#define TECM_ROW_INVALID INT32_MIN
struct RowRef {
  int64_t srow;     // source row index of tap 0 (may be negative when t0 < 0; only used when the tap is valid)
  int32_t t0;       // t_out*stride_t - pad, or TECM_ROW_INVALID when the row is out of range
};

__device__ __forceinline__ RowRef make_rowref(const TecmWin& w, bool win, int64_t m, int64_t rows) {
  RowRef r;
  if (m >= rows) {
    r.srow = 0;
    r.t0 = TECM_ROW_INVALID;
    return r;
  }
  if (!win) {
    r.srow = m;
    r.t0 = 0;
    return r;
  }
  const int64_t q = m / w.N;
  const int32_t n = (int32_t)(m - q * w.N);
  const int64_t bq = q / w.Lout;
  const int32_t t_out = (int32_t)(q - bq * w.Lout);
  r.t0 = t_out * w.stride_t - w.pad;
  r.srow = (bq * w.Lin + r.t0) * (int64_t)w.N + n;
  return r;
}
