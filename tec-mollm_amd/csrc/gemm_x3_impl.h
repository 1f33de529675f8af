// "bf16x3" GEMM: fp32 operands, each split on the way into LDS into a bf16 head and a bf16 remainder
//     x = hi + lo + eps,  hi = bf16(x),  lo = bf16(x - hi),  |eps| <= 2^-17 |x|
// and every product evaluated as three bf16 matrix-core products  lo.hi + hi.lo + hi.hi  (fp32 accumulate).
// On gfx950 the bf16 MFMA runs at 16x the rate of the exact-f32 MFMA (v_mfma_f32_32x32x16_bf16: 32 Kflop in
// 8 passes; v_mfma_f32_32x32x2_f32: 4 Kflop in 16), so three of them still cost 3/16 of the exact product.  The
// result carries ~16 mantissa bits per factor (relative error of a dot product ~1e-5): two orders inside the
// 1e-3 parity bar, but NOT exact fp32 -- an opt-in mode (model_config["precision"] = "bf16x3"), never the default.
//
// Scope: the plain dense contraction  C = epilogue(A[M,K] . B[N,K]^T)  with both operands [row][k] (MK x NK), no
// window view / dropout prologue on A or B -- i.e. the eight GPT-2 GEMMs per layer (the forward ones read the
// cached [N][K] copies of the frozen weights).  Everything else stays on the exact kernel.
//
// Block = 512 threads = 8 waves as 4(m) x 2(n), tile 256 x 128 x 32; a wave owns 64 x 64 = 2 x 2 MFMA tiles and
// issues 2 k-steps x 4 tiles x 3 products = 24 MFMAs per K-tile.  LDS per buffer: hi and lo images of A and B,
// [row][k] bf16 with an 80-byte pitch (32 k + 8 pad: a fragment = one conflict-free ds_read_b128), 61 KiB; two
// buffers, one barrier per K-tile, same register-staged pipeline and epilogue as the bf16 kernel.
#pragma once
#include "gemm_bf16_impl.h"

namespace tecm_gemm3 {

using tecm_gemm::DropCtx;
using tecm_gemm::EpiCol;
using tecm_gemm::EpiRow;
using tecm_gemm::epi_col;
using tecm_gemm::epi_elem;
using tecm_gemm::epi_row;
using tecm_gemm::epi_vec4;
using tecm_gemm::gload;
using tecm_gemm::make_drop;
using tecm_gemm::static_for;
using tecm_gemm16::bf16x2;
using tecm_gemm16::bf16x8;

constexpr int BM = 256;
constexpr int BN = 128;
constexpr int BK = 32;
constexpr int NTH = 512;
constexpr int LDH = BK + 8;          // 80-byte rows

// hi/lo split of a float pair, packed as two bf16x2 words
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  bf16x2 h;
  h[0] = (__bf16)a;
  h[1] = (__bf16)b;
  bf16x2 l;
  l[0] = (__bf16)(a - (float)h[0]);
  l[1] = (__bf16)(b - (float)h[1]);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// [row][k] source, float4 along k; rows pre-clamped (out-of-range rows feed accumulator rows that are never stored)
template <int ROWS>
struct SplitStager {
  static constexpr int VPR = BK / 4;                  // 8 vectors per row
  static constexpr int NV = ROWS * VPR / NTH;         // 4 (A) or 2 (B)
  static constexpr int RSTEP = NTH / VPR;             // 64 rows between a thread's vectors
  float regs[NV][4];
  const float* ptr[NV];
  uint32_t okmask;                                    // bit i: vector i of the tile in regs lies inside K
  int32_t kk;

  __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, int64_t row0, int64_t rows_total,
                                       int32_t kbeg) {
    const int cv = (threadIdx.x % VPR) * 4;
    const int r0 = threadIdx.x / VPR;
    okmask = 0;
    kk = kbeg + cv;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int64_t row = row0 + r0 + i * RSTEP;
      row = row < rows_total ? row : rows_total - 1;
      ptr[i] = P + row * ld + kbeg + cv;
    }
  }
  // masked form: zero-fill past klim (K tails, split-K chunk ends)
  template <int IB, int IE>
  __device__ __forceinline__ void load_part(const float* __restrict__ P, int32_t klim) {
    if constexpr (IB >= IE) return;
    const bool ok = kk < klim;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      gload<4>(ptr[i], P, ok, regs[i]);
      okmask = (okmask & ~(1u << i)) | ((ok ? 1u : 0u) << i);
      ptr[i] += BK;
    }
    if constexpr (IE == NV) kk += BK;
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_steady() {
    if constexpr (IB >= IE) return;
    if constexpr (IB == 0) okmask = ~0u;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      gload<4>(ptr[i], ptr[i], true, regs[i]);
      ptr[i] += BK;
    }
    if constexpr (IE == NV) kk += BK;
  }
  template <int IB, int IE, bool MASKED>
  __device__ __forceinline__ void store_part(__bf16* hi_img, __bf16* lo_img) {
    if constexpr (IB >= IE) return;
    const int cv = (threadIdx.x % VPR) * 4;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (!MASKED || ((okmask >> i) & 1u)) ? regs[i][e] : 0.f;
      uint2 h, l;
      split2(v[0], v[1], h.x, l.x);
      split2(v[2], v[3], h.y, l.y);
      const int off = (r0 + i * RSTEP) * LDH + cv;
      *reinterpret_cast<uint2*>(hi_img + off) = h;
      *reinterpret_cast<uint2*>(lo_img + off) = l;
    }
  }
};

__global__ __launch_bounds__(NTH, 2) void gemm_x3_kernel(const TecmGemm g, int tiles_m, int tiles_n, int k_chunk) {
  constexpr int WN = 2, WM = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;          // 64 x 64 per wave
  constexpr int MT = WTM / 32, NT = WTN / 32;
  using AStager = SplitStager<BM>;
  using BStager = SplitStager<BN>;
  constexpr int A_ELEMS = BM * LDH, B_ELEMS = BN * LDH;
  constexpr int TILE_ELEMS = 2 * (A_ELEMS + B_ELEMS);  // [A_hi | A_lo | B_hi | B_lo], bf16 elements
  constexpr int STG_LD = WTN + 4;
  constexpr int STG_BYTES = 8 * 32 * STG_LD * 4;
  constexpr int SMEM_BYTES = 2 * TILE_ELEMS * 2 > STG_BYTES ? 2 * TILE_ELEMS * 2 : STG_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);

  // XCD-aware bijective block -> tile map with 4-m-tile groups (see gemm_impl.h)
  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * BM;
  const int64_t n0 = (int64_t)tn * BN;
  const int32_t kbeg = blockIdx.z * k_chunk;
  const int32_t kend = min((int32_t)g.K, kbeg + k_chunk);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  AStager sa;
  BStager sb;
  sa.init(g.A, g.lda, m0, g.M, kbeg);
  sb.init(g.B, g.ldb, n0, g.N, kbeg);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  constexpr int ANV = AStager::NV, BNV = BStager::NV;
  auto images = [&](int buf, __bf16*& ah, __bf16*& al, __bf16*& bh, __bf16*& bl) {
    ah = smem + buf * TILE_ELEMS;
    al = ah + A_ELEMS;
    bh = al + A_ELEMS;
    bl = bh + B_ELEMS;
  };

  // prologue: K-tile 0 -> LDS buffer 0, K-tile 1 -> registers (in flight)
  {
    __bf16 *ah, *al, *bh, *bl;
    images(0, ah, al, bh, bl);
    sa.load_part<0, ANV>(g.A, kend);
    sb.load_part<0, BNV>(g.B, kend);
    sa.store_part<0, ANV, true>(ah, al);
    sb.store_part<0, BNV, true>(bh, bl);
    if (kbeg + BK < kend) {
      sa.load_part<0, ANV>(g.A, kend);
      sb.load_part<0, BNV>(g.B, kend);
    }
  }
  __syncthreads();

  int cur = 0;
  auto tile = [&](int32_t k0, auto fullc) {
    constexpr bool FULL = decltype(fullc)::value;         // K-tiles t+1 and t+2 lie entirely inside [kbeg, kend)
    __bf16 *ah, *al, *bh, *bl, *nah, *nal, *nbh, *nbl;
    images(cur, ah, al, bh, bl);
    images(cur ^ 1, nah, nal, nbh, nbl);
    static_for<2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;              // k-step of 16 inside the 32-deep tile
      bf16x8 afh[MT], afl[MT], bfh[NT], bfl[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int off = (wm * WTM + i * 32 + r) * LDH + 16 * s + 8 * h;
        afh[i] = *reinterpret_cast<const bf16x8*>(ah + off);
        afl[i] = *reinterpret_cast<const bf16x8*>(al + off);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int off = (wn * WTN + j * 32 + r) * LDH + 16 * s + 8 * h;
        bfh[j] = *reinterpret_cast<const bf16x8*>(bh + off);
        bfl[j] = *reinterpret_cast<const bf16x8*>(bl + off);
      }
      // the two cross terms first, the head product last
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl[i], bfh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[i], bfl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[i], bfh[j], acc[i][j], 0, 0, 0);
      // half of the staging work per k-step
      constexpr int AB = (ANV * s) / 2, AE = (ANV * (s + 1)) / 2;
      constexpr int BB = (BNV * s) / 2, BE = (BNV * (s + 1)) / 2;
      if constexpr (FULL) {
        sa.store_part<AB, AE, false>(nah, nal);
        sb.store_part<BB, BE, false>(nbh, nbl);
        sa.load_steady<AB, AE>();
        sb.load_steady<BB, BE>();
      } else {
        if (k0 + BK < kend) {
          sa.store_part<AB, AE, true>(nah, nal);
          sb.store_part<BB, BE, true>(nbh, nbl);
        }
        if (k0 + 2 * BK < kend) {
          sa.load_part<AB, AE>(g.A, kend);
          sb.load_part<BB, BE>(g.B, kend);
        }
      }
    });
    __syncthreads();
    cur ^= 1;
  };
  int32_t k0 = kbeg;
  for (; k0 + 3 * BK <= kend; k0 += BK) tile(k0, std::true_type{});
  for (; k0 < kend; k0 += BK) tile(k0, std::false_type{});

  // ---- epilogue: as the bf16 kernel's, two 32-row slabs per wave through LDS
  const DropCtx odc = make_drop(g.out_drop);
  const bool split = gridDim.z > 1;
  float* stg = reinterpret_cast<float*>(smem_raw) + wave * (32 * STG_LD);
  static_for<MT>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if (i > 0) __syncthreads();
    static_for<16>([&](auto ec) {
      constexpr int e = decltype(ec)::value;
      static_for<NT>([&](auto jc) {
        constexpr int jn = decltype(jc)::value;
        stg[((e & 3) + 8 * (e >> 2) + 4 * h) * STG_LD + jn * 32 + r] = acc[i][jn][e];
      });
    });
    __syncthreads();
    if (g._p0 != 0) {
      constexpr int LPR = WTN / 4, RPI = 64 / LPR;
      const int lcol = (lane % LPR) * 4, lrow = lane / LPR;
      const EpiCol ecol = epi_col(g, n0 + wn * WTN + lcol);
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.bias && ecol.ok) bias4 = *reinterpret_cast<const float4*>(g.bias + ecol.n);
#pragma unroll 1
      for (int it = 0; it < 32 / RPI; ++it) {
        const int rl = it * RPI + lrow;
        const int64_t m = m0 + wm * WTM + i * 32 + rl;
        if (m < g.M && ecol.ok) {
          const float4 v = *reinterpret_cast<const float4*>(&stg[rl * STG_LD + lcol]);
          if (split) {
            *reinterpret_cast<float4*>(g.workspace + ((int64_t)blockIdx.z * g.M + m) * g.N + ecol.n) = v;
          } else {
            const EpiRow er = epi_row(g, odc, m);
            epi_vec4(g, odc, er, ecol, bias4, v);
          }
        }
      }
    } else {
      const int lcol = lane % WTN, lrow = lane / WTN;
      const EpiCol ecol = epi_col(g, n0 + wn * WTN + lcol);
#pragma unroll 1
      for (int it = 0; it < 32; ++it) {
        const int rl = it + lrow;
        const int64_t m = m0 + wm * WTM + i * 32 + rl;
        if (m < g.M && ecol.ok) {
          const float v = stg[rl * STG_LD + lcol];
          if (split) {
            g.workspace[((int64_t)blockIdx.z * g.M + m) * g.N + ecol.n] = v;
          } else {
            const EpiRow er = epi_row(g, odc, m);
            epi_elem(g, odc, er, ecol, v);
          }
        }
      }
    }
  });
}

inline int launch_x3(const TecmGemm& g, hipStream_t st) {
  const int tiles_m = (int)((g.M + BM - 1) / BM);
  const int tiles_n = (int)((g.N + BN - 1) / BN);
  int splits = g.split_k > 1 ? g.split_k : 1;
  int k_chunk = (int)(((g.K + splits - 1) / splits + BK - 1) / BK) * BK;
  splits = (int)((g.K + k_chunk - 1) / k_chunk);
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)splits);
  hipLaunchKernelGGL(gemm_x3_kernel, grid, dim3(NTH), 0, st, g, tiles_m, tiles_n, k_chunk);
  TECM_CHECK_LAUNCH("tecm_gemm_bf16x3");
  return splits;
}

}  // namespace tecm_gemm3

int tecm_gemm_x3_dispatch(const TecmGemm& g, hipStream_t st);
