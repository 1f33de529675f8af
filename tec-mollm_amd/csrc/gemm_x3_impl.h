// "bf16x3" GEMM: fp32 operands, each split on the way into LDS into a bf16 head and a bf16 remainder
//     x = hi + lo + eps,  hi = bf16(x),  lo = bf16(x - hi),  |eps| <= 2^-17 |x|
// and every product evaluated as three bf16 matrix-core products  lo.hi + hi.lo + hi.hi  (fp32 accumulate).
// On gfx950 the bf16 MFMA runs at 16x the rate of the exact-f32 MFMA (v_mfma_f32_32x32x16_bf16: 32 Kflop in
// 8 passes; v_mfma_f32_32x32x2_f32: 4 Kflop in 16), so three of them still cost 3/16 of the exact product.  The
// result carries ~16 mantissa bits per factor (relative error of a dot product ~1e-5): two orders inside the
// 1e-3 parity bar, but NOT exact fp32 -- an opt-in mode (model_config["precision"] = "bf16x3"), never the default.
//
// The same kernel with a THREE-way split (hi + mid + lo carries all 24 mantissa bits of an fp32 value) and the six
// products  hi.lo + lo.hi + mid.mid + hi.mid + mid.hi + hi.hi  ("bf16x6") drops only terms below 2^-24 of the
// product: fp32-grade accuracy (GEMM error vs fp64 ~ the exact kernel's) at 6/16 of the exact matrix time.
//
// Scope: the plain dense contraction  C = epilogue(A[M,K] . B[N,K]^T)  with both operands [row][k] (MK x NK), no
// window view / dropout prologue on A or B -- i.e. the eight GPT-2 GEMMs per layer (the forward ones read the
// cached [N][K] copies of the frozen weights).  Everything else stays on the exact kernel.
//
// Block = 512 threads = 8 waves as 4(m) x 2(n), tile 256 x 128 x BK (x3: BK = 32, x6: BK = 16); a wave owns
// 64 x 64 = 2 x 2 MFMA tiles and issues 24 MFMAs per K-tile either way.  LDS per buffer: NS images of A and B,
// [row][k] bf16 with a (BK + 8)-element pitch (a fragment = one conflict-free ds_read_b128): 61 / 55 KiB; two
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
constexpr int NTH = 512;

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

// three-way split: hi + mid + lo reproduces every mantissa bit of a finite fp32 value
__device__ __forceinline__ void split3(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  bf16x2 h, m, l;
  h[0] = (__bf16)a;
  h[1] = (__bf16)b;
  const float ra = a - (float)h[0], rb = b - (float)h[1];
  m[0] = (__bf16)ra;
  m[1] = (__bf16)rb;
  l[0] = (__bf16)(ra - (float)m[0]);
  l[1] = (__bf16)(rb - (float)m[1]);
  hi = __builtin_bit_cast(uint32_t, h);
  mid = __builtin_bit_cast(uint32_t, m);
  lo = __builtin_bit_cast(uint32_t, l);
}

// [row][k] source, float4 along k; rows pre-clamped (out-of-range rows feed accumulator rows that are never stored)
template <int ROWS, int NS, int BK>
struct SplitStager {
  static constexpr int LDH = BK + 8;
  static constexpr int IMG = ROWS * LDH;              // elements of one image
  static constexpr int VPR = BK / 4;                  // vectors per row
  static constexpr int NV = ROWS * VPR / NTH;
  static constexpr int RSTEP = NTH / VPR;             // rows between a thread's vectors
  static_assert(NV >= 1 && ROWS * VPR % NTH == 0, "tile / thread mapping");
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
  // images of one operand are consecutive in LDS: [hi | (mid) | lo]
  template <int IB, int IE, bool MASKED>
  __device__ __forceinline__ void store_part(__bf16* img) {
    if constexpr (IB >= IE) return;
    const int cv = (threadIdx.x % VPR) * 4;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (!MASKED || ((okmask >> i) & 1u)) ? regs[i][e] : 0.f;
      const int off = (r0 + i * RSTEP) * LDH + cv;
      if constexpr (NS == 2) {
        uint2 h, l;
        split2(v[0], v[1], h.x, l.x);
        split2(v[2], v[3], h.y, l.y);
        *reinterpret_cast<uint2*>(img + off) = h;
        *reinterpret_cast<uint2*>(img + IMG + off) = l;
      } else {
        uint2 h, m, l;
        split3(v[0], v[1], h.x, m.x, l.x);
        split3(v[2], v[3], h.y, m.y, l.y);
        *reinterpret_cast<uint2*>(img + off) = h;
        *reinterpret_cast<uint2*>(img + IMG + off) = m;
        *reinterpret_cast<uint2*>(img + 2 * IMG + off) = l;
      }
    }
  }
};

template <int NS, int BK>
__global__ __launch_bounds__(NTH, 2) void gemm_x3_kernel(const TecmGemm g, int tiles_m, int tiles_n, int k_chunk) {
  constexpr int WN = 2, WM = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;          // 64 x 64 per wave
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int KSTEPS = BK / 16;
  using AStager = SplitStager<BM, NS, BK>;
  using BStager = SplitStager<BN, NS, BK>;
  constexpr int LDH = AStager::LDH;
  constexpr int A_IMG = AStager::IMG, B_IMG = BStager::IMG;
  constexpr int TILE_ELEMS = NS * (A_IMG + B_IMG);     // [A images | B images], bf16 elements
  constexpr int STG_LD = WTN + 4;
  constexpr int STG_BYTES = 8 * 32 * STG_LD * 4;
  constexpr int SMEM_BYTES = 2 * TILE_ELEMS * 2 > STG_BYTES ? 2 * TILE_ELEMS * 2 : STG_BYTES;
  static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
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

  // prologue: K-tile 0 -> LDS buffer 0, K-tile 1 -> registers (in flight)
  sa.template load_part<0, ANV>(g.A, kend);
  sb.template load_part<0, BNV>(g.B, kend);
  sa.template store_part<0, ANV, true>(smem);
  sb.template store_part<0, BNV, true>(smem + NS * A_IMG);
  if (kbeg + BK < kend) {
    sa.template load_part<0, ANV>(g.A, kend);
    sb.template load_part<0, BNV>(g.B, kend);
  }
  __syncthreads();

  int cur = 0;
  auto tile = [&](int32_t k0, auto fullc) {
    constexpr bool FULL = decltype(fullc)::value;         // K-tiles t+1 and t+2 lie entirely inside [kbeg, kend)
    const __bf16* aimg = smem + cur * TILE_ELEMS;
    const __bf16* bimg = aimg + NS * A_IMG;
    __bf16* naimg = smem + (cur ^ 1) * TILE_ELEMS;
    __bf16* nbimg = naimg + NS * A_IMG;
    static_for<KSTEPS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;              // k-step of 16 inside the tile
      bf16x8 af[NS][MT], bf[NS][NT];                      // [0] = hi ... [NS-1] = lo
#pragma unroll
      for (int p = 0; p < NS; ++p) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[p][i] = *reinterpret_cast<const bf16x8*>(aimg + p * A_IMG + (wm * WTM + i * 32 + r) * LDH + 16 * s + 8 * h);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bf[p][j] = *reinterpret_cast<const bf16x8*>(bimg + p * B_IMG + (wn * WTN + j * 32 + r) * LDH + 16 * s + 8 * h);
      }
      auto prod = [&](auto pa, auto pb) {
        constexpr int PA = decltype(pa)::value, PB = decltype(pb)::value;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][i], bf[PB][j], acc[i][j], 0, 0, 0);
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      // smallest terms first, the head product last
      if constexpr (NS == 2) {
        prod(I1{}, I0{}); prod(I0{}, I1{}); prod(I0{}, I0{});
      } else {
        prod(I0{}, I2{}); prod(I2{}, I0{}); prod(I1{}, I1{}); prod(I0{}, I1{}); prod(I1{}, I0{}); prod(I0{}, I0{});
      }
      // 1/KSTEPS of the staging work per k-step
      constexpr int AB = (ANV * s) / KSTEPS, AE = (ANV * (s + 1)) / KSTEPS;
      constexpr int BB = (BNV * s) / KSTEPS, BE = (BNV * (s + 1)) / KSTEPS;
      if constexpr (FULL) {
        sa.template store_part<AB, AE, false>(naimg);
        sb.template store_part<BB, BE, false>(nbimg);
        sa.template load_steady<AB, AE>();
        sb.template load_steady<BB, BE>();
      } else {
        if (k0 + BK < kend) {
          sa.template store_part<AB, AE, true>(naimg);
          sb.template store_part<BB, BE, true>(nbimg);
        }
        if (k0 + 2 * BK < kend) {
          sa.template load_part<AB, AE>(g.A, kend);
          sb.template load_part<BB, BE>(g.B, kend);
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
    if (g.io_bf16 & TECM_P0_VEC4) {
      constexpr int LPR = WTN / 4, RPI = 64 / LPR;
      const int lcol = (lane % LPR) * 4, lrow = lane / LPR;
      const EpiCol ecol = epi_col(g, n0 + wn * WTN + lcol);
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + (ecol.ok ? ecol.n : 0));   // (lanes outside the matrix: column 0, never stored)
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

template <int NS, int BK>
int launch_split(const TecmGemm& g, hipStream_t st) {
  const int tiles_m = (int)((g.M + BM - 1) / BM);
  const int tiles_n = (int)((g.N + BN - 1) / BN);
  int splits = g.split_k > 1 ? g.split_k : 1;
  int k_chunk = (int)(((g.K + splits - 1) / splits + BK - 1) / BK) * BK;
  splits = (int)((g.K + k_chunk - 1) / k_chunk);
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)splits);
  hipLaunchKernelGGL((gemm_x3_kernel<NS, BK>), grid, dim3(NTH), 0, st, g, tiles_m, tiles_n, k_chunk);
  TECM_CHECK_LAUNCH("tecm_gemm_bf16x3/x6");
  return splits;
}

}  // namespace tecm_gemm3

int tecm_gemm_x3_dispatch(const TecmGemm& g, int products, hipStream_t st);      // products = 3 or 6
