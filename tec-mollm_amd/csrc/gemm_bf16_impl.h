// bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for BASELINE configs[2]
// ("bf16 autocast, MFMA QKV/proj"): same descriptor, views, prologue dropout and epilogue as the fp32 kernel
// (gemm_impl.h); only the operand path differs.
//
//   * tensors stay fp32 in HBM; each operand element is rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while it is
//     staged into LDS -- exactly what torch.autocast does to the inputs of Linear / Conv1d / Conv1D
//     (reference train.py:68), with fp32 accumulation and fp32 outputs;
//   * block = 512 threads = 8 waves as 4(m) x 2(n), tile 256 x 128 x 64; a wave owns 64 x 64 = 2 x 2 MFMA
//     tiles (64 accumulator VGPRs) and issues 16 MFMAs per K-tile;
//   * LDS tiles are ALWAYS [row][k] bf16 with a 144-byte row pitch (64 k + 8 pad): a fragment (8 consecutive k
//     of one row) is one conflict-free ds_read_b128.  Sources whose contiguous dimension is the row index
//     ([k][m] / [k][n]: the KM / KN layouts) are transposed on the way in: a thread loads the float4s of two
//     adjacent k rows and writes four packed (k, k+1) bf16 pairs with ds_write_b32;
//   * two LDS operand buffers, one barrier per K-tile, tile t+1 parked / tile t+2 fetched between the MFMAs
//     of tile t (same pipeline as the fp32 kernel).
#pragma once
#include "gemm_impl.h"
#include <type_traits>

namespace tecm_gemm16 {

using tecm_gemm::DropCtx;
using tecm_gemm::EpiCol;
using tecm_gemm::EpiRow;
using tecm_gemm::WinRow;
using tecm_gemm::apply_drop;
using tecm_gemm::epi_col;
using tecm_gemm::epi_elem;
using tecm_gemm::epi_row;
using tecm_gemm::epi_vec4;
using tecm_gemm::gload;
using tecm_gemm::make_drop;
using tecm_gemm::static_for;
using tecm_gemm::win_row;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 256;
constexpr int BN = 128;
constexpr int BK = 64;
constexpr int NTH = 512;
constexpr int LDH = BK + 8;          // LDS row pitch in bf16 elements (144 B: conflict-free b128 fragment reads)

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;                 // plain casts: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-safe)
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

// ---------------------------------------------------------------------------------------------------------
// Direct stager: source rows are the tile rows, k contiguous ([m][k] / [n][k]).  One vector = float4 along k.
template <int ROWS, bool WIN, bool DROP>
struct DStager {
  static constexpr int VPR = BK / 4;                  // 16 vectors per row
  static constexpr int NV = ROWS * VPR / NTH;         // 8 (A) or 4 (B)
  static constexpr int RSTEP = NTH / VPR;             // 32 rows between a thread's vectors
  static constexpr int NITEMS = NV;
  float regs[NV][4];
  const float* ptr[NV];
  uint32_t rowok, okbits;
  int64_t didx[DROP ? NV : 1], dsave[DROP ? NV : 1];
  WinRow wr[WIN ? NV : 1];
  int32_t tap, c, kk;

  __device__ __forceinline__ void init(const float* __restrict__ P, const TecmWin& w, int64_t ld, int64_t row0,
                                       int64_t rows_total, int32_t kbeg, const DropCtx& dc) {
    const int cv = (threadIdx.x % VPR) * 4;
    const int r0 = threadIdx.x / VPR;
    rowok = 0;
    okbits = 0;
    tap = 0;
    c = 0;
    kk = kbeg + cv;
    const bool wen = WIN && w.enabled;
    if (!wen) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int64_t row = row0 + r0 + i * RSTEP;
        if (row < rows_total) rowok |= 1u << i;
        else row = rows_total - 1;            // clamped: a valid address; the tile row it feeds is never stored
        ptr[i] = P + row * ld + kbeg + cv;
        if constexpr (DROP) didx[i] = row * dc.ld + kbeg + cv;
      }
    } else if constexpr (WIN) {
#pragma unroll
      for (int i = 0; i < NV; ++i) wr[i] = win_row(w, row0 + r0 + i * RSTEP, rows_total);
      tap = kk / w.Cw;
      c = kk - tap * w.Cw;
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_part(const float* __restrict__ P, const TecmWin& w, int64_t ld, int32_t k0,
                                            int32_t klim, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    constexpr bool LAST = IE == NV;
    const bool wen = WIN && w.enabled;
    const bool kok = kk < klim;
    if (!wen) {
#pragma unroll
      for (int i = IB; i < IE; ++i) {
        const bool ok = kok && ((rowok >> i) & 1u);
        gload<4>(ptr[i], P, ok, regs[i]);
        okbits = (okbits & ~(1u << i)) | ((ok ? 1u : 0u) << i);
        if constexpr (DROP) { dsave[i] = didx[i]; didx[i] += BK; }
        ptr[i] += BK;
      }
      if constexpr (LAST) kk += BK;
    } else if constexpr (WIN) {
      const int64_t tapoff = (int64_t)tap * w.N;
#pragma unroll
      for (int i = IB; i < IE; ++i) {
        const int32_t t_in = wr[i].t0 + tap;
        const bool ok = kok && t_in >= 0 && t_in < w.Lin;
        const int64_t row = wr[i].srow + tapoff;
        gload<4>(P + row * ld + c, P, ok, regs[i]);
        okbits = (okbits & ~(1u << i)) | ((ok ? 1u : 0u) << i);
        if constexpr (DROP) dsave[i] = row * dc.ld + c;
      }
      if constexpr (LAST) {
        kk += BK;
        c += BK;
        while (c >= w.Cw) { c -= w.Cw; ++tap; }
      }
    }
    (void)k0;
  }
  // Steady state of the plain view (tile entirely inside [kbeg, kend), rows pre-clamped): no masks, no selects.
  template <int IB, int IE>
  __device__ __forceinline__ void load_steady(int64_t /*ld*/, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    if constexpr (IB == 0) okbits = ~0u;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      gload<4>(ptr[i], ptr[i], true, regs[i]);
      if constexpr (DROP) { dsave[i] = didx[i]; didx[i] += BK; }
      ptr[i] += BK;
    }
    if constexpr (IE == NV) kk += BK;
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_steady(__bf16* lds, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    const int cv = (threadIdx.x % VPR) * 4;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = regs[i][e];
      if constexpr (DROP) apply_drop<4>(dc, dsave[i], v);
      uint2 pk;
      pk.x = pack_bf16(v[0], v[1]);
      pk.y = pack_bf16(v[2], v[3]);
      *reinterpret_cast<uint2*>(lds + (r0 + i * RSTEP) * LDH + cv) = pk;
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_part(__bf16* lds, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    const int cv = (threadIdx.x % VPR) * 4;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      float v[4];
      const bool ok = (okbits >> i) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? regs[i][e] : 0.f;
      if constexpr (DROP) apply_drop<4>(dc, dsave[i], v);
      uint2 pk;
      pk.x = pack_bf16(v[0], v[1]);
      pk.y = pack_bf16(v[2], v[3]);
      *reinterpret_cast<uint2*>(lds + (r0 + i * RSTEP) * LDH + cv) = pk;
    }
  }
};

// Transposing stager: source rows are k, the tile-row index is contiguous ([k][m] / [k][n]).
// One item = k rows (2kp, 2kp+1) x 4 consecutive tile rows: two float4 loads, four ds_write_b32 of (k,k+1) pairs.
template <int ROWS, bool WIN, bool DROP>
struct TStager {
  static constexpr int QPR = ROWS / 4;                // row quads per k pair
  static constexpr int NI = (BK / 2) * QPR / NTH;     // 4 (A, 256 rows) or 2 (B, 128 rows)
  static constexpr int KSTEP = NTH / QPR;             // k pairs between a thread's items
  static constexpr int NITEMS = NI;
  float regs[NI][8];
  const float* ptr[NI];
  bool inner_ok;
  uint32_t okbits;                                    // bit 2i: row 2kp valid, bit 2i+1: row 2kp+1 valid
  int64_t didx[DROP ? NI : 1], dsave[DROP ? NI : 1], dsave_hi[DROP ? NI : 1];
  int32_t tap, c, inner;
  // window view (weight-gradient form): (n, t_out, b) of view row k0 + 2*(kp0 + i*KSTEP), advanced by BK rows per
  // K-tile instead of two integer divisions per row per tile
  int32_t rn[WIN ? NI : 1], rt[WIN ? NI : 1], rb[WIN ? NI : 1];

  __device__ __forceinline__ void init(const float* __restrict__ P, const TecmWin& w, int64_t ld, int32_t kbeg,
                                       int64_t fixed0, int64_t fixed_lim, const DropCtx& dc) {
    const int q4 = (threadIdx.x % QPR) * 4;
    const int kp0 = threadIdx.x / QPR;
    inner = (int32_t)(fixed0 + q4);
    inner_ok = inner < fixed_lim;
    if (!inner_ok) inner = 0;                 // clamped column quad (valid address, never stored); WIN: tap/c of column 0
    okbits = 0;
    tap = 0;
    c = 0;
    const bool wen = WIN && w.enabled;
    if (!wen) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int64_t krow = (int64_t)kbeg + 2 * (kp0 + i * KSTEP);
        ptr[i] = P + krow * ld + inner;
        if constexpr (DROP) didx[i] = krow * dc.ld + inner;
      }
    } else if constexpr (WIN) {
      tap = inner / w.Cw;
      c = inner - tap * w.Cw;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint32_t row = (uint32_t)(kbeg + 2 * (kp0 + i * KSTEP));
        const uint32_t q = row / (uint32_t)w.N;
        rn[i] = (int32_t)(row - q * (uint32_t)w.N);
        rb[i] = (int32_t)(q / (uint32_t)w.Lout);
        rt[i] = (int32_t)(q - (uint32_t)rb[i] * (uint32_t)w.Lout);
      }
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_part(const float* __restrict__ P, const TecmWin& w, int64_t ld, int32_t k0,
                                            int32_t klim, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    const int kp0 = threadIdx.x / QPR;
    const bool wen = WIN && w.enabled;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      const int32_t krow = k0 + 2 * (kp0 + i * KSTEP);
      bool ok0, ok1;
      float lo[4], hi[4];
      if (!wen) {
        ok0 = inner_ok && krow < klim;
        ok1 = inner_ok && krow + 1 < klim;
        gload<4>(ptr[i], P, ok0, lo);
        gload<4>(ptr[i] + ld, P, ok1, hi);
        if constexpr (DROP) { dsave[i] = didx[i]; dsave_hi[i] = didx[i] + dc.ld; didx[i] += (int64_t)BK * dc.ld; }
        ptr[i] += (int64_t)BK * ld;
      } else {
        // window rows are the reduction index (weight-gradient form): view rows krow and krow + 1
        if constexpr (WIN) {
          int32_t n1 = rn[i] + 1, t1 = rt[i], b1 = rb[i];                  // row krow + 1
          if (n1 >= w.N) { n1 = 0; if (++t1 >= w.Lout) { t1 = 0; ++b1; } }
          const int32_t ta = rt[i] * w.stride_t - w.pad + tap, tb = t1 * w.stride_t - w.pad + tap;
          ok0 = inner_ok && krow < klim && ta >= 0 && ta < w.Lin;
          ok1 = inner_ok && krow + 1 < klim && tb >= 0 && tb < w.Lin;
          const int64_t rowa = ((int64_t)rb[i] * w.Lin + ta) * (int64_t)w.N + rn[i];
          const int64_t rowb = ((int64_t)b1 * w.Lin + tb) * (int64_t)w.N + n1;
          gload<4>(P + rowa * ld + c, P, ok0, lo);
          gload<4>(P + rowb * ld + c, P, ok1, hi);
          if constexpr (DROP) { dsave[i] = rowa * dc.ld + c; dsave_hi[i] = rowb * dc.ld + c; }
          rn[i] += BK;                                                      // next K-tile
          while (rn[i] >= w.N) {
            rn[i] -= w.N;
            if (++rt[i] >= w.Lout) { rt[i] = 0; ++rb[i]; }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { regs[i][e] = lo[e]; regs[i][4 + e] = hi[e]; }
      okbits = (okbits & ~(3u << (2 * i))) | ((ok0 ? 1u : 0u) << (2 * i)) | ((ok1 ? 1u : 0u) << (2 * i + 1));
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_steady(int64_t ld, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    if constexpr (IB == 0) okbits = ~0u;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      float lo[4], hi[4];
      gload<4>(ptr[i], ptr[i], true, lo);
      gload<4>(ptr[i] + ld, ptr[i], true, hi);
      if constexpr (DROP) { dsave[i] = didx[i]; dsave_hi[i] = didx[i] + dc.ld; didx[i] += (int64_t)BK * dc.ld; }
      ptr[i] += (int64_t)BK * ld;
#pragma unroll
      for (int e = 0; e < 4; ++e) { regs[i][e] = lo[e]; regs[i][4 + e] = hi[e]; }
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_steady(__bf16* lds, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    const int q4 = (threadIdx.x % QPR) * 4;
    const int kp0 = threadIdx.x / QPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      float lo[4], hi[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] = regs[i][e]; hi[e] = regs[i][4 + e]; }
      if constexpr (DROP) {
        apply_drop<4>(dc, dsave[i], lo);
        apply_drop<4>(dc, dsave_hi[i], hi);
      }
      const int kcol = 2 * (kp0 + i * KSTEP);
      uint32_t* base = reinterpret_cast<uint32_t*>(lds + q4 * LDH + kcol);
#pragma unroll
      for (int e = 0; e < 4; ++e) base[e * (LDH / 2)] = pack_bf16(lo[e], hi[e]);
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_part(__bf16* lds, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    const int q4 = (threadIdx.x % QPR) * 4;
    const int kp0 = threadIdx.x / QPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      float lo[4], hi[4];
      const bool ok0 = (okbits >> (2 * i)) & 1u, ok1 = (okbits >> (2 * i + 1)) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] = ok0 ? regs[i][e] : 0.f; hi[e] = ok1 ? regs[i][4 + e] : 0.f; }
      if constexpr (DROP) {
        apply_drop<4>(dc, dsave[i], lo);
        apply_drop<4>(dc, dsave_hi[i], hi);
      }
      const int kcol = 2 * (kp0 + i * KSTEP);
      uint32_t* base = reinterpret_cast<uint32_t*>(lds + q4 * LDH + kcol);
#pragma unroll
      for (int e = 0; e < 4; ++e) base[e * (LDH / 2)] = pack_bf16(lo[e], hi[e]);
    }
  }
};

// bf16 source stager: the operand already lives in HBM as bf16, [row][k] with k contiguous (activations written
// as bf16 by the producing kernel, cached bf16 copies of the frozen weights).  Pure copy: one 16-byte load and one
// ds_write_b128 per 8 k -- no conversion VALU, half the bytes.  Plain view only (no window / dropout prologue).
template <int ROWS>
struct HStager {
  static constexpr int VPR = BK / 8;                  // 8 vectors of 8 bf16 per row
  static constexpr int NV = ROWS * VPR / NTH;         // 4 (A) or 2 (B)
  static constexpr int RSTEP = NTH / VPR;             // 64 rows between a thread's vectors
  static constexpr int NITEMS = NV;
  uint4 regs[NV];
  const __bf16* ptr[NV];
  uint32_t okbits;
  int32_t kk;

  __device__ __forceinline__ void init(const float* __restrict__ P, const TecmWin&, int64_t ld, int64_t row0,
                                       int64_t rows_total, int32_t kbeg, const DropCtx&) {
    const __bf16* Ph = reinterpret_cast<const __bf16*>(P);
    const int cv = (threadIdx.x % VPR) * 8;
    const int r0 = threadIdx.x / VPR;
    okbits = 0;
    kk = kbeg + cv;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int64_t row = row0 + r0 + i * RSTEP;
      row = row < rows_total ? row : rows_total - 1;  // clamped: feeds an accumulator row that is never stored
      ptr[i] = Ph + row * ld + kbeg + cv;
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_part(const float* __restrict__ P, const TecmWin&, int64_t, int32_t, int32_t klim,
                                            const DropCtx&) {
    if constexpr (IB >= IE) return;
    const bool ok = kk < klim;                        // K % 8 == 0: a vector is entirely inside or outside
    const __bf16* safe = reinterpret_cast<const __bf16*>(P);
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      regs[i] = *reinterpret_cast<const uint4*>(ok ? ptr[i] : safe);
      okbits = (okbits & ~(1u << i)) | ((ok ? 1u : 0u) << i);
      ptr[i] += BK;
    }
    if constexpr (IE == NV) kk += BK;
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_steady(int64_t, const DropCtx&) {
    if constexpr (IB >= IE) return;
    if constexpr (IB == 0) okbits = ~0u;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      regs[i] = *reinterpret_cast<const uint4*>(ptr[i]);
      ptr[i] += BK;
    }
    if constexpr (IE == NV) kk += BK;
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_part(__bf16* lds, const DropCtx&) {
    if constexpr (IB >= IE) return;
    const int cv = (threadIdx.x % VPR) * 8;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      const uint4 v = ((okbits >> i) & 1u) ? regs[i] : make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(lds + (r0 + i * RSTEP) * LDH + cv) = v;
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_steady(__bf16* lds, const DropCtx&) {
    if constexpr (IB >= IE) return;
    const int cv = (threadIdx.x % VPR) * 8;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) *reinterpret_cast<uint4*>(lds + (r0 + i * RSTEP) * LDH + cv) = regs[i];
  }
};

// bf16 source, [row][k], with an optional temporal-window view of the rows (decided at run time): the A operand of
// the strided 1x1 conv (gelu(GroupNorm(.)) written as bf16 by gn_gelu_fwd) and of the conv dX GEMMs (dy written as bf16
// by gn_gelu_bwd).  Vectors of 8 bf16 never straddle a tap (host: Cw % 8 == 0).
template <int ROWS>
struct HStagerW {
  static constexpr int VPR = BK / 8;
  static constexpr int NV = ROWS * VPR / NTH;
  static constexpr int RSTEP = NTH / VPR;
  static constexpr int NITEMS = NV;
  uint4 regs[NV];
  const __bf16* ptr[NV];
  WinRow wr[NV];
  uint32_t okbits;
  int32_t kk, tap, c;

  __device__ __forceinline__ void init(const float* __restrict__ P, const TecmWin& w, int64_t ld, int64_t row0,
                                       int64_t rows_total, int32_t kbeg, const DropCtx&) {
    const __bf16* Ph = reinterpret_cast<const __bf16*>(P);
    const int cv = (threadIdx.x % VPR) * 8;
    const int r0 = threadIdx.x / VPR;
    okbits = 0;
    kk = kbeg + cv;
    tap = 0;
    c = 0;
    if (!w.enabled) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int64_t row = row0 + r0 + i * RSTEP;
        row = row < rows_total ? row : rows_total - 1;  // clamped: feeds an accumulator row that is never stored
        ptr[i] = Ph + row * ld + kbeg + cv;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) wr[i] = win_row(w, row0 + r0 + i * RSTEP, rows_total);
      tap = kk / w.Cw;
      c = kk - tap * w.Cw;
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_part(const float* __restrict__ P, const TecmWin& w, int64_t ld, int32_t, int32_t klim,
                                            const DropCtx&) {
    if constexpr (IB >= IE) return;
    const __bf16* Ph = reinterpret_cast<const __bf16*>(P);
    const bool kok = kk < klim;                       // K % 8 == 0: a vector is entirely inside or outside
    if (!w.enabled) {
#pragma unroll
      for (int i = IB; i < IE; ++i) {
        regs[i] = *reinterpret_cast<const uint4*>(kok ? ptr[i] : Ph);
        okbits = (okbits & ~(1u << i)) | ((kok ? 1u : 0u) << i);
        ptr[i] += BK;
      }
      if constexpr (IE == NV) kk += BK;
    } else {
      const int64_t tapoff = (int64_t)tap * w.N;
#pragma unroll
      for (int i = IB; i < IE; ++i) {
        const int32_t t_in = wr[i].t0 + tap;          // INVALID + tap stays hugely negative
        const bool ok = kok && t_in >= 0 && t_in < w.Lin;
        const int64_t row = wr[i].srow + tapoff;
        regs[i] = *reinterpret_cast<const uint4*>(ok ? Ph + row * ld + c : Ph);
        okbits = (okbits & ~(1u << i)) | ((ok ? 1u : 0u) << i);
      }
      if constexpr (IE == NV) {
        kk += BK;
        c += BK;
        while (c >= w.Cw) { c -= w.Cw; ++tap; }
      }
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_steady(int64_t, const DropCtx&) {
    if constexpr (IB >= IE) return;
    if constexpr (IB == 0) okbits = ~0u;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      regs[i] = *reinterpret_cast<const uint4*>(ptr[i]);
      ptr[i] += BK;
    }
    if constexpr (IE == NV) kk += BK;
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_part(__bf16* lds, const DropCtx&) {
    if constexpr (IB >= IE) return;
    const int cv = (threadIdx.x % VPR) * 8;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      const uint4 v = ((okbits >> i) & 1u) ? regs[i] : make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(lds + (r0 + i * RSTEP) * LDH + cv) = v;
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_steady(__bf16* lds, const DropCtx&) {
    if constexpr (IB >= IE) return;
    const int cv = (threadIdx.x % VPR) * 8;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) *reinterpret_cast<uint4*>(lds + (r0 + i * RSTEP) * LDH + cv) = regs[i];
  }
};

// bf16 source whose contiguous dimension is the tile-row index ([k][m] / [k][n]: the weight-gradient forms -- dy
// written as bf16 by gn_gelu_bwd is the A operand of the conv dW GEMMs, gelu(GroupNorm(.)) the windowed B operand of the
// 1x1 conv's dW).  One item = k rows (2kp, 2kp+1) x 8 consecutive tile rows: two 16-byte loads, eight ds_write_b32 of
// (k, k+1) pairs built with v_perm_b32.  Groups of 8 tile rows are entirely in or out (host: M resp. N % 8 == 0) and
// never straddle a tap (Cw % 8 == 0).
template <int ROWS>
struct TStager16 {
  static constexpr int QPR = ROWS / 8;                // row octets per k pair
  static constexpr int NI = (BK / 2) * QPR / NTH;     // 2 (A, 256 rows) or 1 (B, 128 rows)
  static constexpr int KSTEP = NTH / QPR;             // k pairs between a thread's items
  static constexpr int NITEMS = NI;
  uint4 lo[NI], hi[NI];
  const __bf16* ptr[NI];
  bool inner_ok;
  uint32_t okbits;                                    // bit 2i: row 2kp valid, bit 2i+1: row 2kp+1 valid
  int32_t tap, c, inner;
  int32_t rn[NI], rt[NI], rb[NI];                     // window view: (n, t_out, b) of view row k0 + 2*(kp0 + i*KSTEP)

  __device__ __forceinline__ void init(const float* __restrict__ P, const TecmWin& w, int64_t ld, int32_t kbeg,
                                       int64_t fixed0, int64_t fixed_lim, const DropCtx&) {
    const __bf16* Ph = reinterpret_cast<const __bf16*>(P);
    const int q8 = (threadIdx.x % QPR) * 8;
    const int kp0 = threadIdx.x / QPR;
    inner = (int32_t)(fixed0 + q8);
    inner_ok = inner < fixed_lim;
    if (!inner_ok) inner = 0;                         // clamped octet (valid address, never stored); WIN: tap/c of column 0
    okbits = 0;
    tap = 0;
    c = 0;
    if (!w.enabled) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int64_t krow = (int64_t)kbeg + 2 * (kp0 + i * KSTEP);
        ptr[i] = Ph + krow * ld + inner;
      }
    } else {
      tap = inner / w.Cw;
      c = inner - tap * w.Cw;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint32_t row = (uint32_t)(kbeg + 2 * (kp0 + i * KSTEP));
        const uint32_t q = row / (uint32_t)w.N;
        rn[i] = (int32_t)(row - q * (uint32_t)w.N);
        rb[i] = (int32_t)(q / (uint32_t)w.Lout);
        rt[i] = (int32_t)(q - (uint32_t)rb[i] * (uint32_t)w.Lout);
      }
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_part(const float* __restrict__ P, const TecmWin& w, int64_t ld, int32_t k0,
                                            int32_t klim, const DropCtx&) {
    if constexpr (IB >= IE) return;
    const __bf16* Ph = reinterpret_cast<const __bf16*>(P);
    const int kp0 = threadIdx.x / QPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      const int32_t krow = k0 + 2 * (kp0 + i * KSTEP);
      bool ok0, ok1;
      if (!w.enabled) {
        ok0 = inner_ok && krow < klim;
        ok1 = inner_ok && krow + 1 < klim;
        lo[i] = *reinterpret_cast<const uint4*>(ok0 ? ptr[i] : Ph);
        hi[i] = *reinterpret_cast<const uint4*>(ok1 ? ptr[i] + ld : Ph);
        ptr[i] += (int64_t)BK * ld;
      } else {
        int32_t n1 = rn[i] + 1, t1 = rt[i], b1 = rb[i];                    // row krow + 1
        if (n1 >= w.N) { n1 = 0; if (++t1 >= w.Lout) { t1 = 0; ++b1; } }
        const int32_t ta = rt[i] * w.stride_t - w.pad + tap, tb = t1 * w.stride_t - w.pad + tap;
        ok0 = inner_ok && krow < klim && ta >= 0 && ta < w.Lin;
        ok1 = inner_ok && krow + 1 < klim && tb >= 0 && tb < w.Lin;
        const int64_t rowa = ((int64_t)rb[i] * w.Lin + ta) * (int64_t)w.N + rn[i];
        const int64_t rowb = ((int64_t)b1 * w.Lin + tb) * (int64_t)w.N + n1;
        lo[i] = *reinterpret_cast<const uint4*>(ok0 ? Ph + rowa * ld + c : Ph);
        hi[i] = *reinterpret_cast<const uint4*>(ok1 ? Ph + rowb * ld + c : Ph);
        rn[i] += BK;                                                        // next K-tile
        while (rn[i] >= w.N) {
          rn[i] -= w.N;
          if (++rt[i] >= w.Lout) { rt[i] = 0; ++rb[i]; }
        }
      }
      okbits = (okbits & ~(3u << (2 * i))) | ((ok0 ? 1u : 0u) << (2 * i)) | ((ok1 ? 1u : 0u) << (2 * i + 1));
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_steady(int64_t ld, const DropCtx&) {
    if constexpr (IB >= IE) return;
    if constexpr (IB == 0) okbits = ~0u;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      lo[i] = *reinterpret_cast<const uint4*>(ptr[i]);
      hi[i] = *reinterpret_cast<const uint4*>(ptr[i] + ld);
      ptr[i] += (int64_t)BK * ld;
    }
  }
  __device__ __forceinline__ void put(__bf16* lds, int i, const uint4& a, const uint4& b) {
    const int q8 = (threadIdx.x % QPR) * 8;
    const int kp0 = threadIdx.x / QPR;
    const int kcol = 2 * (kp0 + i * KSTEP);
    uint32_t* base = reinterpret_cast<uint32_t*>(lds + q8 * LDH + kcol);
    const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // tile rows 2j and 2j+1 of the octet: (k, k+1) = (low half of a, low half of b) / (high halves)
      base[(2 * j) * (LDH / 2)] = __builtin_amdgcn_perm(bv[j], av[j], 0x05040100u);
      base[(2 * j + 1) * (LDH / 2)] = __builtin_amdgcn_perm(bv[j], av[j], 0x07060302u);
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_steady(__bf16* lds, const DropCtx&) {
    if constexpr (IB >= IE) return;
#pragma unroll
    for (int i = IB; i < IE; ++i) put(lds, i, lo[i], hi[i]);
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_part(__bf16* lds, const DropCtx&) {
    if constexpr (IB >= IE) return;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = IB; i < IE; ++i)
      put(lds, i, ((okbits >> (2 * i)) & 1u) ? lo[i] : z, ((okbits >> (2 * i + 1)) & 1u) ? hi[i] : z);
  }
};

// Plain (no window) bf16 [k][row] source, transposed in registers: one item = 8 k rows x 8 consecutive tile rows --
// eight 16-byte loads (a wave covers 8 k rows x 128 contiguous bytes per instruction), an 8 x 8 transpose of 16-bit
// elements with v_perm_b32, eight ds_write_b128 (one per tile row: lanes 0..7 of a group write the 128 contiguous
// bytes of a row, conflict-free).  TStager16's (k, k+1)-pair writes put all 32 lanes of a ds_write_b32 on ONE bank
// with this LDS pitch (row * 36 words, rows a multiple of 8 apart): the conv dW GEMMs ran 45 % slower with it than
// from fp32.  The A operand of those GEMMs (dy, [k][m]) takes this stager.
template <int ROWS>
struct TStager8x8 {
  static constexpr int OCT = ROWS / 8;                // row octets
  static constexpr int KCH = BK / 8;                  // k chunks of 8
  static_assert(OCT * KCH <= NTH, "one item per thread at most");
  static constexpr int NITEMS = 8;                    // the 8 k rows of the item: the pipeline loads them in quarters
  uint4 L[8];
  const __bf16* ptr;
  bool inner_ok, active;
  uint32_t okbits;
  int32_t krow0;

  __device__ __forceinline__ void init(const float* __restrict__ P, const TecmWin&, int64_t ld, int32_t kbeg,
                                       int64_t fixed0, int64_t fixed_lim, const DropCtx&) {
    const __bf16* Ph = reinterpret_cast<const __bf16*>(P);
    const int kc = threadIdx.x % KCH, oct = threadIdx.x / KCH;
    active = oct < OCT;
    const int64_t inner = fixed0 + oct * 8;
    inner_ok = active && inner < fixed_lim;
    krow0 = kbeg + kc * 8;
    ptr = Ph + (int64_t)krow0 * ld + (inner_ok ? inner : 0);
    okbits = 0;
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_part(const float* __restrict__, const TecmWin&, int64_t ld, int32_t, int32_t klim,
                                            const DropCtx&) {
    if constexpr (IB >= IE) return;
#pragma unroll
    for (int j = IB; j < IE; ++j) {
      const bool ok = inner_ok && (krow0 + j) < klim;
      L[j] = ok ? *reinterpret_cast<const uint4*>(ptr + (int64_t)j * ld) : make_uint4(0u, 0u, 0u, 0u);
      okbits = (okbits & ~(1u << j)) | ((ok ? 1u : 0u) << j);
    }
    if constexpr (IE == 8) { ptr += (int64_t)BK * ld; krow0 += BK; }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_steady(int64_t ld, const DropCtx&) {
    if constexpr (IB >= IE) return;
#pragma unroll
    for (int j = IB; j < IE; ++j) L[j] = inner_ok ? *reinterpret_cast<const uint4*>(ptr + (int64_t)j * ld) : make_uint4(0u, 0u, 0u, 0u);
    if constexpr (IE == 8) { ptr += (int64_t)BK * ld; krow0 += BK; okbits = inner_ok ? 0xffu : 0u; }
  }
  // the transpose needs all eight k rows: the whole item is stored with the FIRST part (the pipeline stores part s of
  // tile t+1 before it loads part s of tile t+2, so every register still holds tile t+1 then)
  template <int IB, int IE>
  __device__ __forceinline__ void store_part(__bf16* lds, const DropCtx&) {
    if constexpr (IB != 0 || IB >= IE) return;
    if (!active) return;
    const int kc = threadIdx.x % KCH, oct = threadIdx.x / KCH;
    uint32_t w[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w[j][0] = L[j].x; w[j][1] = L[j].y; w[j][2] = L[j].z; w[j][3] = L[j].w; }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint4 o;
      const uint32_t sel = (r & 1) ? 0x07060302u : 0x05040100u;   // high / low halves of (k+1 word, k word)
      o.x = __builtin_amdgcn_perm(w[1][r >> 1], w[0][r >> 1], sel);
      o.y = __builtin_amdgcn_perm(w[3][r >> 1], w[2][r >> 1], sel);
      o.z = __builtin_amdgcn_perm(w[5][r >> 1], w[4][r >> 1], sel);
      o.w = __builtin_amdgcn_perm(w[7][r >> 1], w[6][r >> 1], sel);
      *reinterpret_cast<uint4*>(lds + (oct * 8 + r) * LDH + kc * 8) = o;
    }
  }
  template <int IB, int IE>
  __device__ __forceinline__ void store_steady(__bf16* lds, const DropCtx& dc) { store_part<IB, IE>(lds, dc); }
};

template <bool TRANS, int ROWS, bool WIN, bool DROP>
struct StagerSel {
  using type = DStager<ROWS, WIN, DROP>;
};
template <int ROWS, bool WIN, bool DROP>
struct StagerSel<true, ROWS, WIN, DROP> {
  using type = TStager<ROWS, WIN, DROP>;
};

// Epilogue shared by the bf16 kernels: identical to the fp32 kernel's, 32-row slabs per wave through LDS.
// HALF: the fast path walks 16-row half slabs (half the registers of the prefetched input stream: the 128-register
// two-blocks-per-CU kernel); same element arithmetic, same bits.
template <int MT, int NT, int WTM, int WTN, bool NTS = false, bool HALF = false>      // NTS: non-temporal stores in the fast path
__device__ __forceinline__ void block_epilogue16(const TecmGemm& g, f32x16 (&acc)[MT][NT], unsigned char* smem_raw, int wave,
                                                 int lane, int wm, int wn, int64_t m0, int64_t n0) {
  constexpr int STG_LD = WTN + 4;
  const int r = lane & 31, h = lane >> 5;
  const DropCtx odc = make_drop(g.out_drop);
  const bool split = gridDim.z > 1;
  float* stg = reinterpret_cast<float*>(smem_raw) + wave * (32 * STG_LD);
  auto stage_slab = [&](auto ic) {                     // 32 accumulator rows of this wave -> its private staging rows
    constexpr int i = decltype(ic)::value;
    static_for<16>([&](auto ec) {
      constexpr int e = decltype(ec)::value;
      static_for<NT>([&](auto jc) {
        constexpr int jn = decltype(jc)::value;
        stg[((e & 3) + 8 * (e >> 2) + 4 * h) * STG_LD + jn * 32 + r] = acc[i][jn][e];
      });
    });
  };
  const int fmode = tecm_gemm::epi_fast_mode(g);
  if (fmode >= 0) {
    // straight-line form (gemm_impl.h): no barriers -- the staging rows are private to the wave, and a
    // __syncthreads would drain every outstanding store -- and the optional input stream runs one slab ahead
    constexpr int LPR = WTN / 4, RPI = 64 / LPR;
    const int lcol = (lane % LPR) * 4, lrow = lane / LPR;
    const EpiCol ecol = epi_col(g, n0 + wn * WTN + lcol);
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + (ecol.ok ? ecol.n : 0));   // (lanes outside the matrix: column 0, never stored)
    if constexpr (HALF) {
      auto stage_half = [&](auto sc) {                   // half slab hs of 32-row slab i: accumulator registers 8hs .. 8hs+7
        constexpr int i = decltype(sc)::value / 2, hs = decltype(sc)::value % 2;
        static_for<8>([&](auto ec) {
          constexpr int e = 8 * hs + decltype(ec)::value;
          static_for<NT>([&](auto jc) {
            constexpr int jn = decltype(jc)::value;
            stg[((e & 3) + 8 * ((e >> 2) & 1) + 4 * h) * STG_LD + jn * 32 + r] = acc[i][jn][e];
          });
        });
      };
      tecm_gemm::epi_fast_dispatch<2 * MT, 16 / RPI, RPI, STG_LD, NTS>(fmode, g, odc, stg, lrow, lcol, m0 + wm * WTM, ecol,
                                                                   bias4, stage_half);
    } else {
      tecm_gemm::epi_fast_dispatch<MT, 32 / RPI, RPI, STG_LD, NTS>(fmode, g, odc, stg, lrow, lcol, m0 + wm * WTM, ecol, bias4,
                                                               stage_slab);
    }
    return;
  }
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
      const int lcol = lane % WTN, lrow = lane / WTN;     // WTN = 64: one row per iteration
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

template <int ALAY, int BLAY, bool WIN, bool DROP, int ADT = 0, int BDT = 0>      // ADT / BDT: 1 = the operand is bf16 in HBM
__global__ __launch_bounds__(NTH, 2) void gemm_bf16_kernel(const TecmGemm g, int tiles_m, int tiles_n, int k_chunk) {
  static_assert((ADT == 0 && BDT == 0) || !DROP, "bf16 sources: no prologue dropout");
  constexpr int WN = 2, WM = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;          // 64 x 64 per wave
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr bool ATR = ALAY == TECM_A_KM, BTR = BLAY == TECM_B_KN;
  using AStager16 = std::conditional_t<ATR, TStager8x8<BM>, std::conditional_t<WIN, HStagerW<BM>, HStager<BM>>>;   // a_win needs MK: [k][m] A is plain
  using BStager16 = std::conditional_t<BTR, TStager16<BN>, std::conditional_t<WIN, HStagerW<BN>, HStager<BN>>>;
  using AStager = std::conditional_t<ADT == 1, AStager16, typename StagerSel<ATR, BM, WIN, DROP>::type>;
  using BStager = std::conditional_t<BDT == 1, BStager16, typename StagerSel<BTR, BN, WIN, DROP>::type>;
  constexpr int A_ELEMS = BM * LDH, B_ELEMS = BN * LDH, TILE_ELEMS = A_ELEMS + B_ELEMS;   // bf16 elements
  constexpr int STG_LD = WTN + 4;
  constexpr int STG_BYTES = 8 * 32 * STG_LD * 4;       // one 32-row slab per wave, fp32
  constexpr int SMEM_BYTES = 2 * TILE_ELEMS * 2 > STG_BYTES ? 2 * TILE_ELEMS * 2 : STG_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);

  // XCD-aware bijective block -> tile map with 8-m-tile groups (see gemm_impl.h)
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
  const DropCtx adc = make_drop(g.a_drop), bdc = make_drop(g.b_drop);

  AStager sa;
  BStager sb;
  if constexpr (ATR) sa.init(g.A, g.a_win, g.lda, kbeg, m0, g.M, adc);
  else sa.init(g.A, g.a_win, g.lda, m0, g.M, kbeg, adc);
  if constexpr (BTR) sb.init(g.B, g.b_win, g.ldb, kbeg, n0, g.N, bdc);
  else sb.init(g.B, g.b_win, g.ldb, n0, g.N, kbeg, bdc);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto a_store = [&](auto ib, auto ie, __bf16* dst) {
    sa.template store_part<decltype(ib)::value, decltype(ie)::value>(dst, adc);
  };
  auto b_store = [&](auto ib, auto ie, __bf16* dst) {
    sb.template store_part<decltype(ib)::value, decltype(ie)::value>(dst, bdc);
  };
  constexpr int ANV = AStager::NITEMS;
  constexpr int BNV = BStager::NITEMS;
  using I0 = std::integral_constant<int, 0>;

  // prologue: tile 0 -> LDS buffer 0, tile 1 -> registers (in flight)
  sa.template load_part<0, ANV>(g.A, g.a_win, g.lda, kbeg, kend, adc);
  sb.template load_part<0, BNV>(g.B, g.b_win, g.ldb, kbeg, kend, bdc);
  a_store(I0{}, std::integral_constant<int, ANV>{}, smem);
  b_store(I0{}, std::integral_constant<int, BNV>{}, smem + A_ELEMS);
  if (kbeg + BK < kend) {
    sa.template load_part<0, ANV>(g.A, g.a_win, g.lda, kbeg + BK, kend, adc);
    sb.template load_part<0, BNV>(g.B, g.b_win, g.ldb, kbeg + BK, kend, bdc);
  }
  __syncthreads();

  int cur = 0;
  auto tile = [&](int32_t k0, auto fullc) {
    constexpr bool FULL = decltype(fullc)::value;         // K-tiles t+1 and t+2 lie entirely inside [kbeg, kend)
    const __bf16* Ac = smem + cur * TILE_ELEMS;
    const __bf16* Bc = Ac + A_ELEMS;
    __bf16* An = smem + (cur ^ 1) * TILE_ELEMS;
    static_for<4>([&](auto sc) {
      constexpr int s = decltype(sc)::value;              // k-step of 16 inside the 64-deep tile
      bf16x8 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(Ac + (wm * WTM + i * 32 + r) * LDH + 16 * s + 8 * h);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        bf[i] = *reinterpret_cast<const bf16x8*>(Bc + (wn * WTN + i * 32 + r) * LDH + 16 * s + 8 * h);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[jn], acc[i][jn], 0, 0, 0);
      // a quarter of the staging work per k-step (see gemm_impl.h)
      constexpr int AB = (ANV * s) / 4, AE = (ANV * (s + 1)) / 4;
      constexpr int BB = (BNV * s) / 4, BE = (BNV * (s + 1)) / 4;
      if constexpr (FULL && !WIN) {
        // mask-free staging: the conversion pass is VALU-bound (4 cvt + pack per float4), every select saved counts
        sa.template store_steady<AB, AE>(An, adc);
        sb.template store_steady<BB, BE>(An + A_ELEMS, bdc);
        sa.template load_steady<AB, AE>(g.lda, adc);
        sb.template load_steady<BB, BE>(g.ldb, bdc);
      } else {
        if (FULL || k0 + BK < kend) {
          a_store(std::integral_constant<int, AB>{}, std::integral_constant<int, AE>{}, An);
          b_store(std::integral_constant<int, BB>{}, std::integral_constant<int, BE>{}, An + A_ELEMS);
        }
        if (FULL || k0 + 2 * BK < kend) {
          sa.template load_part<AB, AE>(g.A, g.a_win, g.lda, k0 + 2 * BK, kend, adc);
          sb.template load_part<BB, BE>(g.B, g.b_win, g.ldb, k0 + 2 * BK, kend, bdc);
        }
      }
    });
    __syncthreads();
    cur ^= 1;
  };
  int32_t k0 = kbeg;
  for (; k0 + 3 * BK <= kend; k0 += BK) tile(k0, std::true_type{});
  for (; k0 < kend; k0 += BK) tile(k0, std::false_type{});

  // non-temporal stores (see gemm_bf16_dma.hip): +1 % on the whole bf16 step on top of the DMA kernel's own gain
  block_epilogue16<MT, NT, WTM, WTN, true>(g, acc, smem_raw, wave, lane, wm, wn, m0, n0);
}

template <int ALAY, int BLAY, bool WIN, bool DROP, int ADT = 0, int BDT = 0>
int launch(const TecmGemm& g, hipStream_t st) {
  const int tiles_m = (int)((g.M + BM - 1) / BM);
  const int tiles_n = (int)((g.N + BN - 1) / BN);
  int splits = g.split_k > 1 ? g.split_k : 1;
  int k_chunk = (int)(((g.K + splits - 1) / splits + BK - 1) / BK) * BK;
  splits = (int)((g.K + k_chunk - 1) / k_chunk);
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)splits);
  hipLaunchKernelGGL((gemm_bf16_kernel<ALAY, BLAY, WIN, DROP, ADT, BDT>), grid, dim3(NTH), 0, st, g, tiles_m, tiles_n, k_chunk);
  TECM_CHECK_LAUNCH("tecm_gemm_bf16");
  return splits;
}

template <int ALAY, int BLAY>
int dispatch(const TecmGemm& g, bool win, bool drop, hipStream_t st) {
  if (!win && !drop) return launch<ALAY, BLAY, false, false>(g, st);
  if (win && !drop) return launch<ALAY, BLAY, true, false>(g, st);
  if (!win && drop) return launch<ALAY, BLAY, false, true>(g, st);
  return launch<ALAY, BLAY, true, true>(g, st);
}

}  // namespace tecm_gemm16

int tecm_gemm16_dispatch_mk_nk(const TecmGemm& g, bool win, bool drop, hipStream_t st);
int tecm_gemm16_dispatch_mk_kn(const TecmGemm& g, bool win, bool drop, hipStream_t st);
int tecm_gemm16_dispatch_km_kn(const TecmGemm& g, bool win, bool drop, hipStream_t st);
