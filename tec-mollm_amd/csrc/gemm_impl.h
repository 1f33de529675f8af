// fp32 GEMM on the exact-f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32) -- kernel template.
//
//   C[M,N] = epilogue(alpha * A_view[M,K] . B_view[K,N])
//
// One kernel family serves every dense contraction of the TEC-MoLLM path (see include/tecmollm.h):
// operands are addressed through "row views" so that Conv1d-over-time, the strided 1x1 conv,
// latent patching and the head's flatten never materialise an im2col / permuted copy.
//
// Tile BM x BN x BK = 128 x {128, 64, 32} x 32 (64 x BN for the M <= 64 weight-gradient GEMMs).
//   BN = 128: 512 threads = 8 waves as 2(m) x 4(n), each wave 64 x 32 = two 32x32 MFMA tiles (32 accumulator VGPRs),
//             108 VGPRs, 72 KiB of LDS -> two blocks (16 waves) per CU
//   BN <= 64: 256 threads = 4 waves
// LDS tiles keep the operand's own orientation:
//   [row][k] tiles (MK / NK): leading dim 36 floats -> ds_read_b128 of 4 consecutive k is
//       conflict-free (16 lanes x 4 banks, row stride 36 = 4 mod 32 hits 16 distinct slots);
//   [k][row] tiles (KM / KN): leading dim rows+4 -> ds_read_b32, lanes 0..31 consecutive banks.
// The k index fed to MFMA step (q,j) by lane half h is k = 8q + 4h + j for BOTH operands, which is
// what makes the b128 read legal (any bijection of k works as long as A and B agree).
// Pipeline: two LDS buffers, one barrier per K-tile; K-tile t+1 is parked in the idle buffer and t+2 fetched
// into registers while the MFMAs of tile t run (the prologue sends K-tiles 0 and 1 out together).
//
// The f32 MFMA is slow enough (64 cycles per 32x32x2) that the loop is MFMA-bound only if the loader
// costs a few dozen VALU instructions per K-tile: all per-row address state is resolved before the
// K loop and advanced incrementally; the window view and the dropout prologue are compile-time
// variants (WIN / DROP) so the plain variant carries none of their code, and in its steady state the plain
// variant stages with no masks at all (rows / columns pre-clamped, see Stager::load_steady).
// Measured limits and the experiments that did not pay: DESIGN.md section 4, tools/experiments/.
#pragma once
#include "common.h"
#ifndef TECM_BIG_THREADS
#define TECM_BIG_THREADS 512
#endif
#include <utility>
#ifndef TECM_ABLATE
#define TECM_ABLATE 0      // experiments only: 1 no staging, 2 no LDS reads either, 3 loads only, 4 stores only
#endif
#ifndef TECM_STAGE_Q
#define TECM_STAGE_Q 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// TecmGemm::io_bf16 as the kernels see it: the caller's bits 0..2 (TECM_IO_*) plus the host-computed epilogue flag
#define TECM_P0_VEC4 0x100      /* every pointer / leading dimension the epilogue touches is 16-byte friendly */

namespace tecm_gemm {

// compile-time loop: indices reach the body as integral constants, so register arrays indexed with
// them can never be demoted to scratch (a plain `#pragma unroll` is only a request).
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr int NTHREADS_MAX = 512;
__host__ __device__ constexpr int threads_for(int bn) { return bn >= 128 ? TECM_BIG_THREADS : 256; }

struct DropCtx {
  uint64_t seed;
  int64_t ld;
  uint32_t thresh;
  float inv;
  const uint64_t* sdev;        // host-made contexts only: the kernel adds *sdev to seed when it starts (tecm_seed_now)
};
inline DropCtx make_drop_host(const TecmDrop& d) {
  DropCtx c;
  c.seed = d.seed;
  c.sdev = d.seed_dev;
  c.ld = d.ld;
  c.thresh = d.p > 0.f ? tecm_drop_thresh(d.p) : 0u;
  c.inv = d.p > 0.f ? 1.0f / (1.0f - d.p) : 1.0f;
  return c;
}
__device__ __forceinline__ DropCtx make_drop(const TecmDrop& d) {
  DropCtx c;
  c.seed = tecm_seed_now(d.seed, d.seed_dev);
  c.sdev = nullptr;
  c.ld = d.ld;
  c.thresh = d.p > 0.f ? tecm_drop_thresh(d.p) : 0u;
  c.inv = d.p > 0.f ? 1.0f / (1.0f - d.p) : 1.0f;
  return c;
}

// Branch-free guarded load: out-of-range lanes read `safe` (always a valid address of the same operand); the
// caller zeroes them later, at LDS-store time (tile_finish), so that nothing consumes the loaded registers --
// and forces an s_waitcnt vmcnt -- right after the load has been issued.  No exec-mask branches => the whole
// K-tile body stays ONE scheduling region, which the sched_group_barrier interleave in the main loop needs.
template <int VEC>
__device__ __forceinline__ void gload(const float* __restrict__ p, const float* __restrict__ safe, bool ok,
                                      float (&out)[VEC]) {
  const float* q = ok ? p : safe;
  if constexpr (VEC == 4) {
    const float4 v = *reinterpret_cast<const float4*>(q);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else if constexpr (VEC == 2) {
    const float2 v = *reinterpret_cast<const float2*>(q);
    out[0] = v.x; out[1] = v.y;
  } else {
    out[0] = q[0];
  }
}

template <int VEC>
__device__ __forceinline__ void apply_drop(const DropCtx& dc, int64_t didx, float (&v)[VEC]) {
  if (dc.thresh) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] *= tecm_drop_mult(dc.seed, (uint64_t)(didx + e), dc.thresh, dc.inv);
  }
}

template <int VEC>
__device__ __forceinline__ void lds_store(float* dst, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
  } else {
    dst[0] = v[0];
  }
}

// 32-bit row decomposition for the window view (host guarantees rows < 2^31)
struct WinRow {
  int64_t srow;   // source row of tap 0
  int32_t t0;     // t_out*stride_t - pad, TECM_ROW_INVALID if the row is out of range
};
__device__ __forceinline__ WinRow win_row(const TecmWin& w, int64_t m, int64_t rows) {
  WinRow r;
  if (m >= rows) {
    r.srow = 0;
    r.t0 = TECM_ROW_INVALID;
    return r;
  }
  const uint32_t mm = (uint32_t)m;
  const uint32_t q = mm / (uint32_t)w.N;
  const int32_t n = (int32_t)(mm - q * (uint32_t)w.N);
  const uint32_t bq = q / (uint32_t)w.Lout;
  const int32_t t_out = (int32_t)(q - bq * (uint32_t)w.Lout);
  r.t0 = t_out * w.stride_t - w.pad;
  r.srow = ((int64_t)bq * w.Lin + r.t0) * (int64_t)w.N + n;
  return r;
}

// Tile stager.  ROWK=false: tile rows are the (fixed) M/N index, inner index is k  ([row][k], LD=36)
//               ROWK=true : tile rows are k, inner index is the (fixed) M/N index ([k][row], LD=ROWS+4)
template <bool ROWK, int ROWS, int VEC, bool WIN, bool DROP, int NTHREADS>
struct Stager {
  static constexpr int R = ROWK ? BK : ROWS;
  static constexpr int CI = ROWK ? ROWS : BK;
  static constexpr int LD = ROWK ? ROWS + 4 : LDK;
  static constexpr int VPR = CI / VEC;
  static constexpr int NV = (R * VPR) / NTHREADS;
  static constexpr int RSTEP = NTHREADS / VPR;
  static_assert((R * VPR) % NTHREADS == 0 && NTHREADS % VPR == 0, "tile/thread mapping");

  // One K-tile's worth of this thread's vectors in flight.  Kept apart from the address state so that the prologue
  // can have K-tiles 0 and 1 in flight at the same time (two Tiles, one round trip instead of two).
  struct Tile {
    float regs[NV][VEC];
    uint32_t okbits = 0;                  // bit i: vector i is in range (else stored as 0)
    int64_t dsave[DROP ? NV : 1];         // dropout index of vector i
  };
  // --- plain view state
  const float* ptr[NV];                   // address of this thread's vector i in the NEXT tile to load
  uint32_t rowok;                         // bit i: fixed row i in range (ROWK=false) / unused
  bool inner_ok;                          // ROWK=true: fixed inner index in range
  int64_t didx[DROP ? NV : 1];            // dropout index of the vector's first element
  // --- window view state
  WinRow wr[(WIN && !ROWK) ? NV : 1];     // fixed rows (A-window)
  // ROWK window (weight-gradient form: the view's rows are the reduction index): (n, t_out, b) of the row each
  // vector reads next, advanced by BK rows per K-tile without the two integer divisions of win_row()
  int32_t rn[(WIN && ROWK) ? NV : 1], rt[(WIN && ROWK) ? NV : 1], rb[(WIN && ROWK) ? NV : 1];
  int32_t tap, c;                         // per-thread tap / channel of the inner index (kk = tap*Cw + c)
  int32_t kk;                             // ROWK=false: current inner index k0 + cv

  __device__ __forceinline__ void init(const float* __restrict__ P, const TecmWin& w, int64_t ld, int64_t row0,
                                       int64_t rows_total, int32_t kbeg, int64_t fixed0, int64_t fixed_lim,
                                       const DropCtx& dc) {
    const int cv = (threadIdx.x % VPR) * VEC;
    const int r0 = threadIdx.x / VPR;
    rowok = 0;
    inner_ok = true;
    tap = 0;
    c = 0;
    kk = kbeg + cv;
    const bool wen = WIN && w.enabled;       // WIN variants still serve plain operands at run time
    if (!wen) {
      if constexpr (!ROWK) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          int64_t row = row0 + r0 + i * RSTEP;
          if (row < rows_total) rowok |= 1u << i;
          else row = rows_total - 1;          // clamped: always a valid address, the tile row it feeds is never stored
          ptr[i] = P + row * ld + kbeg + cv;
          if constexpr (DROP) didx[i] = row * dc.ld + kbeg + cv;
        }
      } else {
        inner_ok = (fixed0 + cv) < fixed_lim;
        const int64_t inner = inner_ok ? fixed0 + cv : 0;      // clamped likewise (columns beyond the limit)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int64_t row = (int64_t)kbeg + r0 + i * RSTEP;
          ptr[i] = P + row * ld + inner;
          if constexpr (DROP) didx[i] = row * dc.ld + inner;
        }
      }
    } else if constexpr (WIN) {
      if constexpr (!ROWK) {
#pragma unroll
        for (int i = 0; i < NV; ++i) wr[i] = win_row(w, row0 + r0 + i * RSTEP, rows_total);
        tap = kk / w.Cw;
        c = kk - tap * w.Cw;
      } else {
        const int32_t inner = (int32_t)(fixed0 + cv);
        inner_ok = inner < fixed_lim;
        tap = inner / w.Cw;
        c = inner - tap * w.Cw;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const uint32_t row = (uint32_t)(kbeg + r0 + i * RSTEP);
          const uint32_t q = row / (uint32_t)w.N;
          rn[i] = (int32_t)(row - q * (uint32_t)w.N);
          rb[i] = (int32_t)(q / (uint32_t)w.Lout);
          rt[i] = (int32_t)(q - (uint32_t)rb[i] * (uint32_t)w.Lout);
        }
      }
    }
  }

  // Load this thread's vectors [IB, IE) of the tile starting at k0; the call that covers the last vector
  // (IE == NV) advances the per-tile state.  load() = all vectors.
  __device__ __forceinline__ void load(Tile& tl, const float* __restrict__ P, const TecmWin& w, int64_t ld, int32_t k0,
                                       int32_t klim, const DropCtx& dc) {
    load_part<0, NV>(tl, P, w, ld, k0, klim, dc);
  }
  template <int IB, int IE>
  __device__ __forceinline__ void load_part(Tile& tl, const float* __restrict__ P, const TecmWin& w, int64_t ld, int32_t k0,
                                            int32_t klim, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    const int r0 = threadIdx.x / VPR;
    const bool wen = WIN && w.enabled;
    constexpr bool LAST = IE == NV;
    if (!wen) {
      if constexpr (!ROWK) {
        const bool kok = kk < klim;
#pragma unroll
        for (int i = IB; i < IE; ++i) {
          const bool ok = kok && ((rowok >> i) & 1u);
          gload<VEC>(ptr[i], P, ok, tl.regs[i]);
          tl.okbits = (tl.okbits & ~(1u << i)) | ((ok ? 1u : 0u) << i);
          if constexpr (DROP) { tl.dsave[i] = didx[i]; didx[i] += BK; }
          ptr[i] += BK;
        }
        if constexpr (LAST) kk += BK;
      } else {
#pragma unroll
        for (int i = IB; i < IE; ++i) {
          const bool ok = inner_ok && (k0 + r0 + i * RSTEP) < klim;
          gload<VEC>(ptr[i], P, ok, tl.regs[i]);
          tl.okbits = (tl.okbits & ~(1u << i)) | ((ok ? 1u : 0u) << i);
          if constexpr (DROP) { tl.dsave[i] = didx[i]; didx[i] += (int64_t)BK * dc.ld; }
          ptr[i] += (int64_t)BK * ld;
        }
      }
    } else if constexpr (WIN) {
      if constexpr (!ROWK) {
        const bool kok = kk < klim;
        const int64_t tapoff = (int64_t)tap * w.N;
#pragma unroll
        for (int i = IB; i < IE; ++i) {
          const int32_t t_in = wr[i].t0 + tap;            // INVALID + tap stays hugely negative
          const bool ok = kok && t_in >= 0 && t_in < w.Lin;
          const int64_t row = wr[i].srow + tapoff;
          gload<VEC>(P + row * ld + c, P, ok, tl.regs[i]);
          tl.okbits = (tl.okbits & ~(1u << i)) | ((ok ? 1u : 0u) << i);
          if constexpr (DROP) tl.dsave[i] = row * dc.ld + c;
        }
        if constexpr (LAST) {
          kk += BK;
          c += BK;
          while (c >= w.Cw) { c -= w.Cw; ++tap; }
        }
      } else {
#pragma unroll
        for (int i = IB; i < IE; ++i) {
          // row k0 + r0 + i*RSTEP of the view = (rb, rt, rn); same result as win_row(), no divisions
          const int32_t t_in = rt[i] * w.stride_t - w.pad + tap;
          const bool ok = inner_ok && (k0 + r0 + i * RSTEP) < klim && t_in >= 0 && t_in < w.Lin;
          const int64_t row = ((int64_t)rb[i] * w.Lin + t_in) * (int64_t)w.N + rn[i];
          gload<VEC>(P + row * ld + c, P, ok, tl.regs[i]);
          tl.okbits = (tl.okbits & ~(1u << i)) | ((ok ? 1u : 0u) << i);
          if constexpr (DROP) tl.dsave[i] = row * dc.ld + c;
          rn[i] += BK;                                       // the next K-tile is BK view rows further
          while (rn[i] >= w.N) {
            rn[i] -= w.N;
            if (++rt[i] >= w.Lout) { rt[i] = 0; ++rb[i]; }
          }
        }
      }
    }
  }

  // Steady state of the plain view (every k of the tile in range, rows/columns pre-clamped in init): no masks,
  // no selects -- one global load and one 64-bit pointer bump per vector.  Out-of-range rows/columns carry
  // clamped (finite or not, irrelevant) data into accumulator rows/columns that the epilogue never stores.
  __device__ __forceinline__ void load_steady(Tile& tl, int64_t ld, const DropCtx& dc) {
    tl.okbits = ~0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      gload<VEC>(ptr[i], ptr[i], true, tl.regs[i]);
      if constexpr (!ROWK) {
        if constexpr (DROP) { tl.dsave[i] = didx[i]; didx[i] += BK; }
        ptr[i] += BK;
      } else {
        if constexpr (DROP) { tl.dsave[i] = didx[i]; didx[i] += (int64_t)BK * dc.ld; }
        ptr[i] += (int64_t)BK * ld;
      }
    }
    if constexpr (!ROWK) kk += BK;
  }
  __device__ __forceinline__ void store_steady(const Tile& tl, float* lds, const DropCtx& dc) {
    const int cv = (threadIdx.x % VPR) * VEC;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if constexpr (DROP) {
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = tl.regs[i][e];
        apply_drop<VEC>(dc, tl.dsave[i], v);
        lds_store<VEC>(lds + (r0 + i * RSTEP) * LD + cv, v);
      } else {
        lds_store<VEC>(lds + (r0 + i * RSTEP) * LD + cv, tl.regs[i]);
      }
    }
  }

  __device__ __forceinline__ void store(const Tile& tl, float* lds, const DropCtx& dc) { store_part<0, NV>(tl, lds, dc); }
  template <int IB, int IE>
  __device__ __forceinline__ void store_part(const Tile& tl, float* lds, const DropCtx& dc) {
    if constexpr (IB >= IE) return;
    const int cv = (threadIdx.x % VPR) * VEC;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = IB; i < IE; ++i) {
      float v[VEC];
      const bool ok = (tl.okbits >> i) & 1u;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = ok ? tl.regs[i][e] : 0.f;
      if constexpr (DROP) apply_drop<VEC>(dc, tl.dsave[i], v);
      lds_store<VEC>(lds + (r0 + i * RSTEP) * LD + cv, v);
    }
  }
};

// ---------------------------------------------------------------------------------------- epilogue
// Row-dependent state (offsets, window decomposition, row bias) is resolved once per accumulator row,
// column-dependent state (bias, window tap) once per accumulator column; the per-element core is a
// handful of instructions behind wave-uniform feature branches.
struct EpiRow {
  int64_t m;
  int64_t crow;        // plain: m*ldc ; window: source row of tap 0 (srow)
  int64_t drow;        // plain: m*drop_ld
  int32_t t0;          // window only
  const float* rb;     // row bias row (or nullptr)
};
struct EpiCol {
  int32_t n;
  int32_t tap, c;      // window only
  float bias;
  bool ok;
};

__device__ __forceinline__ EpiRow epi_row(const TecmGemm& g, const DropCtx& odc, int64_t m) {
  EpiRow r;
  r.m = m;
  r.t0 = 0;
  if (g.c_win.enabled) {
    const WinRow w = win_row(g.c_win, m, g.M);
    r.crow = w.srow;
    r.t0 = w.t0;
    r.drow = 0;
  } else {
    r.crow = m * g.ldc;
    r.drow = m * odc.ld;
  }
  r.rb = g.rowbias ? g.rowbias + (int64_t)(((uint32_t)m / (uint32_t)g.rb_div) % (uint32_t)g.rb_mod) * g.rb_ld
                   : nullptr;
  return r;
}
__device__ __forceinline__ EpiCol epi_col(const TecmGemm& g, int64_t n) {
  EpiCol c;
  c.n = (int32_t)n;
  c.ok = n < g.N;
  c.bias = (g.bias && c.ok) ? g.bias[n] : 0.f;
  c.tap = 0;
  c.c = 0;
  if (g.c_win.enabled) {
    c.tap = c.n / g.c_win.Cw;
    c.c = c.n - c.tap * g.c_win.Cw;
  }
  return c;
}
__device__ __forceinline__ void epi_elem(const TecmGemm& g, const DropCtx& odc, const EpiRow& r, const EpiCol& c,
                                         float v) {
  const int32_t n = c.n;
  v = v * g.alpha + c.bias;
  if (r.rb) v += r.rb[n];
  if (g.preact) g.preact[r.m * g.ldp + n] = v;
  if (g.dact_src)
    v *= dgelu_tanh(g.dact_src[r.m * g.ldd + n]);          // backward through tanh-GELU
  else if (g.act)
    v = gelu_tanh(v);
  int64_t off, didx;
  if (g.c_win.enabled) {
    const int32_t t_in = r.t0 + c.tap;
    if (t_in < 0 || t_in >= g.c_win.Lin) return;
    const int64_t row = r.crow + (int64_t)c.tap * g.c_win.N;
    off = row * g.ldc + c.c;
    didx = row * odc.ld + c.c;
  } else {
    off = r.crow + n;
    didx = r.drow + n;
  }
  if (odc.thresh) v *= tecm_drop_mult(odc.seed, (uint64_t)didx, odc.thresh, odc.inv);
  if (g.residual) v += g.residual[r.m * g.ldr + n];
  if (g.accumulate) v += g.C[off];
  g.C[off] = v;
}
// one-element form for the split-K reducer
__device__ __forceinline__ void epilogue_store(const TecmGemm& g, const DropCtx& odc, int64_t m, int32_t n, float v) {
  const EpiRow r = epi_row(g, odc, m);
  const EpiCol c = epi_col(g, n);
  epi_elem(g, odc, r, c, v);
}

// Four consecutive columns of one row (n % 4 == 0).  Only used when every pointer / leading dimension the
// epilogue touches is 16-byte friendly (checked on the host: TecmGemm::_p0 carries the flag).
__device__ __forceinline__ void epi_vec4(const TecmGemm& g, const DropCtx& odc, const EpiRow& r, const EpiCol& c,
                                         const float4& bias4, float4 v) {
  const int32_t n = c.n;
  float o[4] = {v.x * g.alpha + bias4.x, v.y * g.alpha + bias4.y, v.z * g.alpha + bias4.z, v.w * g.alpha + bias4.w};
  if (r.rb) {
    const float4 t = *reinterpret_cast<const float4*>(r.rb + n);
    o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
  }
  const bool p16 = (g.io_bf16 & TECM_IO_PRE_BF16) != 0;
  if (g.preact) {
    if (p16) {                             // bf16 pre-activation: act() below is evaluated at the ROUNDED value, the
      tecm_bf16x4 h;                       // point the backward's act'() reads back
      h[0] = (__bf16)o[0]; h[1] = (__bf16)o[1]; h[2] = (__bf16)o[2]; h[3] = (__bf16)o[3];
      *reinterpret_cast<tecm_bf16x4*>(reinterpret_cast<__bf16*>(g.preact) + r.m * g.ldp + n) = h;
      o[0] = (float)h[0]; o[1] = (float)h[1]; o[2] = (float)h[2]; o[3] = (float)h[3];
    } else {
      *reinterpret_cast<float4*>(g.preact + r.m * g.ldp + n) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  if (g.dact_src) {
    float4 t;
    if (p16) {
      const tecm_bf16x4 h = *reinterpret_cast<const tecm_bf16x4*>(reinterpret_cast<const __bf16*>(g.dact_src) + r.m * g.ldd + n);
      t = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
      t = *reinterpret_cast<const float4*>(g.dact_src + r.m * g.ldd + n);
    }
    o[0] *= dgelu_tanh(t.x); o[1] *= dgelu_tanh(t.y);
    o[2] *= dgelu_tanh(t.z); o[3] *= dgelu_tanh(t.w);
  } else if (g.act) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gelu_tanh(o[e]);
  }
  int64_t off, didx;
  if (g.c_win.enabled) {
    const int32_t t_in = r.t0 + c.tap;
    if (t_in < 0 || t_in >= g.c_win.Lin) return;
    const int64_t row = r.crow + (int64_t)c.tap * g.c_win.N;
    off = row * g.ldc + c.c;
    didx = row * odc.ld + c.c;
  } else {
    off = r.crow + n;
    didx = r.drow + n;
  }
  if (odc.thresh) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] *= tecm_drop_mult(odc.seed, (uint64_t)(didx + e), odc.thresh, odc.inv);
  }
  if (g.residual) {
    const float4 t = *reinterpret_cast<const float4*>(g.residual + r.m * g.ldr + n);
    o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
  }
  if (g.accumulate) {
    const float4 t = *reinterpret_cast<const float4*>(g.C + off);
    o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
  }
  if (g.io_bf16 & TECM_IO_C_BF16) {       // the consumer is a bf16-source GEMM: round once here instead of at its loader
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t h;
    h[0] = (__bf16)o[0]; h[1] = (__bf16)o[1]; h[2] = (__bf16)o[2]; h[3] = (__bf16)o[3];
    *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(g.C) + off) = h;
    return;
  }
  *reinterpret_cast<float4*>(g.C + off) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---------------------------------------------------------------------------------------- fast epilogue rows
// gfx9 counts loads AND stores in one counter (vmcnt).  In the generic rolled row loop below every iteration ends in
// a store and the next one starts with code that MAY load (residual / dact_src / accumulate / row bias behind
// wave-uniform branches): the compiler has to cover that with `s_waitcnt vmcnt(0)`, which also waits for the
// previous iteration's store to be acknowledged -- a store round trip per row group, serialised.  (Measured on the
// bf16 kernels, where it shows most: the epilogue of a K = 768 GEMM cost as much as its whole K loop and did not
// depend on the bytes written.)  The common cases therefore run through this straight-line form: NIT row groups
// fully unrolled, the one optional input stream (MODE 1: residual, MODE 2: GELU' source) loaded for ALL groups up
// front, then LDS read -> arithmetic -> stores back to back with counted waits only.  Element arithmetic is
// epi_vec4's, expression for expression.
#ifndef TECM_EPI_STAMP                                   // diagnostics hook (gemm_bf16_p8.hip -DP8_STAMPS); nothing otherwise
#define TECM_EPI_STAMP(i) do { } while (0)
#endif
template <int NIT, int RPI, int MODE>
__device__ __forceinline__ void epi_fast_load(const TecmGemm& g, int lrow, int64_t mrow0, const EpiCol& ecol,
                                              float4 (&in)[NIT]) {
  const int32_t ncol = ecol.ok ? ecol.n : 0;             // lanes outside the matrix load column 0 (valid, never stored)
  if constexpr (MODE == 5) {                             // row bias: row ((m / rb_div) % rb_mod) of a small table (wpe)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int64_t m = mrow0 + it * RPI + lrow;
      m = m < g.M ? m : g.M - 1;
      const uint32_t rr = ((uint32_t)m / (uint32_t)g.rb_div) % (uint32_t)g.rb_mod;
      in[it] = *reinterpret_cast<const float4*>(g.rowbias + (int64_t)rr * g.rb_ld + ncol);
    }
  }
  if constexpr (MODE >= 1 && MODE <= 3) {
    const float* src = MODE == 1 ? g.residual : (MODE == 2 ? g.dact_src : g.C);
    const int64_t ld = MODE == 1 ? g.ldr : (MODE == 2 ? g.ldd : g.ldc);
    // one 64-bit multiply per slab: row group `it` lies it * RPI rows = it * step elements further on (integer multiplies
    // are quarter rate; three of them per address and row group were most of this epilogue's issue time); rows past M read
    // the last row instead (a valid address, the value is never stored)
    const int32_t mb = (int32_t)mrow0 + lrow;            // M < 2^31 (host)
    const int64_t off0 = (int64_t)mb * ld + ncol, step = (int64_t)RPI * ld, off_last = (g.M - 1) * ld + ncol;
    if (MODE == 2 && (g.io_bf16 & TECM_IO_PRE_BF16)) {   // bf16 GELU' source: the RAW 8 bytes travel in in[].x/.y and are
#pragma unroll                                            // widened where they are used (converting here would be a wait)
      for (int it = 0; it < NIT; ++it) {
        const int64_t off = mb + it * RPI < (int32_t)g.M ? off0 + it * step : off_last;
        const float2 raw = *reinterpret_cast<const float2*>(reinterpret_cast<const __bf16*>(src) + off);
        in[it].x = raw.x;
        in[it].y = raw.y;
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t off = mb + it * RPI < (int32_t)g.M ? off0 + it * step : off_last;
      in[it] = *reinterpret_cast<const float4*>(src + off);
    }
  }
}
// FEAT: -1 = the features of the call are tested at run time (wave-uniform branches inside every row group: eight or so
// scalar branches per group, each a fetch redirect -- by the time stamps of DESIGN_HISTORY B.11 most of a plain row group's
// 150 ns); >= 0 = a compile-time feature mask, for the handful of combinations the training step runs (epi_fast_dispatch):
enum { EPI_F_PRE = 1, EPI_F_P16 = 2, EPI_F_ACT = 4, EPI_F_DROP = 8, EPI_F_C16 = 16 };
template <int FEAT, int BIT>
__device__ __forceinline__ bool epi_feat(bool dynamic) {
  if constexpr (FEAT < 0) return dynamic;
  else return (FEAT & BIT) != 0;
}
template <int NIT, int RPI, int STG_LD, int MODE, bool NT, int FEAT = -1>
__device__ __forceinline__ void epi_fast_rows(const TecmGemm& g, const DropCtx& odc, const float* stg, int lrow, int lcol,
                                              int64_t mrow0, const EpiCol& ecol, const float4& bias4,
                                              const float4 (&in)[NIT]) {
  // (Round 5, from per-block time stamps and the ISA of the eight-phase kernel -- DESIGN_HISTORY B.11: a lane outside the
  //  matrix used to `return` here.  That divergent skip of a whole slab made every later slab's first use of bias4 wait
  //  with vmcnt(0) -- the path that skipped still "had the bias load pending" -- i.e. for ALL stores of the previous slab;
  //  and with one LDS read per row group inside the feature branches every group waited for its own read.  Now the
  //  column test only masks the stores, and a slab's staged rows are read four groups at a time before any of them is used.)
  const int32_t n = ecol.n;
  const bool c16 = epi_feat<FEAT, EPI_F_C16>((g.io_bf16 & TECM_IO_C_BF16) != 0);
  const bool p16 = epi_feat<FEAT, EPI_F_P16>((g.io_bf16 & TECM_IO_PRE_BF16) != 0);
  const bool has_pre = epi_feat<FEAT, EPI_F_PRE>(g.preact != nullptr);
  const bool has_act = epi_feat<FEAT, EPI_F_ACT>(g.act != 0);
  const bool has_drop = epi_feat<FEAT, EPI_F_DROP>(odc.thresh != 0);
  constexpr int GRP = NIT < 4 ? NIT : 4;              // (8 at a time spilled 15 registers in the eight-phase kernel)
  static_assert(NIT % GRP == 0, "row groups are read from the staging rows in whole batches");
  // addresses: one 64-bit multiply per tensor and slab, then it * (RPI * ld) per row group (see epi_fast_load)
  const int32_t mb = (int32_t)mrow0 + lrow;              // M < 2^31 (host)
  const int64_t offC0 = (int64_t)mb * g.ldc + n, stepC = (int64_t)RPI * g.ldc;
  const int64_t offP0 = (int64_t)mb * g.ldp + n, stepP = (int64_t)RPI * g.ldp;
  const int64_t idxD0 = (int64_t)mb * odc.ld + n, stepD = (int64_t)RPI * odc.ld;
  float4 vv[GRP];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (it % GRP == 0) {
#pragma unroll
      for (int j = 0; j < GRP; ++j)
        vv[j] = *reinterpret_cast<const float4*>(&stg[((it + j) * RPI + lrow) * STG_LD + lcol]);
    }
    const int64_t m = mb + it * RPI;
    const float4 v = vv[it % GRP];
    const bool row_ok = mb + it * RPI < (int32_t)g.M && ecol.ok;
    if constexpr (MODE == 4) {                         // split-K slab: the raw partial sums
      if (row_ok) *reinterpret_cast<float4*>(g.workspace + ((int64_t)blockIdx.z * g.M + m) * g.N + n) = v;
      continue;
    }
    float o[4] = {v.x * g.alpha + bias4.x, v.y * g.alpha + bias4.y, v.z * g.alpha + bias4.z, v.w * g.alpha + bias4.w};
    if constexpr (MODE == 5) {                         // + row bias, where epi_vec4 adds it: before pre-activation / act / dropout
      o[0] += in[it].x; o[1] += in[it].y; o[2] += in[it].z; o[3] += in[it].w;
    }
    if (has_pre && p16) {                              // bf16 pre-activation (see epi_vec4)
      tecm_bf16x4 hv;
      hv[0] = (__bf16)o[0]; hv[1] = (__bf16)o[1]; hv[2] = (__bf16)o[2]; hv[3] = (__bf16)o[3];
      if (row_ok) {
        tecm_bf16x4* dst = reinterpret_cast<tecm_bf16x4*>(reinterpret_cast<__bf16*>(g.preact) + (offP0 + it * stepP));
        if constexpr (NT) __builtin_nontemporal_store(hv, dst);
        else *dst = hv;
      }
      o[0] = (float)hv[0]; o[1] = (float)hv[1]; o[2] = (float)hv[2]; o[3] = (float)hv[3];
    } else if (has_pre && row_ok) {
      if constexpr (NT) {                              // streamed past L2: the operand panels stay (bf16 LDS-DMA kernel)
        f32x4 nv = {o[0], o[1], o[2], o[3]};
        __builtin_nontemporal_store(nv, reinterpret_cast<f32x4*>(g.preact + (offP0 + it * stepP)));
      } else {
        *reinterpret_cast<float4*>(g.preact + (offP0 + it * stepP)) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    if constexpr (MODE == 2) {
      float t0 = in[it].x, t1 = in[it].y, t2 = in[it].z, t3 = in[it].w;
      if (p16) {                                       // widen the raw bf16 quad carried in .x/.y
        const uint32_t lo = __builtin_bit_cast(uint32_t, in[it].x), hi = __builtin_bit_cast(uint32_t, in[it].y);
        t0 = __builtin_bit_cast(float, lo << 16); t1 = __builtin_bit_cast(float, lo & 0xffff0000u);
        t2 = __builtin_bit_cast(float, hi << 16); t3 = __builtin_bit_cast(float, hi & 0xffff0000u);
      }
      o[0] *= dgelu_tanh(t0); o[1] *= dgelu_tanh(t1);
      o[2] *= dgelu_tanh(t2); o[3] *= dgelu_tanh(t3);
    } else {
      if (has_act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gelu_tanh(o[e]);
      }
    }
    int64_t off = offC0 + it * stepC, didx = idxD0 + it * stepD;
    bool st_ok = row_ok;
    if constexpr (MODE == 6) {                         // window scatter (epi_vec4's address arithmetic): column n = (tap, c)
      const WinRow w = win_row(g.c_win, m < g.M ? m : g.M - 1, g.M);
      const int32_t t_in = w.t0 + ecol.tap;
      const int64_t row = w.srow + (int64_t)ecol.tap * g.c_win.N;
      st_ok = row_ok && t_in >= 0 && t_in < g.c_win.Lin;
      off = row * g.ldc + ecol.c;
      didx = row * odc.ld + ecol.c;
    }
    if (has_drop) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] *= tecm_drop_mult(odc.seed, (uint64_t)(didx + e), odc.thresh, odc.inv);
    }
    if constexpr (MODE == 1 || MODE == 3) {            // + residual, or + the previous contents of C (accumulate)
      o[0] += in[it].x; o[1] += in[it].y; o[2] += in[it].z; o[3] += in[it].w;
    }
    if (st_ok) {
      if constexpr (NT) {
        if (c16) {
          tecm_bf16x4 hv;
          hv[0] = (__bf16)o[0]; hv[1] = (__bf16)o[1]; hv[2] = (__bf16)o[2]; hv[3] = (__bf16)o[3];
          __builtin_nontemporal_store(hv, reinterpret_cast<tecm_bf16x4*>(reinterpret_cast<__bf16*>(g.C) + off));
        } else {
          f32x4 nv = {o[0], o[1], o[2], o[3]};
          __builtin_nontemporal_store(nv, reinterpret_cast<f32x4*>(g.C + off));
        }
      } else {
        if (c16) tecm_store_bf16x4(reinterpret_cast<__bf16*>(g.C) + off, o[0], o[1], o[2], o[3]);
        else *reinterpret_cast<float4*>(g.C + off) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
#ifdef TECM_EPI_SLEEP                                    // diagnostics: pace the epilogue's stores (tools/build_variant.py)
    __builtin_amdgcn_s_sleep(TECM_EPI_SLEEP);
#endif
  }
}
// One wave's staged block of SLABS x (NIT * RPI) rows: the input stream of slab s+1 is requested BEFORE the stores
// of slab s are issued, so that waiting for it does not mean waiting for those stores (vmcnt retires in order).
// stage(s) parks slab s of the accumulators in the wave's PRIVATE staging rows (no barrier: nobody else reads them).
template <int SLABS, int NIT, int RPI, int STG_LD, int MODE, bool NT, int FEAT = -1, typename StageFn>
__device__ __forceinline__ void epi_fast_block(const TecmGemm& g, const DropCtx& odc, const float* stg, int lrow, int lcol,
                                               int64_t mrow0, const EpiCol& ecol, const float4& bias4, StageFn&& stage) {
  float4 in[2][NIT];
  epi_fast_load<NIT, RPI, MODE>(g, lrow, mrow0, ecol, in[0]);
  static_for<SLABS>([&](auto sc) {
    constexpr int sl = decltype(sc)::value;
    TECM_EPI_STAMP(3 * sl);
    stage(sc);
    TECM_EPI_STAMP(3 * sl + 1);
    if constexpr (sl + 1 < SLABS)
      epi_fast_load<NIT, RPI, MODE>(g, lrow, mrow0 + (int64_t)(sl + 1) * NIT * RPI, ecol, in[(sl + 1) & 1]);
    epi_fast_rows<NIT, RPI, STG_LD, MODE, NT, FEAT>(g, odc, stg, lrow, lcol, mrow0 + (int64_t)sl * NIT * RPI, ecol, bias4, in[sl & 1]);
    TECM_EPI_STAMP(3 * sl + 2);
  });
}
// which epilogues the straight-line form serves (everything else: the generic loop)
// MODE: 0 no input stream, 1 residual, 2 GELU' source, 3 accumulate into C, 4 split-K slab store, 5 row bias (the patch
// projection's + wpe), 6 window scatter of C (the patch projection's input gradient); -1: generic loop
// (Round 5: 5 and 6 were generic-loop cases -- the patch projection forward took 184 us against 87 for the same contraction
//  with a plain epilogue, its d-input 144 against 83: a store round trip per row group, see "fast epilogue rows" above.)
__device__ __forceinline__ int epi_fast_mode(const TecmGemm& g) {
  if (!(g.io_bf16 & TECM_P0_VEC4)) return -1;
  if (gridDim.z > 1) return 4;
  const int streams = (g.residual ? 1 : 0) + (g.dact_src ? 1 : 0) + (g.accumulate ? 1 : 0);
  if (g.c_win.enabled)                                 // 6: fp32 window scatter with bias / act / dropout, nothing else
    return (streams == 0 && !g.rowbias && !g.preact && !(g.io_bf16 & (TECM_IO_C_BF16 | TECM_IO_PRE_BF16))) ? 6 : -1;
  if (g.rowbias) return streams == 0 ? 5 : -1;        // 5: row bias in place of the input stream
  if (streams > 1) return -1;
  return g.residual ? 1 : (g.dact_src ? 2 : (g.accumulate ? 3 : 0));
}
// dispatch on the (wave-uniform) mode
// SPECIAL (the eight-phase bf16 kernel): the four feature combinations of the bf16 training step run branch-free
template <int SLABS, int NIT, int RPI, int STG_LD, bool NT = false, bool SPECIAL = false, typename StageFn>
__device__ __forceinline__ void epi_fast_dispatch(int mode, const TecmGemm& g, const DropCtx& odc, const float* stg, int lrow,
                                                  int lcol, int64_t mrow0, const EpiCol& ecol, const float4& bias4,
                                                  StageFn&& stage) {
  if constexpr (SPECIAL) {
    const int feat = (g.preact ? EPI_F_PRE : 0) | ((g.io_bf16 & TECM_IO_PRE_BF16) ? EPI_F_P16 : 0) | (g.act ? EPI_F_ACT : 0) |
                     (odc.thresh ? EPI_F_DROP : 0) | ((g.io_bf16 & TECM_IO_C_BF16) ? EPI_F_C16 : 0);
    if (mode == 0 && feat == EPI_F_C16)                                  // bf16 result (+ bias): c_attn, the d-input GEMMs
      return epi_fast_block<SLABS, NIT, RPI, STG_LD, 0, NT, EPI_F_C16>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
    if (mode == 0 && feat == (EPI_F_PRE | EPI_F_P16 | EPI_F_ACT | EPI_F_C16))                    // mlp.c_fc
      return epi_fast_block<SLABS, NIT, RPI, STG_LD, 0, NT, EPI_F_PRE | EPI_F_P16 | EPI_F_ACT | EPI_F_C16>(g, odc, stg, lrow, lcol, mrow0,
                                                                                                    ecol, bias4, stage);
    if (mode == 1 && feat == EPI_F_DROP)                                 // attn.c_proj / mlp.c_proj: dropout + residual, fp32
      return epi_fast_block<SLABS, NIT, RPI, STG_LD, 1, NT, EPI_F_DROP>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
    if (mode == 2 && (feat & ~EPI_F_ACT) == (EPI_F_P16 | EPI_F_C16))     // d mlp.c_proj x GELU' (mode 2 ignores `act`)
      return epi_fast_block<SLABS, NIT, RPI, STG_LD, 2, NT, EPI_F_P16 | EPI_F_C16>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
  }
  if (mode == 0) epi_fast_block<SLABS, NIT, RPI, STG_LD, 0, NT>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
  else if (mode == 1) epi_fast_block<SLABS, NIT, RPI, STG_LD, 1, NT>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
  else if (mode == 2) epi_fast_block<SLABS, NIT, RPI, STG_LD, 2, NT>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
  else if (mode == 3) epi_fast_block<SLABS, NIT, RPI, STG_LD, 3, NT>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
  else if (mode == 5) epi_fast_block<SLABS, NIT, RPI, STG_LD, 5, NT>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
  else if (mode == 6) epi_fast_block<SLABS, NIT, RPI, STG_LD, 6, NT>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
  else epi_fast_block<SLABS, NIT, RPI, STG_LD, 4, NT>(g, odc, stg, lrow, lcol, mrow0, ecol, bias4, stage);
}

// ---------------------------------------------------------------------------------------- block epilogue
// C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
// Each wave parks its WTM x WTN accumulator block in LDS (the operand buffers are dead by now) and walks it
// row by row in ONE rolled loop (small code: the epilogue is executed once per block and must not thrash the
// instruction cache): row state is resolved once per row, a lane keeps its columns, and the global stores are
// whole contiguous row segments (float4 per lane when the host found every pointer / leading dimension
// 16-byte friendly: g.io_bf16 == 1).
template <int MT, int NT, int WTM, int WTN, int STG_LD>
__device__ __forceinline__ void block_epilogue(const TecmGemm& g, f32x16 (&acc)[MT][NT], float* smem, int wave, int lane,
                                               int wm, int wn, int64_t m0, int64_t n0) {
  const int r = lane & 31, h = lane >> 5;
  const DropCtx odc = make_drop(g.out_drop);
  const bool split = gridDim.z > 1;
  float* stg = smem + wave * (WTM * STG_LD);
  auto stage_all = [&](auto) {
    static_for<MT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      static_for<16>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        static_for<NT>([&](auto jc) {
          constexpr int jn = decltype(jc)::value;
          stg[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * STG_LD + jn * 32 + r] = acc[i][jn][e];
        });
      });
    });
  };
  if (g.io_bf16 & TECM_P0_VEC4) {
    constexpr int LPR = WTN / 4;                       // lanes per row
    constexpr int RPI = 64 / LPR;                      // rows per iteration
    const int lcol = (lane % LPR) * 4, lrow = lane / LPR;
    const EpiCol ecol = epi_col(g, n0 + wn * WTN + lcol);
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + (ecol.ok ? ecol.n : 0));   // (lanes outside the matrix: column 0, never stored)
    const int fmode = epi_fast_mode(g);
    if (fmode >= 0) {
      // staging rows are private to the wave: no barrier (a __syncthreads here would also drain every store)
      epi_fast_dispatch<1, WTM / RPI, RPI, STG_LD>(fmode, g, odc, stg, lrow, lcol, m0 + wm * WTM, ecol, bias4, stage_all);
      return;
    }
    stage_all(0);
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < WTM / RPI; ++it) {
      const int rl = it * RPI + lrow;
      const int64_t m = m0 + wm * WTM + rl;
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
    stage_all(0);
    __syncthreads();
    constexpr int RPI = 64 / WTN;                      // 1 (WTN = 64) or 2 (WTN = 32)
    const int lcol = lane % WTN, lrow = lane / WTN;
    const EpiCol ecol = epi_col(g, n0 + wn * WTN + lcol);
#pragma unroll 1
    for (int it = 0; it < WTM / RPI; ++it) {
      const int rl = it * RPI + lrow;
      const int64_t m = m0 + wm * WTM + rl;
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
}

template <int ALAY, int BLAY, int AVEC, int BVEC, int BN, bool WIN, bool DROP, int BMT = BM>
__global__ __launch_bounds__(threads_for(BN), (AVEC == 4 && BVEC == 4) ? (threads_for(BN) == 512 ? (BMT > 128 ? 2 : 4) : 2) : 1) void gemm_kernel(const TecmGemm g,
                                                                                           int tiles_m, int tiles_n,
                                                                                           int k_chunk) {
  constexpr int NTHREADS = threads_for(BN);
  constexpr int NWAVES = NTHREADS / 64;
  constexpr int WN = BN >= 128 ? (NWAVES == 8 ? (BMT > 128 ? 2 : 4) : 2) : (BN == 64 ? 2 : 1);
  constexpr int WM = NWAVES / WN;
  constexpr int WTM = BMT / WM;
  static_assert(WTM >= 32 && WTM % 32 == 0, "wave tile");
  constexpr int WTN = BN / WN;
  constexpr int MT = WTM / 32;
  constexpr int NT = WTN / 32;
  using AStager = Stager<ALAY == TECM_A_KM, BMT, AVEC, WIN, DROP, NTHREADS>;
  using BStager = Stager<BLAY == TECM_B_KN, BN, BVEC, WIN, DROP, NTHREADS>;
  constexpr int A_FLOATS = AStager::R * AStager::LD;
  constexpr int B_FLOATS = BStager::R * BStager::LD;
  constexpr int TILE_FLOATS = A_FLOATS + B_FLOATS;
  constexpr int STG_LD = WTN + 4;                       // epilogue staging: 32 rows x WTN cols per wave, 16-B rows
  // two operand buffers: tile t+1 is written into the idle one between the MFMAs of tile t (one barrier per tile)
  constexpr int STG_FLOATS = NWAVES * WTM * STG_LD;          // whole accumulator block of the 4 waves
  constexpr int SMEM_FLOATS = 2 * TILE_FLOATS > STG_FLOATS ? 2 * TILE_FLOATS : STG_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];

  // XCD-aware, bijective block -> tile map: blocks that share an XCD (id % 8) get a contiguous
  // run of tiles, n fastest, so an A row-panel is re-read from that XCD's L2.
  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  // grouped order inside the XCD's run: 8 m-tiles x all n-tiles per group, m fastest, so the ~64 blocks
  // resident on one XCD form an (8 m) x (8 n) super-tile whose A and B panels both live in that XCD's 4 MiB L2
  // m-tiles per group (host-tunable: TecmGemm::_p1).  Default from the measured L2-miss volume:
  // 2 when the matrix is at most 8 tiles wide (N = 768 / 800: 1.3 GB per launch instead of 1.7), 8 otherwise
  const int GROUP_M = g._p1 > 0 ? g._p1 : (tiles_n <= 8 ? 2 : 8);
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * BMT;
  const int64_t n0 = (int64_t)tn * BN;
  const int32_t kbeg = blockIdx.z * k_chunk;
  const int32_t kend = min((int32_t)g.K, kbeg + k_chunk);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  const DropCtx adc = make_drop(g.a_drop), bdc = make_drop(g.b_drop);

  AStager sa;
  BStager sb;
  typename AStager::Tile ta;              // the K-tile in flight (main loop); the prologue adds a second pair
  typename BStager::Tile tb;
  sa.init(g.A, g.a_win, g.lda, m0, g.M, kbeg, m0, g.M, adc);
  sb.init(g.B, g.b_win, g.ldb, n0, g.N, kbeg, n0, g.N, bdc);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // one q-step = k sub-range [8q, 8q+8): fragment reads and the MT*NT*4 MFMAs that consume them
  auto read_frags = [&](const float* As, const float* Bs, auto qc, float (&af)[MT][4], float (&bf)[NT][4]) {
    constexpr int q = decltype(qc)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * WTM + i * 32 + r;
      if constexpr (ALAY == TECM_A_MK) {
        const float4 v = *reinterpret_cast<const float4*>(&As[row * LDK + 8 * q + 4 * h]);
        af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) af[i][j] = As[(8 * q + 4 * h + j) * AStager::LD + row];
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int col = wn * WTN + i * 32 + r;
      if constexpr (BLAY == TECM_B_NK) {
        const float4 v = *reinterpret_cast<const float4*>(&Bs[col * LDK + 8 * q + 4 * h]);
        bf[i][0] = v.x; bf[i][1] = v.y; bf[i][2] = v.z; bf[i][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[i][j] = Bs[(8 * q + 4 * h + j) * BStager::LD + col];
      }
    }
  };
  auto do_mfma = [&](const float (&af)[MT][4], const float (&bf)[NT][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j], bf[jn][j], acc[i][jn], 0, 0, 0);
  };

  // prologue: K-tiles 0 and 1 go out together (the accumulators are not live yet, so the second register set is
  // free): one exposed global round trip per output tile instead of two.  Tile 0 -> LDS buffer 0, tile 1 stays
  // in flight in (ta, tb).
  {
    typename AStager::Tile ta0;
    typename BStager::Tile tb0;
    sa.load(ta0, g.A, g.a_win, g.lda, kbeg, kend, adc);
    sb.load(tb0, g.B, g.b_win, g.ldb, kbeg, kend, bdc);
    if (kbeg + BK < kend) {
      sa.load(ta, g.A, g.a_win, g.lda, kbeg + BK, kend, adc);
      sb.load(tb, g.B, g.b_win, g.ldb, kbeg + BK, kend, bdc);
    }
    sa.store(ta0, smem, adc);
    sb.store(tb0, smem + A_FLOATS, bdc);
  }
  __syncthreads();

  // ---- main loop.  One K-tile = 4 q-steps of MT*NT*4 MFMAs.  Per q-step the wave also has to issue
  //   * the fragment reads of the NEXT q-step (second fragment register set),
  //   * a quarter of the staging work: park tile t+1 (already in registers) in the idle LDS buffer and refill
  //     those registers with tile t+2 from global memory.
  // In the steady state (FULL: tiles t+1 and t+2 exist) the body is branch-free, i.e. one scheduling region,
  // and sched_group_barrier pins the order "1 MFMA, 1 memory instruction, 1 MFMA, ..." so every LDS / global
  // instruction issues in the shadow of a 64-cycle MFMA and nothing is waited for right after it was issued.
  float fa[2][MT][4], fb[2][NT][4];
  constexpr int ANV = AStager::NV, BNV = BStager::NV;
  auto tile_body = [&](const float* Ac, const float* Bc, float* An, int32_t k0, auto fullc) {
    constexpr bool FULL = decltype(fullc)::value;
    read_frags(Ac, Bc, std::integral_constant<int, 0>{}, fa[0], fb[0]);
    static_for<4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (q < 3 && !(TECM_ABLATE >= 2 && FULL)) read_frags(Ac, Bc, std::integral_constant<int, q + 1>{}, fa[(q + 1) & 1], fb[(q + 1) & 1]);
      do_mfma(fa[q & 1], fb[q & 1]);
      // Staging lives in the LAST q-step: park tile t+1 (loaded one whole tile ago) in the idle LDS buffer, then
      // refill the same registers with tile t+2 straight away -- every global load gets a full K-tile of MFMAs
      // (32-64 per wave, x the waves sharing the SIMD) to land before its ds_write.  Spreading the staging over
      // the q-steps halves that distance for the vectors staged early, which is what stalls under L2-miss latency.
      if constexpr (q == TECM_STAGE_Q && !((TECM_ABLATE == 1 || TECM_ABLATE == 2) && FULL)) {
        if (FULL || k0 + BK < kend) {
          if constexpr (TECM_ABLATE == 3 && FULL) {
#pragma unroll
            for (int i = 0; i < ANV; ++i) asm volatile("" ::"v"(ta.regs[i][0]), "v"(ta.regs[i][1]), "v"(ta.regs[i][2]), "v"(ta.regs[i][3]));
#pragma unroll
            for (int i = 0; i < BNV; ++i) asm volatile("" ::"v"(tb.regs[i][0]), "v"(tb.regs[i][1]), "v"(tb.regs[i][2]), "v"(tb.regs[i][3]));
          } else if constexpr (FULL && !WIN) {
            sa.store_steady(ta, An, adc);
            sb.store_steady(tb, An + A_FLOATS, bdc);
          } else {
            sa.store(ta, An, adc);
            sb.store(tb, An + A_FLOATS, bdc);
          }
        }
        if constexpr (FULL && !WIN && TECM_ABLATE != 4) {
          sa.load_steady(ta, g.lda, adc);
          sb.load_steady(tb, g.ldb, bdc);
        } else if ((FULL && TECM_ABLATE != 4) || (!FULL && k0 + 2 * BK < kend)) {
          sa.load(ta, g.A, g.a_win, g.lda, k0 + 2 * BK, kend, adc);
          sb.load(tb, g.B, g.b_win, g.ldb, k0 + 2 * BK, kend, bdc);
        }
      }
      if constexpr (FULL && !WIN && !DROP && AVEC == 4 && BVEC == 4) {
        // pin "1 MFMA, then memory instructions" so every LDS / global instruction issues in the shadow of a
        // 64-cycle MFMA: q < 3 -> the next q-step's fragment reads; q == 3 -> first the LDS writes, then the loads
        constexpr int NMFMA = MT * NT * 4;
        constexpr int NREAD = MT * (ALAY == TECM_A_MK ? 1 : 4) + NT * (BLAY == TECM_B_NK ? 1 : 4);
        constexpr int HALF = NMFMA / 2;
        constexpr int SPS = (ANV + BNV + HALF - 1) / HALF;
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
          if constexpr (q < 3) {
            if (m < NREAD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
          }
          if constexpr (q == TECM_STAGE_Q) {
            if (m < HALF) __builtin_amdgcn_sched_group_barrier(0x200, SPS, 0);     // DS writes
            else __builtin_amdgcn_sched_group_barrier(0x020, SPS, 0);              // VMEM reads
          }
        }
      }
    });
  };

  int cur = 0;
  int32_t k0 = kbeg;
  for (; k0 + 3 * BK <= kend; k0 += BK) {                // steady state: tiles t+1 and t+2 lie entirely inside [kbeg, kend)
    tile_body(smem + cur * TILE_FLOATS, smem + cur * TILE_FLOATS + A_FLOATS, smem + (cur ^ 1) * TILE_FLOATS, k0,
              std::true_type{});
    __syncthreads();
    cur ^= 1;
  }
  for (; k0 < kend; k0 += BK) {                          // last (up to three) tiles, masked loads
    tile_body(smem + cur * TILE_FLOATS, smem + cur * TILE_FLOATS + A_FLOATS, smem + (cur ^ 1) * TILE_FLOATS, k0,
              std::false_type{});
    __syncthreads();
    cur ^= 1;
  }

#ifdef TECM_ABLATE_NOEPI                                 // diagnostics: K loop without the epilogue (tools/build_variant.py)
  float keep = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) keep += acc[i][j][e];
  if (keep == 12345.678f) g.C[0] = keep;
#else
  block_epilogue<MT, NT, WTM, WTN, STG_LD>(g, acc, smem, wave, lane, wm, wn, m0, n0);
#endif
}

template <int ALAY, int BLAY, int AVEC, int BVEC, int BN, bool WIN, bool DROP, int BMT = BM>
int launch(const TecmGemm& g, hipStream_t st) {
  const int tiles_m = (int)((g.M + BMT - 1) / BMT);
  const int tiles_n = (int)((g.N + BN - 1) / BN);
  int splits = g.split_k > 1 ? g.split_k : 1;
  int k_chunk = (int)(((g.K + splits - 1) / splits + BK - 1) / BK) * BK;
  splits = (int)((g.K + k_chunk - 1) / k_chunk);
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)splits);
  hipLaunchKernelGGL((gemm_kernel<ALAY, BLAY, AVEC, BVEC, BN, WIN, DROP, BMT>), grid, dim3(threads_for(BN)), 0, st, g, tiles_m,
                     tiles_n, k_chunk);
  TECM_CHECK_LAUNCH("tecm_gemm_f32");
  return splits;      // > 0: number of K splits actually launched
}

// (4,4) vectors: all four WIN/DROP variants; narrower vectors: the general (WIN, DROP) variant only.
template <int ALAY, int BLAY, int BN>
int dispatch(const TecmGemm& g, int avec, int bvec, bool win, bool drop, hipStream_t st) {
  if (avec == 4 && bvec == 4) {
    if (!win && !drop) return launch<ALAY, BLAY, 4, 4, BN, false, false>(g, st);
    if (win && !drop) return launch<ALAY, BLAY, 4, 4, BN, true, false>(g, st);
    if (!win && drop) return launch<ALAY, BLAY, 4, 4, BN, false, true>(g, st);
    return launch<ALAY, BLAY, 4, 4, BN, true, true>(g, st);
  }
  if constexpr (ALAY == TECM_A_MK) {
    if (avec >= 2) return launch<ALAY, BLAY, 2, 1, BN, true, true>(g, st);
  } else {
    if (avec == 4 && bvec >= 2) return launch<ALAY, BLAY, 4, 2, BN, true, true>(g, st);
  }
  return launch<ALAY, BLAY, 1, 1, BN, true, true>(g, st);
}

// M <= 64 (the Conv1d weight gradients of the first block: M = C_out = 64, K = 1.1 M rows): 64-row block tiles,
// so that half of every MFMA is not spent on clamped rows.  float4 loaders only.
template <int ALAY, int BLAY, int BN>
int dispatch_m64(const TecmGemm& g, bool win, bool drop, hipStream_t st) {
  if (!win && !drop) return launch<ALAY, BLAY, 4, 4, BN, false, false, 64>(g, st);
  if (win && !drop) return launch<ALAY, BLAY, 4, 4, BN, true, false, 64>(g, st);
  if (!win && drop) return launch<ALAY, BLAY, 4, 4, BN, false, true, 64>(g, st);
  return launch<ALAY, BLAY, 4, 4, BN, true, true, 64>(g, st);
}

}  // namespace tecm_gemm

// one translation unit per layout pair (parallel compilation)
int tecm_gemm_dispatch_mk_nk(const TecmGemm& g, int avec, int bvec, bool win, bool drop, hipStream_t st);
int tecm_gemm_dispatch_mk_kn(const TecmGemm& g, int avec, int bvec, bool win, bool drop, hipStream_t st);
int tecm_gemm_dispatch_km_kn(const TecmGemm& g, int avec, int bvec, bool win, bool drop, hipStream_t st);
