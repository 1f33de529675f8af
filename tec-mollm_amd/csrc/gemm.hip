// fp32 GEMM on the exact-f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   C[M,N] = epilogue(alpha * A_view[M,K] . B_view[K,N])
//
// One kernel family serves every dense contraction of the TEC-MoLLM path (see include/tecmollm.h):
// operands are addressed through "row views" so that Conv1d-over-time, the strided 1x1 conv,
// latent patching and the head's flatten never materialise an im2col / permuted copy.
//
// Block = 256 threads = 4 waves, tile BM x BN x BK = 128 x {128,32} x 32.
//   BN=128: waves 2(m) x 2(n), each wave 64x64 = 2x2 MFMA 32x32 tiles (64 accumulator VGPRs)
//   BN= 32: waves 4(m) x 1(n), each wave 32x32
// LDS tiles keep the operand's own orientation:
//   [row][k] tiles (MK / NK): leading dim 36 floats -> ds_read_b128 of 4 consecutive k is
//       conflict-free (16 lanes x 4 banks, row stride 36 = 4 mod 32 hits 16 distinct slots);
//   [k][row] tiles (KM / KN): leading dim rows+4 -> ds_read_b32, lanes 0..31 consecutive banks.
// The k index fed to MFMA step (q,j) by lane half h is k = 8q + 4h + j for BOTH operands, which is
// what makes the b128 read legal (any bijection of k works as long as A and B agree).
// Global -> register prefetch of tile t+1 is issued before the MFMAs of tile t (one LDS buffer).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr int NTHREADS = 256;

struct DropCtx {
  uint64_t seed;
  int64_t ld;
  uint32_t thresh;
  float inv;
};
__device__ __forceinline__ DropCtx make_drop(const TecmDrop& d) {
  DropCtx c;
  c.seed = d.seed;
  c.ld = d.ld;
  c.thresh = d.p > 0.f ? tecm_drop_thresh(d.p) : 0u;
  c.inv = d.p > 0.f ? 1.0f / (1.0f - d.p) : 1.0f;
  return c;
}

// Load VEC consecutive inner elements of one view row.  `inner` and `inner_lim` are multiples of VEC.
template <int VEC>
__device__ __forceinline__ void load_elems(const float* __restrict__ P, const TecmWin& w, const RowRef& r,
                                           int64_t ld, int32_t inner, int32_t inner_lim, const DropCtx& dc,
                                           float (&out)[VEC]) {
  bool ok = r.t0 != TECM_ROW_INVALID && inner < inner_lim;
  int64_t off, didx;
  if (w.enabled) {
    const int32_t tap = inner / w.Cw;
    const int32_t c = inner - tap * w.Cw;
    const int32_t t_in = r.t0 + tap;
    ok = ok && t_in >= 0 && t_in < w.Lin;
    const int64_t row = r.srow + (int64_t)tap * w.N;
    off = row * ld + c;
    didx = row * dc.ld + c;
  } else {
    off = r.srow * ld + inner;
    didx = r.srow * dc.ld + inner;
  }
  if (ok) {
    if constexpr (VEC == 4) {
      const float4 v = *reinterpret_cast<const float4*>(P + off);
      out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else if constexpr (VEC == 2) {
      const float2 v = *reinterpret_cast<const float2*>(P + off);
      out[0] = v.x; out[1] = v.y;
    } else {
      out[0] = P[off];
    }
    if (dc.thresh) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) out[e] *= tecm_drop_mult(dc.seed, (uint64_t)(didx + e), dc.thresh, dc.inv);
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[e] = 0.f;
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

// A tile stager.  ROWK=false: tile rows are the (fixed) M/N index, inner index is k  ([row][k], LD=36)
//                 ROWK=true : tile rows are k, inner index is the (fixed) M/N index ([k][row], LD=ROWS+4)
template <bool ROWK, int ROWS, int VEC>
struct Stager {
  static constexpr int R = ROWK ? BK : ROWS;          // LDS rows
  static constexpr int CI = ROWK ? ROWS : BK;         // LDS inner extent
  static constexpr int LD = ROWK ? ROWS + 4 : LDK;
  static constexpr int VPR = CI / VEC;
  static constexpr int NV = (R * VPR) / NTHREADS;
  static constexpr int RSTEP = NTHREADS / VPR;
  static_assert((R * VPR) % NTHREADS == 0 && NTHREADS % VPR == 0, "tile/thread mapping");
  float regs[NV][VEC];
  RowRef fixed[ROWK ? 1 : NV];   // per-thread row refs when the rows are fixed across the K loop

  __device__ __forceinline__ void init(const TecmWin& w, int64_t row0, int64_t rows_total) {
    if constexpr (!ROWK) {
      const int r0 = threadIdx.x / VPR;
#pragma unroll
      for (int i = 0; i < NV; ++i) fixed[i] = make_rowref(w, w.enabled, row0 + r0 + i * RSTEP, rows_total);
    }
  }
  // k0: first k of this tile; klim: K (or split end); fixed0/fixed_lim: first fixed index / its bound
  __device__ __forceinline__ void load(const float* __restrict__ P, const TecmWin& w, int64_t ld, int32_t k0,
                                       int32_t klim, int64_t fixed0, int64_t fixed_lim, const DropCtx& dc) {
    const int cv = (threadIdx.x % VPR) * VEC;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if constexpr (!ROWK) {
        load_elems<VEC>(P, w, fixed[i], ld, k0 + cv, klim, dc, regs[i]);
      } else {
        // rows are k: for the window view the row is a reduction index m' = k, inner = fixed index
        const RowRef rr = make_rowref(w, w.enabled, (int64_t)k0 + r0 + i * RSTEP, klim);
        load_elems<VEC>(P, w, rr, ld, (int32_t)(fixed0 + cv), (int32_t)fixed_lim, dc, regs[i]);
      }
    }
  }
  __device__ __forceinline__ void store(float* lds) const {
    const int cv = (threadIdx.x % VPR) * VEC;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int i = 0; i < NV; ++i) lds_store<VEC>(lds + (r0 + i * RSTEP) * LD + cv, regs[i]);
  }
};


__device__ __forceinline__ void epilogue_store(const TecmGemm& g, const DropCtx& odc, int64_t m, int32_t n, float v) {
  v *= g.alpha;
  if (g.bias) v += g.bias[n];
  if (g.rowbias) v += g.rowbias[(int64_t)((m / g.rb_div) % g.rb_mod) * g.rb_ld + n];
  if (g.preact) g.preact[m * g.ldp + n] = v;
  if (g.dact_src)
    v *= apply_dact(g.act, g.dact_src[m * g.ldd + n]);   // backward through `act`
  else
    v = apply_act(g.act, v);
  int64_t off, didx;
  bool ok = true;
  if (g.c_win.enabled) {
    const RowRef r = make_rowref(g.c_win, true, m, g.M);
    const int32_t tap = n / g.c_win.Cw;
    const int32_t c = n - tap * g.c_win.Cw;
    const int32_t t_in = r.t0 + tap;
    ok = t_in >= 0 && t_in < g.c_win.Lin;
    const int64_t row = r.srow + (int64_t)tap * g.c_win.N;
    off = row * g.ldc + c;
    didx = row * odc.ld + c;
  } else {
    off = m * g.ldc + n;
    didx = m * odc.ld + n;
  }
  if (!ok) return;
  if (odc.thresh) v *= tecm_drop_mult(odc.seed, (uint64_t)didx, odc.thresh, odc.inv);
  if (g.residual) v += g.residual[m * g.ldr + n];
  if (g.accumulate) v += g.C[off];
  g.C[off] = v;
}

template <int ALAY, int BLAY, int AVEC, int BVEC, int BN>
__global__ __launch_bounds__(NTHREADS, (AVEC == 4 && BVEC == 4) ? 2 : 1) void gemm_kernel(const TecmGemm g, int tiles_m, int tiles_n, int k_chunk) {
  constexpr int WN = BN >= 128 ? 2 : 1;
  constexpr int WM = 4 / WN;
  constexpr int WTM = BM / WM;
  constexpr int WTN = BN / WN;
  constexpr int MT = WTM / 32;
  constexpr int NT = WTN / 32;
  using AStager = Stager<ALAY == TECM_A_KM, BM, AVEC>;
  using BStager = Stager<BLAY == TECM_B_KN, BN, BVEC>;
  __shared__ __attribute__((aligned(16))) float As[AStager::R * AStager::LD];
  __shared__ __attribute__((aligned(16))) float Bs[BStager::R * BStager::LD];

  // XCD-aware, bijective block -> tile map: blocks that share an XCD (id % 8) get a contiguous
  // run of tiles, n fastest, so an A row-panel is re-read from that XCD's L2.
  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
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
  sa.init(g.a_win, m0, g.M);
  sb.init(g.b_win, n0, g.N);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  sa.load(g.A, g.a_win, g.lda, kbeg, kend, m0, g.M, adc);
  sb.load(g.B, g.b_win, g.ldb, kbeg, kend, n0, g.N, bdc);
  sa.store(As);
  sb.store(Bs);
  __syncthreads();

  for (int32_t k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      sa.load(g.A, g.a_win, g.lda, k0 + BK, kend, m0, g.M, adc);
      sb.load(g.B, g.b_win, g.ldb, k0 + BK, kend, n0, g.N, bdc);
    }
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      float af[MT][4], bf[NT][4];
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
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int jn = 0; jn < NT; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j], bf[jn][j], acc[i][jn], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      sa.store(As);
      sb.store(Bs);
      __syncthreads();
    }
  }

  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
  const DropCtx odc = make_drop(g.out_drop);
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t m = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (m >= g.M) continue;
#pragma unroll
      for (int jn = 0; jn < NT; ++jn) {
        const int64_t n = n0 + wn * WTN + jn * 32 + r;
        if (n >= g.N) continue;
        if (split)
          g.workspace[((int64_t)blockIdx.z * g.M + m) * g.N + n] = acc[i][jn][e];
        else
          epilogue_store(g, odc, m, (int32_t)n, acc[i][jn][e]);
      }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const TecmGemm g, int splits) {
  const int64_t total = g.M * g.N;
  const DropCtx odc = make_drop(g.out_drop);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += g.workspace[(int64_t)s * total + i];
    const int64_t m = i / g.N;
    epilogue_store(g, odc, m, (int32_t)(i - m * g.N), v);
  }
}

template <int ALAY, int BLAY, int AVEC, int BVEC, int BN>
int launch(const TecmGemm& g, hipStream_t st) {
  const int tiles_m = (int)((g.M + BM - 1) / BM);
  const int tiles_n = (int)((g.N + BN - 1) / BN);
  int splits = g.split_k > 1 ? g.split_k : 1;
  int k_chunk = (int)(((g.K + splits - 1) / splits + BK - 1) / BK) * BK;
  splits = (int)((g.K + k_chunk - 1) / k_chunk);
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)splits);
  hipLaunchKernelGGL((gemm_kernel<ALAY, BLAY, AVEC, BVEC, BN>), grid, dim3(NTHREADS), 0, st, g, tiles_m, tiles_n,
                     k_chunk);
  TECM_CHECK_LAUNCH("tecm_gemm_f32");
  if (splits > 1) {
    const int64_t total = g.M * g.N;
    const int64_t want = (total + 255) / 256;
    const int blocks = (int)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, g, splits);
    TECM_CHECK_LAUNCH("tecm_gemm_f32/splitk_reduce");
  }
  return TECM_OK;
}

template <int ALAY, int BLAY, int BN>
int dispatch_vec(const TecmGemm& g, int avec, int bvec, hipStream_t st) {
  if (avec == 4 && bvec == 4) return launch<ALAY, BLAY, 4, 4, BN>(g, st);
  if constexpr (ALAY == TECM_A_MK) {
    if (avec >= 2) return launch<ALAY, BLAY, 2, 1, BN>(g, st);
  } else {
    if (avec == 4 && bvec >= 2) return launch<ALAY, BLAY, 4, 2, BN>(g, st);
  }
  return launch<ALAY, BLAY, 1, 1, BN>(g, st);
}

// widest vector (4, 2 or 1 floats) the operand's address pattern allows
int pick_vec(const float* p, int64_t ld, const TecmWin& w, bool inner_is_k, int64_t K, int64_t inner_extent) {
  int v = 4;
  while (v > 1) {
    bool ok = tecm_aligned(p, 4 * v) && (ld % v == 0);
    if (w.enabled) ok = ok && (w.Cw % v == 0);
    if (inner_is_k) ok = ok && (K % v == 0);      // zero-fill past K must be exact
    (void)inner_extent;
    if (ok) break;
    v >>= 1;
  }
  return v;
}

bool win_ok(const TecmWin& w) {
  return !w.enabled || (w.N > 0 && w.Lin > 0 && w.Lout > 0 && w.stride_t > 0 && w.taps > 0 && w.Cw > 0 && w.pad >= 0);
}

}  // namespace

extern "C" int tecm_gemm_f32(const TecmGemm* d, void* stream) {
  TECM_REQUIRE(d != nullptr, TECM_E_ARG, "tecm_gemm_f32: null descriptor");
  const TecmGemm& g = *d;
  TECM_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, TECM_E_ARG, "tecm_gemm_f32: M,N,K must be positive (%lld,%lld,%lld)",
               (long long)g.M, (long long)g.N, (long long)g.K);
  TECM_REQUIRE(g.K < (1ll << 30) && g.N < (1ll << 30), TECM_E_ARG, "tecm_gemm_f32: N/K too large");
  TECM_REQUIRE(g.A && g.B && g.C, TECM_E_ARG, "tecm_gemm_f32: null operand");
  TECM_REQUIRE(win_ok(g.a_win) && win_ok(g.b_win) && win_ok(g.c_win), TECM_E_ARG, "tecm_gemm_f32: bad window view");
  TECM_REQUIRE(!(g.a_win.enabled && g.a_layout != TECM_A_MK), TECM_E_ARG,
               "tecm_gemm_f32: a_win needs the MK layout (view rows are m)");
  TECM_REQUIRE(!(g.b_win.enabled && g.b_layout != TECM_B_KN), TECM_E_ARG,
               "tecm_gemm_f32: b_win needs the KN layout (view rows are the reduction index)");
  TECM_REQUIRE(!(g.a_win.enabled && (int64_t)g.a_win.taps * g.a_win.Cw != g.K), TECM_E_ARG,
               "tecm_gemm_f32: a_win taps*Cw != K");
  TECM_REQUIRE(!(g.b_win.enabled && (int64_t)g.b_win.taps * g.b_win.Cw != g.N), TECM_E_ARG,
               "tecm_gemm_f32: b_win taps*Cw != N");
  TECM_REQUIRE(!(g.c_win.enabled && (int64_t)g.c_win.taps * g.c_win.Cw != g.N), TECM_E_ARG,
               "tecm_gemm_f32: c_win taps*Cw != N");
  TECM_REQUIRE(!(g.rowbias && (g.rb_div <= 0 || g.rb_mod <= 0)), TECM_E_ARG, "tecm_gemm_f32: bad rowbias spec");
  TECM_REQUIRE(!(g.split_k > 1 && g.workspace == nullptr), TECM_E_ARG, "tecm_gemm_f32: split_k needs a workspace");
  TECM_REQUIRE(g.a_layout == TECM_A_MK || g.b_layout == TECM_B_KN, TECM_E_ARG,
               "tecm_gemm_f32: layout combination KM x NK is not built");
  hipStream_t st = (hipStream_t)stream;
  const bool a_inner_k = g.a_layout == TECM_A_MK;
  const bool b_inner_k = g.b_layout == TECM_B_NK;
  const int avec = pick_vec(g.A, g.lda, g.a_win, a_inner_k, g.K, g.M);
  const int bvec = pick_vec(g.B, g.ldb, g.b_win, b_inner_k, g.K, g.N);
  const bool narrow = g.N <= 32;
  if (g.a_layout == TECM_A_MK && g.b_layout == TECM_B_NK)
    return narrow ? dispatch_vec<TECM_A_MK, TECM_B_NK, 32>(g, avec, bvec, st)
                  : dispatch_vec<TECM_A_MK, TECM_B_NK, 128>(g, avec, bvec, st);
  if (g.a_layout == TECM_A_MK && g.b_layout == TECM_B_KN)
    return narrow ? dispatch_vec<TECM_A_MK, TECM_B_KN, 32>(g, avec, bvec, st)
                  : dispatch_vec<TECM_A_MK, TECM_B_KN, 128>(g, avec, bvec, st);
  return narrow ? dispatch_vec<TECM_A_KM, TECM_B_KN, 32>(g, avec, bvec, st)
                : dispatch_vec<TECM_A_KM, TECM_B_KN, 128>(g, avec, bvec, st);
}
