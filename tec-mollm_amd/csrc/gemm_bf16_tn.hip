// bf16 weight-gradient GEMM  C[M x N] = A^T . B  for operands that live in HBM as bf16 in their NATURAL orientation:
// A = [K][M] (the upstream gradient: one row per token / sequence / conv output step), B = [K][N] (the layer's input, plain
// or behind a pad-free temporal-window view), K = 23 288 ... 558 912 rows, M, N <= 2304.  This is every trainable weight
// of the path in bf16 mode (BASELINE configs[2]): the head's two Linears, the patch projection, the strided 1x1 conv of both
// conv blocks, lora_A and lora_B of every GPT-2 layer (reference train.py:85 backward of modules.py:41,116,181,284).
//
// Before (gemm_bf16_kernel<KM, KN>): both tiles go HBM -> registers -> 8x8 register transposes -> LDS [row][k] images ->
// ds_read_b128 fragments, on 256 x 128 output tiles: 1.2-3.3 TB/s on the byte-bound shapes (the 64 x 192 gradient of the
// first 1x1 conv issues 4x its MFMA work as padding), 187 TFLOP/s on the head's 576 x 2304 x 23 288.
// Here nothing is transposed and nothing passes through registers on the way in:
//   * a stage = 32 k-rows of both operands, copied by LDS-DMA (`global_load_lds_dwordx4`) as they lie in HBM: a row of
//     the image is 32 * BM (BN) / 16 bytes of consecutive columns, so every wave-instruction moves whole 128-byte+ runs;
//   * the contraction index is the ROW of both images, which is what `ds_read_b64_tr_b16` (gfx950) is for: a 16-lane group
//     reads 4 rows x 16 columns and each lane receives its column of the 4 rows -- two of them are one operand of
//     v_mfma_f32_32x32x16_bf16 (k = 8h .. 8h+7 of column lane % 32).  No k-permutation: both operands use the same rows;
//   * bank conflicts: a 32-lane half of a transposed read touches 4 rows x 64 B.  The image is linear (the DMA writes
//     wave base + lane * 16), so the 64-byte segment s of row r is stored at segment s ^ f(r) -- f = r & 3 when the row
//     pitch is a multiple of 256 B, (r >> 1) & 1 when it is an odd multiple of 128 B, 0 when it is an odd multiple of 64 B
//     -- applied to the lane's SOURCE address and again to the fragment read address: the 4 rows then fall into the 4
//     quarters of the 64 banks;
//   * a four-slot ring with counted waits (`s_waitcnt vmcnt(P * 2)`: all but the two youngest stages of this wave), one
//     barrier per stage, stage t+3 requested while stage t is multiplied: 50-110 KB in flight per CU;
//   * split-K over blockIdx (XCD-contiguous: the tiles of one K range share an L2), raw partial sums into the slabs the
//     common reducer (gemm.hip: splitk_reduce_kernel) adds in a fixed order and finishes (alpha, accumulate, ...);
//   * 4 waves as WR x WC, FM x FN accumulator tiles each: instantiated for the shapes listed in tn_try below.
// Arithmetic: bf16 operands as stored, fp32 accumulation -- the same products as gemm_bf16_kernel, another summation
// order.  Rows past K in the last stage are copied from the clamped last row and then zeroed in LDS by the block.
#include <type_traits>
#include "gemm_bf16_impl.h"

namespace tecm_gemm16 {

typedef __attribute__((address_space(3))) void tn_lds_void;
typedef const __attribute__((address_space(1))) void tn_glb_void;
typedef __bf16 tn_bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char* tn_lds_ptr;

__device__ __forceinline__ void tn_dma16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((tn_glb_void*)src, (tn_lds_void*)lds_wave_base, 16, 0, 0);
}
// The transposed fragment reads are issued BY HAND (inline asm) and awaited with counted lgkmcnt: through the builtin the
// compiler's wait-count pass treats every LDS read as a possible reader of every LDS-DMA still in flight and puts
// `s_waitcnt vmcnt(0)` in front of the first read of each stage -- the ring would drain once per stage.  LDS operations
// return in order, so "at most n younger reads outstanding" is exact; tn_land ties the registers to the wait.
template <int OFF>
__device__ __forceinline__ void tn_tr_issue(tn_bf16x4& dst, unsigned lds_addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF));
}
template <int NEWER>
__device__ __forceinline__ void tn_wait_lds() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEWER) : "memory");
}
__device__ __forceinline__ void tn_land(tn_bf16x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ bf16x8 tn_join(const tn_bf16x4& lo, const tn_bf16x4& hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int I, int N, typename F>
__device__ __forceinline__ void tn_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    tn_static_for<I + 1, N>(f);
  }
}

constexpr int TN_KC = 32;      // k-rows per stage
constexpr int TN_ST = 4;       // ring slots

// segment swizzle of an image whose rows are W bf16 columns wide (see the header)
template <int W>
__device__ __forceinline__ int tn_swz(int row) {
  constexpr int S = W / 32;                      // 64-byte segments per row
  if constexpr (S % 4 == 0) return row & 3;
  else if constexpr (S % 2 == 0) return (row >> 1) & 1;
  else return 0;
}

struct TnArgs {
  const __bf16* A;             // [K][lda], columns m0.. of the output's rows
  const __bf16* B;             // [K][ldb] or the window view
  float* ws;                   // [splits][M][N]
  int64_t lda, ldb;
  int32_t M, N, K;
  int32_t tiles_m, tiles_n, splits, k_chunk;     // k_chunk % 32 == 0
  TecmWin bw;                  // B's row view (pad == 0, every tap inside the sequence)
};

template <int WR, int WC, int FM, int FN, bool BWIN>
__global__ __launch_bounds__(256, (WR * FM * 32 + WC * FN * 32) * TN_KC * 2 * TN_ST <= 80 * 1024 && FM * FN <= 6 ? 2 : 1)
void gemm_bf16_tn_kernel(const TnArgs a) {
  constexpr int BM = WR * FM * 32, BN = WC * FN * 32;
  constexpr int A_BYTES = TN_KC * BM * 2, B_BYTES = TN_KC * BN * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NIA = A_BYTES / 1024, NIB = B_BYTES / 1024, NI = NIA + NIB;   // wave-instructions per stage
  constexpr int P = (NI + 3) / 4;                                               // per wave (the last ones may be duplicates)
  constexpr int CPRA = BM / 8, CPRB = BN / 8;                                   // 16-byte chunks per image row
  static_assert(WR * WC == 4 && A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "4 waves; whole wave-instructions per operand");
  static_assert(P * (TN_ST - 2) <= 60, "vmcnt range");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[TN_ST * STAGE];

  // block -> (split, tile): XCD-contiguous runs so that the tiles of one K range meet in one L2
  const int nwg = a.tiles_m * a.tiles_n * a.splits;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  const int tiles = a.tiles_m * a.tiles_n;
  const int split = wg / tiles, tile = wg - split * tiles;
  const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int k_begin = split * a.k_chunk;
  const int k_end = min(k_begin + a.k_chunk, a.K);
  const int nst = k_end > k_begin ? (k_end - k_begin + TN_KC - 1) / TN_KC : 0;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / WC, wc = wave % WC;

  // ---- the DMA pieces of this wave: instruction i = wave + 4 j of the stage's NI (duplicates wrap around)
  int dst[P];                  // byte offset of the wave-instruction's 1 KiB inside a stage
  int prow[P];                 // image row of this lane's chunk
  int64_t pcol[P];             // element offset inside the source row (plain view) / window: tap * N * ld + c
  bool pisb[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    int i = wave + 4 * j;
    if (i >= NI) i -= NI;
    const bool isb = i >= NIA;
    const int ii = isb ? i - NIA : i;
    const int c = ii * 64 + lane;                              // chunk index in the operand's linear image
    const int cpr = isb ? CPRB : CPRA;
    const int row = c / cpr, cp = c - row * cpr;               // physical chunk of the row
    const int sw = isb ? tn_swz<BN>(row) : tn_swz<BM>(row);
    const int cl = (((cp >> 2) ^ sw) << 2) | (cp & 3);         // the logical chunk stored there
    dst[j] = (isb ? A_BYTES : 0) + ii * 1024;
    prow[j] = row;
    pisb[j] = isb;
    if (!isb) {
      pcol[j] = m0 + cl * 8;
    } else {
      const int col = n0 + cl * 8;
      if constexpr (BWIN) {
        const int tap = col / a.bw.Cw;
        pcol[j] = (int64_t)tap * a.bw.N * a.ldb + (col - tap * a.bw.Cw);
      } else {
        pcol[j] = col;
      }
    }
  }
  auto src_of = [&](int j, int k) -> const __bf16* {           // k already clamped to [0, K)
    if (!pisb[j]) return a.A + (int64_t)k * a.lda + pcol[j];
    if constexpr (BWIN) {
      const int bt = k / a.bw.N, n = k - bt * a.bw.N;          // k = (b * Lout + t_out) * N + n
      const int b = bt / a.bw.Lout, to = bt - b * a.bw.Lout;
      const int64_t r = ((int64_t)b * a.bw.Lin + (int64_t)to * a.bw.stride_t) * a.bw.N + n;
      return a.B + r * a.ldb + pcol[j];
    } else {
      return a.B + (int64_t)k * a.ldb + pcol[j];
    }
  };
  auto issue_stage = [&](int t) {
    unsigned char* buf = smem + (t % TN_ST) * STAGE;
    const int kb = k_begin + t * TN_KC;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      int k = kb + prow[j];
      k = k < a.K ? k : a.K - 1;                               // the block zeroes these rows once the stage has landed
      tn_dma16(src_of(j, k), buf + dst[j]);
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- fragment addresses: this lane supplies row q of a 4-row group, columns 16 g1 + 4 p .. + 3 of a 32-column block
  const int q = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1, h = lane >> 5;
  const int rbase = 8 * h + q;                                 // + 16 s (k-step) + 4 (second read)
  const int inseg = 32 * g1 + 8 * p;
  // the swizzle only depends on (row & 3) or ((row >> 1) & 1) with row = 16 s + 8 h + 4 u + q: that is q's
  int a_off[FM], b_off[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int seg = (wr * FM + i) ^ tn_swz<BM>(q);
    a_off[i] = rbase * (BM * 2) + seg * 64 + inseg;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int seg = (wc * FN + j) ^ tn_swz<BN>(q);
    b_off[j] = A_BYTES + rbase * (BN * 2) + seg * 64 + inseg;
  }
  const unsigned lds0 = (unsigned)(size_t)(tn_lds_ptr)smem;       // LDS byte address of the ring

  if (nst > 0) {
#pragma unroll
    for (int t = 0; t < TN_ST - 1; ++t)
      if (t < nst) issue_stage(t);
    for (int t = 0; t < nst; ++t) {
      // stage t has landed once only the stages behind it (at most TN_ST - 2 of them) are outstanding
      if (t + TN_ST - 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * (TN_ST - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                            // ... for every wave; and stage t-1's slot is free
      if (t + TN_ST - 1 < nst) issue_stage(t + TN_ST - 1);
      const int slot = t % TN_ST;
      const int valid = k_end - (k_begin + t * TN_KC);         // rows of this stage that exist
      if (valid < TN_KC) {                                     // only the last stage of the last split
        for (int c = threadIdx.x; c < (TN_KC - valid) * (CPRA + CPRB); c += 256) {
          const int ra = c / (CPRA + CPRB), cc = c - ra * (CPRA + CPRB);
          const int row = valid + ra;
          unsigned char* ptr = smem + slot * STAGE + (cc < CPRA ? row * (BM * 2) + cc * 16 : A_BYTES + row * (BN * 2) + (cc - CPRA) * 16);
          *reinterpret_cast<uint4*>(ptr) = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
      }
      // both k-steps' fragments are requested at once; the first step's MFMAs start when only the second step's reads
      // are still outstanding
      const unsigned sb = lds0 + slot * STAGE;
      tn_bf16x4 fa[2][FM][2], fb[2][FN][2];
      tn_static_for<0, 2>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        tn_static_for<0, FM>([&](auto i_) {
          constexpr int i = decltype(i_)::value;
          tn_tr_issue<s * 16 * (BM * 2)>(fa[s][i][0], sb + a_off[i]);
          tn_tr_issue<(s * 16 + 4) * (BM * 2)>(fa[s][i][1], sb + a_off[i]);
        });
        tn_static_for<0, FN>([&](auto j_) {
          constexpr int j = decltype(j_)::value;
          tn_tr_issue<s * 16 * (BN * 2)>(fb[s][j][0], sb + b_off[j]);
          tn_tr_issue<(s * 16 + 4) * (BN * 2)>(fb[s][j][1], sb + b_off[j]);
        });
      });
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 0) tn_wait_lds<2 * (FM + FN)>();
        else tn_wait_lds<0>();
#pragma unroll
        for (int i = 0; i < FM; ++i) { tn_land(fa[s][i][0]); tn_land(fa[s][i][1]); }
#pragma unroll
        for (int j = 0; j < FN; ++j) { tn_land(fb[s][j][0]); tn_land(fb[s][j][1]); }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tn_join(fa[s][i][0], fa[s][i][1]), tn_join(fb[s][j][0], fb[s][j][1]),
                                                                acc[i][j], 0, 0, 0);
        if (s == 0) __builtin_amdgcn_sched_barrier(0);      // the second wait must not be hoisted above these MFMAs
      }
    }
  }
  // ---- raw partial sums of this split: element e of a 32 x 32 tile = row (e & 3) + 8 (e >> 2) + 4 h, column lane % 32
  float* out = a.ws + (int64_t)split * a.M * a.N;
  const int r = lane & 31;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int mb = m0 + (wr * FM + i) * 32, nb = n0 + (wc * FN + j) * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2) + 4 * h;
        out[(int64_t)m * a.N + nb] = acc[i][j][e];
      }
    }
}

template <int WR, int WC, int FM, int FN>
static int tn_launch(const TecmGemm& g, int splits, hipStream_t st) {
  constexpr int BM = WR * FM * 32, BN = WC * FN * 32;
  TnArgs a;
  a.A = reinterpret_cast<const __bf16*>(g.A);
  a.B = reinterpret_cast<const __bf16*>(g.B);
  a.ws = g.workspace;
  a.lda = g.lda; a.ldb = g.ldb;
  a.M = (int32_t)g.M; a.N = (int32_t)g.N; a.K = (int32_t)g.K;
  a.tiles_m = a.M / BM; a.tiles_n = a.N / BN;
  int k_chunk = (int)(((g.K + splits - 1) / splits + TN_KC - 1) / TN_KC) * TN_KC;
  splits = (int)((g.K + k_chunk - 1) / k_chunk);
  a.splits = splits; a.k_chunk = k_chunk;
  a.bw = g.b_win;
  const unsigned grid = (unsigned)(a.tiles_m * a.tiles_n * splits);
  if (g.b_win.enabled)
    hipLaunchKernelGGL((gemm_bf16_tn_kernel<WR, WC, FM, FN, true>), dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((gemm_bf16_tn_kernel<WR, WC, FM, FN, false>), dim3(grid), dim3(256), 0, st, a);
  TECM_CHECK_LAUNCH("tecm_gemm_bf16/tn");
  return splits;
}

// tile geometry for an output shape: index into the instantiation list, or -1
static int tn_geometry(int64_t M, int64_t N) {
  if (M % 192 == 0 && N % 256 == 0) return 0;      // head 576 x 2304, patch 768 x 512 (2 x 2 waves of 96 x 128)
  if (M % 128 == 0 && N % 192 == 0) return 1;      // second 1x1 conv 128 x 384        (2 x 2 waves of 64 x 96)
  if (M % 64 == 0 && N % 192 == 0) return 2;       // first 1x1 conv 64 x 192          (2 x 2 waves of 32 x 96)
  if (M % 256 == 0 && N % 32 == 0 && N < 64) return 3;     // lora_B 2304 x 32          (4 x 1 waves of 64 x 32)
  if (M % 32 == 0 && M < 64 && N % 256 == 0) return 4;     // lora_A 32 x 768           (1 x 4 waves of 32 x 64)
  return -1;
}
static void tn_tile(int geo, int& bm, int& bn) {
  static const int T[5][2] = {{192, 256}, {128, 192}, {64, 192}, {256, 32}, {32, 256}};
  bm = T[geo][0]; bn = T[geo][1];
}

}  // namespace tecm_gemm16

// Split count the TN kernel wants for an output shape (one round of blocks on 256 CUs, two where two blocks fit a CU), or
// 0 when it does not serve the shape.  tecmollm/ops.py:pick_split_k asks; the kernel accepts any count.
extern "C" int32_t tecm_gemm_tn_splits(int64_t M, int64_t N, int64_t K) {
  const int geo = tecm_gemm16::tn_geometry(M, N);
  if (geo < 0 || K < 4096) return 0;
  int bm, bn;
  tecm_gemm16::tn_tile(geo, bm, bn);
  const int64_t tiles = (M / bm) * (N / bn);
  // two blocks per CU where they fit (all geometries but the 192 x 256 one) -- unless the slabs (written once, read once
  // by the reducer) would then cost more than a tenth of the operand bytes
  int64_t s = (geo == 0 ? 256 : 512) / tiles;
  if (s > 256) s = 256;
  if (geo != 0 && 8 * s * M * N > (2 * K * (M + N)) / 10) s = 256 / tiles;
  if (s < 1) s = 1;
  const int64_t maxs = K / 1024 > 1 ? K / 1024 : 1;          // at least 32 stages per block
  if (s > maxs) s = maxs;
  return (int32_t)s;
}

// < 0: error; 0: not served (the caller falls through to gemm_bf16_kernel); > 0: the number of slabs written
int tecm_gemm16_tn_try(const TecmGemm& g, hipStream_t st) {
  using namespace tecm_gemm16;
  const char* env = std::getenv("TECM_BF16_TN");                // "0": keep the register-transposing kernel (A/B, tests)
  if (env && env[0] == '0') return 0;
  const int io = g.io_bf16;
  if (!(io & TECM_IO_A_BF16) || !(io & TECM_IO_B_BF16) || g.a_layout != TECM_A_KM || g.b_layout != TECM_B_KN) return 0;
  if (g.split_k < 2 || !g.workspace || g.a_win.enabled || g.a_drop.p > 0.f || g.b_drop.p > 0.f) return 0;
  if (!tecm_aligned(g.A, 16) || !tecm_aligned(g.B, 16) || g.lda % 8 || g.ldb % 8 || g.K < 4096) return 0;
  const TecmWin& w = g.b_win;
  if (w.enabled) {                                              // pad-free views whose taps all lie inside the sequence
    if (w.pad != 0 || w.Cw % 8 || (int64_t)(w.Lout - 1) * w.stride_t + w.taps > w.Lin) return 0;
    if (g.K % ((int64_t)w.Lout * w.N) != 0) return 0;
  }
  const int geo = tn_geometry(g.M, g.N);
  if (geo < 0) return 0;
  switch (geo) {
    case 0: return tn_launch<2, 2, 3, 4>(g, g.split_k, st);
    case 1: return tn_launch<2, 2, 2, 3>(g, g.split_k, st);
    case 2: return tn_launch<2, 2, 1, 3>(g, g.split_k, st);
    case 3: return tn_launch<4, 1, 2, 1>(g, g.split_k, st);
    default: return tn_launch<1, 4, 1, 2>(g, g.split_k, st);
  }
}
