// bf16 MFMA GEMM for operands that already live in HBM as bf16, both [row][k] (activations written as bf16 by their
// producers, cached bf16 copies of the frozen GPT-2 weights): the plain MK x NK contraction behind 7 of the 8 big
// GPT-2 GEMMs of a layer in bf16 mode (BASELINE configs[2]).  Same descriptor and epilogue as gemm_bf16_impl.h; the
// operand path is a pure copy, so it is done by LDS-DMA and the tile is sized for it:
//
//   * block = 512 threads = 8 waves as 2(m) x 4(n), tile 256 x 256 x 64, ONE block per CU (128 KiB of LDS, up to 256
//     registers per lane).  A wave owns 128 x 64 = 4 x 2 MFMA tiles (v_mfma_f32_32x32x16_bf16, 128 accumulators) and
//     per 16-deep k-step issues 6 ds_read_b128 for 8 MFMAs (the 256 x 128 kernel: 4 for 4) -- 0.75 KiB of LDS reads
//     per MFMA instead of 1 KiB, and 64 KiB of operand per 8.4 MFLOP instead of 48 KiB per 4.2;
//   * global -> LDS by `global_load_lds_dwordx4` (1 KiB per wave-instruction, no VGPR round trip, no ds_write): a
//     K-tile is 64 pieces of 8 rows x 128 B; wave w moves pieces 4w..4w+3 of A and of B.  The DMA writes
//     wave-uniform base + lane * 16, so a tile is stored linearly and the bank-conflict swizzle is applied to the
//     per-lane SOURCE address and again on the fragment read: 16-byte chunk c of row r sits at position
//     c ^ ((r >> 1) & 7) -- the four 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) then touch 16
//     distinct 4-bank groups;
//   * two LDS buffers, one barrier per K-tile: the DMA of tile t+1 is issued before the MFMAs of tile t and retired
//     by the s_waitcnt vmcnt(0) in front of the barrier that ends tile t;
//   * K % 64 == 32 (c_attn with its 32 LoRA columns: K = 800): in the last tile the lanes whose source chunk lies
//     beyond K do not take part in the DMA (it honours EXEC) and store zeros into their slot instead, so the K loop
//     has one body (a second, shorter body makes the compiler copy the 128 accumulators).
//
// Eligibility (tecm_gemm16_dma_try): A and B bf16, plain views, no prologue dropout, split_k <= 1, K % 32 == 0,
// K >= 64, 16-byte friendly operands, float4 epilogue.  Everything else stays on gemm_bf16_kernel.
//
// Measured (round 2, M = 69 864, operands bf16, C fp32; TFLOP/s): N=3072 K=768 574 (register-staged kernel 497),
// N=2304 K=800 502 (460), N=768 K=3072 737 (741), 8192^3 1078 (972).  Ablations of this kernel on N=3072 K=768
// (-DDMA_ABLATE_*, tools/build_variant.py): K loop with neither DMA nor epilogue 260 us (1.27 PFLOP/s), + DMA 328,
// + epilogue 507, both 604 -- the three costs ADD, and the epilogue's 250-280 us for 858 MB of C is twice what a
// plain fill of the same bytes takes (128 us).  Neither an accumulator-direct epilogue (no LDS staging, no barriers),
// nor two co-resident 256 x 128 blocks per CU, nor persistent blocks, nor a bf16 C (half the bytes) moved that term
// (tools/experiments/gemm_bf16_dma_persistent.hip, tools/experiments/README.md).
#include <cstdlib>
#include "gemm_bf16_impl.h"

namespace tecm_gemm16 {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void dma16(const __bf16* src, __bf16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

constexpr int DBM = 256, DBN = 256, DBK = 64, DNTH = 512;

__global__ __launch_bounds__(DNTH, 1) void gemm_bf16_dma_kernel(const TecmGemm g, int tiles_m, int tiles_n) {
  constexpr int WM = 2, WN = 4;
  constexpr int WTM = DBM / WM, WTN = DBN / WN;        // 128 x 64 per wave
  constexpr int MT = WTM / 32, NT = WTN / 32;          // 4 x 2
  constexpr int A_ELEMS = DBM * DBK, B_ELEMS = DBN * DBK, TILE_ELEMS = A_ELEMS + B_ELEMS;   // bf16 elements: 64 KiB
  constexpr int PIECE = 8 * DBK;                       // 8 rows = 1 KiB
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[2 * TILE_ELEMS * 2];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
  static_assert(2 * TILE_ELEMS * 2 >= 8 * 32 * (WTN + 4) * 4, "epilogue slabs fit in the operand buffers");

  // block -> tile map: XCD-contiguous runs, GROUP_M m-tiles per L2 super-tile (as gemm_bf16_kernel)
  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  const int GROUP_M = g._p1 > 0 ? g._p1 : 4;              // m-tiles per L2 super-tile (ops.GROUP_M sweeps it)
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * DBM;
  const int64_t n0 = (int64_t)tn * DBN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int K = (int)g.K;
  const int ntiles = (K + DBK - 1) / DBK;
  const bool ktail = (K % DBK) != 0;                   // K % 64 == 32

  // ---- per-lane DMA sources: piece p = 4 * wave + i covers tile rows 8p .. 8p+7; lane -> (row 8p + lane/8, position
  // lane % 8); the chunk fetched into that position is position ^ swizzle(row)
  const __bf16* Ah = reinterpret_cast<const __bf16*>(g.A);
  const __bf16* Bh = reinterpret_cast<const __bf16*>(g.B);
  const __bf16* asrc[4];
  const __bf16* bsrc[4];
  int chunk_of[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    chunk_of[i] = chunk;
    int64_t gm = m0 + row;
    gm = gm < g.M ? gm : g.M - 1;                      // clamped rows feed accumulator rows that are never stored
    int64_t gn = n0 + row;
    gn = gn < g.N ? gn : g.N - 1;
    if (g.a_win.enabled) {                             // pad-free window view (host-checked): tap 0 of row (bq, t_out, n)
      const int64_t bt = gm / g.a_win.N, n = gm - bt * g.a_win.N;
      const int64_t bq = bt / g.a_win.Lout, to = bt - bq * g.a_win.Lout;
      gm = (bq * g.a_win.Lin + to * g.a_win.stride_t) * g.a_win.N + n;
    }
    asrc[i] = Ah + gm * g.lda + chunk * 8;
    bsrc[i] = Bh + gn * g.ldb + chunk * 8;
  }
  // a window view's K runs tap by tap: after Cw / 64 K-tiles the source row moves one time step (N rows) on
  const int tiles_per_tap = g.a_win.enabled ? g.a_win.Cw / DBK : 0x7fffffff;
  const int64_t tap_jump = g.a_win.enabled ? (int64_t)g.a_win.N * g.lda - g.a_win.Cw : 0;
  int tap_tile = 0;
  auto issue_tile = [&](__bf16* buf, bool tail) {
    __bf16* a_dst = buf + (wave * 4) * PIECE;
    __bf16* b_dst = buf + A_ELEMS + (wave * 4) * PIECE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!tail || chunk_of[i] < 4) {
        dma16(asrc[i], a_dst + i * PIECE);
        dma16(bsrc[i], b_dst + i * PIECE);
      } else {                                         // beyond K: the slot the DMA would have filled holds zeros
        *reinterpret_cast<uint4*>(a_dst + i * PIECE + lane * 8) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(b_dst + i * PIECE + lane * 8) = make_uint4(0u, 0u, 0u, 0u);
      }
      asrc[i] += DBK;
      bsrc[i] += DBK;
    }
    if (++tap_tile == tiles_per_tap) {
      tap_tile = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) asrc[i] += tap_jump;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addresses (bf16 elements within a tile buffer) and row swizzles
  int a_off[MT], a_sw[MT], b_off[NT], b_sw[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * WTM + i * 32 + r;
    a_off[i] = row * DBK;
    a_sw[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * WTN + j * 32 + r;
    b_off[j] = A_ELEMS + row * DBK;
    b_sw[j] = (row >> 1) & 7;
  }
  auto read_frags = [&](const __bf16* T, int s, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) {
    const int c = 2 * s + h;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(T + a_off[i] + ((c ^ a_sw[i]) << 3));
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(T + b_off[j] + ((c ^ b_sw[j]) << 3));
  };
  auto do_mfma = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  };

  issue_tile(smem, ktail && ntiles == 1);
  __syncthreads();                                      // vmcnt(0) + barrier: tile 0 has landed

  bf16x8 fa[2][MT], fb[2][NT];
  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const __bf16* Tc = smem + cur * TILE_ELEMS;
#ifndef DMA_ABLATE_NOLOAD
    if (t + 1 < ntiles) issue_tile(smem + (cur ^ 1) * TILE_ELEMS, ktail && t + 2 == ntiles);
#endif
    read_frags(Tc, 0, fa[0], fb[0]);
    static_for<4>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
#ifdef DMA_ABLATE_HALFREADS                              // diagnostics: fragments of every other k-step are not read (wrong results)
      if constexpr (s == 1) read_frags(Tc, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
      if constexpr (s == 0 || s == 2) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[(s + 1) & 1][i] = fa[s & 1][i];
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[(s + 1) & 1][j] = fb[s & 1][j];
      }
#else
      if constexpr (s < 3) read_frags(Tc, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#endif
      do_mfma(fa[s & 1], fb[s & 1]);
      // interleave: one LDS read behind each of the first six MFMAs of the step
#pragma unroll
      for (int m = 0; m < MT * NT; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  // 1 MFMA
        if (s < 3 && m < MT + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // 1 DS read
      }
    });
#ifndef DMA_ABLATE_NOBARRIER                             // diagnostics: no barrier per K-tile (racy: timing only)
    __syncthreads();                                    // DMA of tile t+1 retired, every wave done with tile t
#endif
    cur ^= 1;
  }

#ifdef DMA_ABLATE_NOEPI                                  // diagnostics (tools/build_variant.py): K loop without the epilogue
  float keep = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) keep += acc[i][j][e];
  if (keep == 12345.678f) reinterpret_cast<float*>(g.C)[0] = keep;
#else
  // non-temporal stores: C and the pre-activation stream past L2 instead of evicting the operand panels the next
  // tiles of this XCD are about to read (N=3072 K=768: 512 -> 476 us, N=2304 K=800: 409 -> 367; no effect on the fp32
  // kernel, whose matrix time hides it)
  block_epilogue16<MT, NT, WTM, WTN, true>(g, acc, smem_raw, wave, lane, wm, wn, m0, n0);
#endif
}


// ------------------------------------------------------------------------------------------------------------------
// Second geometry, for matrices whose width the 256-column tile quantises badly (N = 800, the dX of c_attn: four
// n-tiles of which the last is 32 / 256 wide): 256 x 128 x 32 tiles, 8 waves as 4(m) x 2(n) of 64 x 64 (2 x 2 MFMA tiles,
// 64 accumulators), a THREE-slot LDS ring of 24 KiB K-tiles (72 KiB) and at most 128 registers per lane, so that TWO
// blocks = 16 waves share a CU.  Measured against the 256 x 256 kernel (M = 69 864, us per launch, round 3,
// tools/dma_ab.sh): N=800 K=2304 375 vs 402 (-7 %), N=2304 K=800 374 vs 381, N=768 K=768 136 vs 138, fc1 form 680 vs
// 683, GELU' form 797 vs 729, N=768 K=3072 491 vs 437, 8192^3 1196 vs 1006 -- it only pays where it removes column
// waste, so the dispatcher uses it for N % 256 in [1, 128] and nothing else.  What it does NOT do is hide one block's
// epilogue under the other block's K loop (the reason it was built): two co-resident blocks started together run in
// lockstep, and neither a start-up stagger of the odd block of a CU (per-CU arrival tickets), nor persistent blocks,
// nor pacing the epilogue's stores with s_sleep changed a launch by more than noise (tools/experiments/README.md).
//   * K-tile = 256 + 128 rows of 64 B: 24 DMA pieces of 16 rows x 64 B, three per wave; lane -> (row 16p + lane/4,
//     position lane%4), the chunk fetched into a position is position ^ ((row >> 2) & 3) -- with 64-byte rows the four
//     16-lane groups of a ds_read_b128 then cover all 64 banks once;
//   * ring: tiles t+1 and t+2 are in flight while tile t is multiplied; `s_waitcnt vmcnt(3)` (the three youngest DMAs
//     = tile t+1) + one barrier per K-tile; the slot refilled at step t was last read at step t-1, which every wave has
//     left when it passes the barrier of step t.
constexpr int D2M = 256, D2N = 128, D2K = 32, D2TH = 512, D2ST = 3;

__global__ __launch_bounds__(D2TH, 4) void gemm_bf16_dma2_kernel(const TecmGemm g, int tiles_m, int tiles_n) {
  constexpr int WM = 4, WN = 2;
  constexpr int WTM = D2M / WM, WTN = D2N / WN;        // 64 x 64 per wave
  constexpr int MT = WTM / 32, NT = WTN / 32;          // 2 x 2
  constexpr int A_ELEMS = D2M * D2K, B_ELEMS = D2N * D2K, TILE_ELEMS = A_ELEMS + B_ELEMS;   // 24 KiB
  constexpr int PIECE = 16 * D2K;                      // 16 rows of 64 B = 1 KiB
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[D2ST * TILE_ELEMS * 2];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
  static_assert(D2ST * TILE_ELEMS * 2 >= 8 * 32 * (WTN + 4) * 4, "epilogue slabs fit in the operand ring");

  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  const int GROUP_M = g._p1 > 0 ? g._p1 : 4;
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * D2M;
  const int64_t n0 = (int64_t)tn * D2N;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int ntiles = (int)g.K / D2K;                   // K % 32 == 0 (host)

  // pieces 0..15 = A rows 16p.., pieces 16..23 = B rows; wave w moves pieces w, w + 8, w + 16
  const __bf16* Ah = reinterpret_cast<const __bf16*>(g.A);
  const __bf16* Bh = reinterpret_cast<const __bf16*>(g.B);
  const __bf16* src[3];
  int dst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = wave + 8 * i;
    const bool isb = p >= 16;
    const int row = (isb ? p - 16 : p) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    if (!isb) {
      int64_t gm = m0 + row;
      gm = gm < g.M ? gm : g.M - 1;                    // clamped rows feed accumulator rows that are never stored
      src[i] = Ah + gm * g.lda + chunk * 8;
      dst[i] = p * PIECE;
    } else {
      int64_t gn = n0 + row;
      gn = gn < g.N ? gn : g.N - 1;
      src[i] = Bh + gn * g.ldb + chunk * 8;
      dst[i] = A_ELEMS + (p - 16) * PIECE;
    }
  }
  auto issue_tile = [&](int slot) {
    __bf16* buf = smem + slot * TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dma16(src[i], buf + dst[i]);
      src[i] += D2K;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int a_off[MT], b_off[NT], a_sw[MT], b_sw[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * WTM + i * 32 + r;
    a_off[i] = row * D2K;
    a_sw[i] = (row >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * WTN + j * 32 + r;
    b_off[j] = A_ELEMS + row * D2K;
    b_sw[j] = (row >> 2) & 3;
  }
  auto read_frags = [&](const __bf16* T, int s_, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) {
    const int c = 2 * s_ + h;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(T + a_off[i] + ((c ^ a_sw[i]) << 3));
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(T + b_off[j] + ((c ^ b_sw[j]) << 3));
  };
  auto do_mfma = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  };

  issue_tile(0);
  if (ntiles > 1) issue_tile(1);
  int slot = 0;
  bf16x8 fa[2][MT], fb[2][NT];
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed once at most the DMAs of tile t+1 (the three youngest) are outstanding
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < ntiles) issue_tile(slot == 0 ? 2 : slot - 1);      // slot (t + 2) % 3
    const __bf16* Tc = smem + slot * TILE_ELEMS;
    read_frags(Tc, 0, fa[0], fb[0]);
    read_frags(Tc, 1, fa[1], fb[1]);
    do_mfma(fa[0], fb[0]);
    do_mfma(fa[1], fb[1]);
    slot = slot == 2 ? 0 : slot + 1;
  }
  __syncthreads();                                      // every wave has left the last K-tile: the ring becomes staging
  block_epilogue16<MT, NT, WTM, WTN, true, true>(g, acc, smem_raw, wave, lane, wm, wn, m0, n0);
}


// ------------------------------------------------------------------------------------------------------------------
// Third geometry (round 3): the 256 x 256 tile of the first kernel on a FOUR-slot ring of 32-deep K-tiles (4 x 32 KiB),
// because what the two-slot kernel waits for is its own operand DMA: a tile is requested one K-tile (4 k-steps, ~0.9 us
// of MFMA) before it is needed, which is the latency of an L2 hit under load and less than that of a miss.  Here
//   * tile t+3 is requested while tile t is multiplied (three K-tiles = ~1.4 us ahead), waits are counted
//     (`s_waitcnt vmcnt(4)`: everything but the youngest tile's four DMAs of this wave);
//   * the barrier that publishes tile t+1 sits BETWEEN the two 16-deep k-steps of tile t, and the fragments of tile t+1's
//     first k-step are read behind it, under the MFMAs of tile t's second k-step: no LDS read latency is exposed after a
//     barrier, and the slot the new DMA overwrites (tile t-1's) was left by every wave a whole k-step ago.
// 64-byte LDS rows, swizzle and piece layout as the second geometry; 8 waves as 2(m) x 4(n) of 128 x 64 as the first.
constexpr int D3M = 256, D3N = 256, D3K = 32, D3TH = 512, D3ST = 4;

__global__ __launch_bounds__(D3TH, 1) void gemm_bf16_dma3_kernel(const TecmGemm g, int tiles_m, int tiles_n) {
  constexpr int WM = 2, WN = 4;
  constexpr int WTM = D3M / WM, WTN = D3N / WN;        // 128 x 64 per wave
  constexpr int MT = WTM / 32, NT = WTN / 32;          // 4 x 2
  constexpr int A_ELEMS = D3M * D3K, B_ELEMS = D3N * D3K, TILE_ELEMS = A_ELEMS + B_ELEMS;   // 32 KiB
  constexpr int PIECE = 16 * D3K;                      // 16 rows of 64 B = 1 KiB
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[D3ST * TILE_ELEMS * 2];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
  static_assert(D3ST * TILE_ELEMS * 2 >= 8 * 32 * (WTN + 4) * 4, "epilogue slabs fit in the operand ring");

  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  const int GROUP_M = g._p1 > 0 ? g._p1 : 4;
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * D3M;
  const int64_t n0 = (int64_t)tn * D3N;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int ntiles = (int)g.K / D3K;                   // K % 32 == 0 (host)

  // pieces 0..15 = A rows 16p.., 16..31 = B rows; wave w moves pieces w, w + 8 (A) and 16 + w, 24 + w (B)
  const __bf16* Ah = reinterpret_cast<const __bf16*>(g.A);
  const __bf16* Bh = reinterpret_cast<const __bf16*>(g.B);
  const __bf16* src[4];
  int dst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wave + 8 * i;
    const bool isb = p >= 16;
    const int row = (isb ? p - 16 : p) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    if (!isb) {
      int64_t gm = m0 + row;
      gm = gm < g.M ? gm : g.M - 1;                    // clamped rows feed accumulator rows that are never stored
      src[i] = Ah + gm * g.lda + chunk * 8;
      dst[i] = p * PIECE;
    } else {
      int64_t gn = n0 + row;
      gn = gn < g.N ? gn : g.N - 1;
      src[i] = Bh + gn * g.ldb + chunk * 8;
      dst[i] = A_ELEMS + (p - 16) * PIECE;
    }
  }
  auto issue_tile = [&](int slot) {
    __bf16* buf = smem + slot * TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16(src[i], buf + dst[i]);
      src[i] += D3K;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int a_off[MT], b_off[NT], a_sw[MT], b_sw[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * WTM + i * 32 + r;
    a_off[i] = row * D3K;
    a_sw[i] = (row >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * WTN + j * 32 + r;
    b_off[j] = A_ELEMS + row * D3K;
    b_sw[j] = (row >> 2) & 3;
  }
  auto read_frags = [&](const __bf16* T, int s_, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) {
    const int c = 2 * s_ + h;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(T + a_off[i] + ((c ^ a_sw[i]) << 3));
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(T + b_off[j] + ((c ^ b_sw[j]) << 3));
  };
  auto do_mfma = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  };
  auto interleave = [&]() {                            // one LDS read behind each of the first six MFMAs of a k-step
#pragma unroll
    for (int m = 0; m < MT * NT; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (m < MT + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };

  issue_tile(0);
  if (ntiles > 1) issue_tile(1);
  if (ntiles > 2) issue_tile(2);
  // tile 0 has landed once at most the DMAs of the tiles requested after it are outstanding
  if (ntiles > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (ntiles > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 fa[2][MT], fb[2][NT];
  read_frags(smem, 0, fa[0], fb[0]);
  int slot = 0;
  for (int t = 0; t < ntiles; ++t) {
    const __bf16* Tc = smem + slot * TILE_ELEMS;
    const int nslot = (slot + 1) & 3;
    read_frags(Tc, 1, fa[1], fb[1]);
    do_mfma(fa[0], fb[0]);
    interleave();
    if (t + 1 < ntiles) {
      // tile t+1 has landed once at most tile t+2's four DMAs of this wave are outstanding
      if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 3 < ntiles) issue_tile((slot + 3) & 3);  // the slot of tile t-1: every wave left it a k-step ago
      read_frags(smem + nslot * TILE_ELEMS, 0, fa[0], fb[0]);
    }
    do_mfma(fa[1], fb[1]);
    if (t + 1 < ntiles) interleave();
    slot = nslot;
  }
  __syncthreads();                                      // every wave has left the last K-tile: the ring becomes staging
  block_epilogue16<MT, NT, WTM, WTN, true>(g, acc, smem_raw, wave, lane, wm, wn, m0, n0);
}


// ------------------------------------------------------------------------------------------------------------------
// Fourth geometry (round 3): the four-slot ring above with the two waves of every SIMD in ANTI-PHASE (see the loop).

__global__ __launch_bounds__(D3TH, 1) void gemm_bf16_dma4_kernel(const TecmGemm g, int tiles_m, int tiles_n) {
  constexpr int WM = 2, WN = 4;
  constexpr int WTM = D3M / WM, WTN = D3N / WN;        // 128 x 64 per wave
  constexpr int MT = WTM / 32, NT = WTN / 32;          // 4 x 2
  constexpr int A_ELEMS = D3M * D3K, B_ELEMS = D3N * D3K, TILE_ELEMS = A_ELEMS + B_ELEMS;   // 32 KiB
  constexpr int PIECE = 16 * D3K;                      // 16 rows of 64 B = 1 KiB
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[D3ST * TILE_ELEMS * 2];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
  static_assert(D3ST * TILE_ELEMS * 2 >= 8 * 32 * (WTN + 4) * 4, "epilogue slabs fit in the operand ring");

  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  const int GROUP_M = g._p1 > 0 ? g._p1 : 4;
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * D3M;
  const int64_t n0 = (int64_t)tn * D3N;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int ntiles = (int)g.K / D3K;                   // K % 32 == 0 (host)

  // pieces 0..15 = A rows 16p.., 16..31 = B rows; wave w moves pieces w, w + 8 (A) and 16 + w, 24 + w (B)
  const __bf16* Ah = reinterpret_cast<const __bf16*>(g.A);
  const __bf16* Bh = reinterpret_cast<const __bf16*>(g.B);
  const __bf16* src[4];
  int dst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wave + 8 * i;
    const bool isb = p >= 16;
    const int row = (isb ? p - 16 : p) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    if (!isb) {
      int64_t gm = m0 + row;
      gm = gm < g.M ? gm : g.M - 1;                    // clamped rows feed accumulator rows that are never stored
      src[i] = Ah + gm * g.lda + chunk * 8;
      dst[i] = p * PIECE;
    } else {
      int64_t gn = n0 + row;
      gn = gn < g.N ? gn : g.N - 1;
      src[i] = Bh + gn * g.ldb + chunk * 8;
      dst[i] = A_ELEMS + (p - 16) * PIECE;
    }
  }
  auto issue_tile = [&](int slot) {
    __bf16* buf = smem + slot * TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16(src[i], buf + dst[i]);
      src[i] += D3K;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int a_off[MT], b_off[NT], a_sw[MT], b_sw[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * WTM + i * 32 + r;
    a_off[i] = row * D3K;
    a_sw[i] = (row >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * WTN + j * 32 + r;
    b_off[j] = A_ELEMS + row * D3K;
    b_sw[j] = (row >> 2) & 3;
  }
  auto read_frags = [&](const __bf16* T, int s_, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) {
    const int c = 2 * s_ + h;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(T + a_off[i] + ((c ^ a_sw[i]) << 3));
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(T + b_off[j] + ((c ^ b_sw[j]) << 3));
  };
  auto do_mfma = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  };
  auto interleave = [&]() {                            // one LDS read behind each of the first six MFMAs of a k-step
#pragma unroll
    for (int m = 0; m < MT * NT; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (m < MT + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };

  // ---- two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run the same program ONE BARRIER apart: while one
  // group multiplies (8 MFMAs = 256 cycles per SIMD, s_setprio 1) the other reads its next fragments from LDS, requests
  // the next tile and waits at the barrier -- the matrix pipe of a SIMD always has one wave feeding it.  A K-tile is two
  // such phases (accumulator rows 0-1, then 2-3; the B fragments are read once per K-tile), four barriers.
  //   * publish: every wave retires its own DMAs of tile t+1 (counted vmcnt) before its third barrier of tile t; the late
  //     group does so one barrier later, and the early group's first read of tile t+1 sits behind exactly that barrier;
  //   * recycle: tile t+3 goes into tile t-1's slot after the third barrier of tile t -- the late group issued its last
  //     reads of tile t-1 five barriers earlier.
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  issue_tile(0);
  if (ntiles > 1) issue_tile(1);
  if (ntiles > 2) issue_tile(2);
  if (ntiles > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (ntiles > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                         // tile 0 is visible
  if (grp) __builtin_amdgcn_s_barrier();                // the stagger
  bf16x8 fa[2][2], fb[2][NT];
  int slot = 0;
  for (int t = 0; t < ntiles; ++t) {
    const __bf16* Tc = smem + slot * TILE_ELEMS;
    // ---- phase 0: B fragments of the K-tile, A fragments of accumulator rows 0-1
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int c = 2 * s_ + h;
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[s_][j] = *reinterpret_cast<const bf16x8*>(Tc + b_off[j] + ((c ^ b_sw[j]) << 3));
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[s_][i] = *reinterpret_cast<const bf16x8*>(Tc + a_off[i] + ((c ^ a_sw[i]) << 3));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i], fb[s_][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: A fragments of accumulator rows 2-3
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int c = 2 * s_ + h;
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[s_][i] = *reinterpret_cast<const bf16x8*>(Tc + a_off[2 + i] + ((c ^ a_sw[2 + i]) << 3));
    }
    if (t + 1 < ntiles) {                               // this wave's DMAs of tile t+1 have landed
      if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (t + 3 < ntiles) issue_tile((slot + 3) & 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i], fb[s_][j], acc[2 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    slot = (slot + 1) & 3;
  }
  if (!grp) __builtin_amdgcn_s_barrier();               // the early group waits for the late one
  __syncthreads();                                      // every wave has left the last K-tile: the ring becomes staging
  block_epilogue16<MT, NT, WTM, WTN, true>(g, acc, smem_raw, wave, lane, wm, wn, m0, n0);
}


// ------------------------------------------------------------------------------------------------------------------
// Fifth geometry (round 3): the four-slot ring with v_mfma_f32_16x16x32_bf16.  A bare register-only loop of that shape
// sustains 2.08 PFLOP/s on random operands against 1.87 for 32x32x16 (tools/micro/mfma_bf16_shapes.hip: the chip's power
// management sets the ceiling, and the 16x16 shape holds a higher clock), and it is the one variable the other four
// geometries share.  Wave tile 128 x 64 = 8 x 4 tiles of 16 x 16 (128 accumulator registers, as before); a fragment is
// ONE ds_read_b128 of 16 rows x 64 B: lane l reads row l & 15, 16-byte chunk l >> 4, which is exactly the operand map
// (A[row l&15][k = 8 (l>>4) + j]).  With 64-byte rows the four lane groups of a ds_read_b128 cover all 64 banks once
// when chunk c of row r sits at position c ^ (3 * ((r >> 3) & 1)).  Per K-tile (32 deep): 12 fragment reads, 32 MFMAs,
// in two halves (accumulator rows 0-3 / 4-7) with the publishing barrier between them as in the third geometry.
// The summation order inside an MFMA differs from the 32x32x16 kernels': results agree to fp32 rounding, not bit for bit.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// (Round 4: the body is a template on the tile height BM.  BM = 288 -- nine 16-row MFMA tiles per wave instead of eight --
//  exists for ONE reason: M = 69 864 rows are 273 m-tiles of 256, and with N = 768 (three n-tiles) that is 819 tiles = 3.2
//  rounds of 256 CUs: the fourth round runs 51 tiles on 256 CUs and costs a full tile time (measured: M = 65 536 332 us,
//  M = 69 864 407 us at K = 3072).  243 m-tiles of 288 are 729 tiles = three rounds of 1.125x the work: 3.375 against 4.
//  The host picks per shape: rounds(BM) * BM, smaller wins -- tecm_gemm16_dma_try.)
template <int BM>
__device__ __forceinline__ void gemm_bf16_dma5_body(const TecmGemm& g, int tiles_m, int tiles_n, unsigned char* smem_raw) {
  constexpr int WM = 2, WN = 4;
  constexpr int WTM = BM / WM, WTN = D3N / WN;         // 128 (144) x 64 per wave
  constexpr int MT = WTM / 16, NT = WTN / 16;          // 8 (9) x 4 tiles of 16 x 16
  constexpr int H0 = (MT + 1) / 2, H1 = MT - H0;       // row tiles of the two halves of a K-tile
  constexpr int A_ELEMS = BM * D3K, B_ELEMS = D3N * D3K, TILE_ELEMS = A_ELEMS + B_ELEMS;   // 32 (34) KiB
  constexpr int PIECE = 16 * D3K;                      // 16 rows of 64 B = 1 KiB
  constexpr int NPA = BM / 16, NPB = D3N / 16, NP = NPA + NPB, PW = (NP + 7) / 8;   // DMA pieces: per K-tile, per wave
  static_assert(BM % 32 == 0 && H1 >= 1 && H0 <= 5, "two waves of whole 16-row tiles");
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
  static_assert(D3ST * TILE_ELEMS * 2 >= 8 * 32 * (WTN + 4) * 4, "epilogue slabs fit in the operand ring");

  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  const int GROUP_M = g._p1 > 0 ? g._p1 : 4;
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * BM;
  const int64_t n0 = (int64_t)tn * D3N;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntiles = (int)g.K / D3K;                   // K % 32 == 0 (host)

  const __bf16* Ah = reinterpret_cast<const __bf16*>(g.A);
  const __bf16* Bh = reinterpret_cast<const __bf16*>(g.B);
  const __bf16* src[PW];
  int dst[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    int p = wave + 8 * i;                              // pieces 0..NPA-1 = A rows 16p.., then the B rows
    if (p >= NP) p -= NP;                              // (never issued: see `full` below; kept in range for the address math)
    const bool isb = p >= NPA;
    const int row = (isb ? p - NPA : p) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ (3 * ((row >> 3) & 1));
    if (!isb) {
      int64_t gm = m0 + row;
      gm = gm < g.M ? gm : g.M - 1;
      src[i] = Ah + gm * g.lda + chunk * 8;
      dst[i] = p * PIECE;
    } else {
      int64_t gn = n0 + row;
      gn = gn < g.N ? gn : g.N - 1;
      src[i] = Bh + gn * g.ldb + chunk * 8;
      dst[i] = A_ELEMS + (p - NPA) * PIECE;
    }
  }
  // the first NP - 8 (PW - 1) waves move PW pieces per K-tile, the others PW - 1: their counted waits differ accordingly
  const bool full = (NP % 8 == 0) || wave < NP - 8 * (PW - 1);
  auto wait_tiles = [&](auto kc) {                       // all but the youngest K tiles of this wave's DMAs have landed
    constexpr int KT = decltype(kc)::value;
    if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KT * PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KT * (PW - 1)) : "memory");
  };
  auto issue_tile = [&](int slot) {
    __bf16* buf = smem + slot * TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      if (i + 1 < PW || full) dma16(src[i], buf + dst[i]);   // wave-uniform: NP is not a multiple of 8 for the 288-row tile
      src[i] += D3K;
    }
  };

  f32x4v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // fragment addresses: row = tile row 16 i + (lane & 15), chunk lane >> 4 at its swizzled position
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[MT], b_off[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * WTM + i * 16 + fr;
    a_off[i] = row * D3K + ((fq ^ (3 * ((row >> 3) & 1))) << 3);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * WTN + j * 16 + fr;
    b_off[j] = A_ELEMS + row * D3K + ((fq ^ (3 * ((row >> 3) & 1))) << 3);
  }
  auto mfma_half = [&](const bf16x8 (&af)[H0], const bf16x8 (&bf)[NT], auto half) {
    constexpr int HB = decltype(half)::value ? H0 : 0, HN = decltype(half)::value ? H1 : H0;
#pragma unroll
    for (int i = 0; i < HN; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[HB + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[HB + i][j], 0, 0, 0);
  };

  issue_tile(0);
  if (ntiles > 1) issue_tile(1);
  if (ntiles > 2) issue_tile(2);
  if (ntiles > 2) wait_tiles(std::integral_constant<int, 2>{});
  else if (ntiles > 1) wait_tiles(std::integral_constant<int, 1>{});
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 fa[2][H0], fb[2][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(smem + b_off[j]);
#pragma unroll
  for (int i = 0; i < H0; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(smem + a_off[i]);
  int slot = 0;
  // two K-tiles per iteration so that the B-fragment buffers alternate with compile-time indices
  auto ktile = [&](int t, auto parity) {
    constexpr int P = decltype(parity)::value;
    const __bf16* Tc = smem + slot * TILE_ELEMS;
    const int nslot = (slot + 1) & 3;
#pragma unroll
    for (int i = 0; i < H1; ++i) fa[1][i] = *reinterpret_cast<const bf16x8*>(Tc + a_off[H0 + i]);
    mfma_half(fa[0], fb[P], std::integral_constant<int, 0>{});
    if (t + 1 < ntiles) {
      if (t + 2 < ntiles) wait_tiles(std::integral_constant<int, 1>{});
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 3 < ntiles) issue_tile((slot + 3) & 3);  // the slot of tile t-1
      const __bf16* Tn = smem + nslot * TILE_ELEMS;
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[P ^ 1][j] = *reinterpret_cast<const bf16x8*>(Tn + b_off[j]);
#pragma unroll
      for (int i = 0; i < H0; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(Tn + a_off[i]);
    }
    mfma_half(fa[1], fb[P], std::integral_constant<int, 1>{});
    slot = nslot;
  };
  for (int t = 0; t < ntiles; t += 2) {
    ktile(t, std::integral_constant<int, 0>{});
    if (t + 1 < ntiles) ktile(t + 1, std::integral_constant<int, 1>{});
  }
  __syncthreads();                                      // every wave has left the last K-tile: the ring becomes staging
#ifdef DMA_ABLATE_NOEPI                                  // diagnostics (tools/build_variant.py): K loop without the epilogue
  {
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) keep += acc[i][j][e];
    if (keep == 12345.678f) reinterpret_cast<float*>(g.C)[0] = keep;
    return;
  }
#endif

  // ---- epilogue: gemm_impl.h's straight-line form over this wave's private staging rows; only the parking of the
  // accumulators knows the 16x16 C/D map (row = 4 (lane >> 4) + reg, col = lane & 15)
  constexpr int STG_LD = WTN + 4;
  const DropCtx odc = make_drop(g.out_drop);
  float* stg = reinterpret_cast<float*>(smem_raw) + wave * (32 * STG_LD);
  auto stage_slab = [&](auto ic) {                     // 32 accumulator rows = tile rows 2 i, 2 i + 1
    constexpr int i = decltype(ic)::value;
    static_for<2>([&](auto tc) {
      constexpr int ti = decltype(tc)::value;
      static_for<NT>([&](auto jc) {
        constexpr int jn = decltype(jc)::value;
        static_for<4>([&](auto ec) {
          constexpr int e = decltype(ec)::value;
          stg[(16 * ti + 4 * fq + e) * STG_LD + jn * 16 + fr] = acc[2 * i + ti][jn][e];
        });
      });
    });
  };
  const int fmode = tecm_gemm::epi_fast_mode(g);         // >= 0: checked on the host (tecm_gemm16_dma_try)
  constexpr int LPR = WTN / 4, RPI = 64 / LPR;
  const int lcol = (lane % LPR) * 4, lrow = lane / LPR;
  const EpiCol ecol = epi_col(g, n0 + wn * WTN + lcol);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + (ecol.ok ? ecol.n : 0));   // (lanes outside the matrix: column 0, never stored)
  tecm_gemm::epi_fast_dispatch<MT / 2, 32 / RPI, RPI, STG_LD, true>(fmode, g, odc, stg, lrow, lcol, m0 + wm * WTM, ecol, bias4,
                                                                stage_slab);
  if constexpr (MT % 2 == 1) {                           // the ninth row tile: a half slab of 16 rows
    auto stage_last = [&](auto) {
      static_for<NT>([&](auto jc) {
        constexpr int jn = decltype(jc)::value;
        static_for<4>([&](auto ec) {
          constexpr int e = decltype(ec)::value;
          stg[(4 * fq + e) * STG_LD + jn * 16 + fr] = acc[MT - 1][jn][e];
        });
      });
    };
    tecm_gemm::epi_fast_dispatch<1, 16 / RPI, RPI, STG_LD, true>(fmode, g, odc, stg, lrow, lcol, m0 + wm * WTM + (MT - 1) * 16, ecol,
                                                              bias4, stage_last);
  }
}

__global__ __launch_bounds__(D3TH, 1) void gemm_bf16_dma5_kernel(const TecmGemm g, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[D3ST * (D3M + D3N) * D3K * 2];
  gemm_bf16_dma5_body<D3M>(g, tiles_m, tiles_n, smem_raw);
}
// the same on 288-row tiles (see above): 139 264 B of LDS
__global__ __launch_bounds__(D3TH, 1) void gemm_bf16_dma5w_kernel(const TecmGemm g, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[D3ST * (288 + D3N) * D3K * 2];
  gemm_bf16_dma5_body<288>(g, tiles_m, tiles_n, smem_raw);
}


// ------------------------------------------------------------------------------------------------------------------
// Sixth geometry (round 4): FOUR waves per block, TWO blocks per CU.  The fifth geometry's wave tile (128 x 64 = 8 x 4
// tiles of v_mfma_f32_16x16x32_bf16, 128 accumulators, up to 256 registers) in a 256-thread block: block tile
// BM x BN = (WM * 128) x (WN * 64) with WM * WN = 4, a THREE-slot ring of 32-deep K-tiles ((BM + BN) * 64 B = 24 KiB per
// slot, 72 KiB per block).  A CU then holds two INDEPENDENT blocks, one wave of each per SIMD: while one block stores its
// tile (the epilogue moves as many bytes as the K loop of a K = 768 GEMM takes to compute), the other block's waves own
// the matrix pipes; inside the K loop the two waves of a SIMD are uncorrelated instead of meeting at the same barrier.
// Ring protocol (per wave 6 DMA pieces per K-tile): in the middle of tile t every wave has issued ALL its fragment reads of
// tile t (the second half's A fragments are read at the top), waits for them (`lgkmcnt(0)`) and for its own pieces of tile
// t+1 (`vmcnt(6)`: at most tile t+2's six outstanding), passes the barrier, requests tile t+3 into tile t's slot, and reads
// tile t+1's first fragments under the second half's 16 MFMAs.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm_bf16_dma6_kernel(const TecmGemm g, int tiles_m, int tiles_n) {
  static_assert(WM * WN == 4 && BM == WM * 128 && BN == WN * 64, "four waves of 128 x 64");
  constexpr int BK = 32, ST = 3;
  constexpr int WTM = 128, WTN = 64;
  constexpr int MT = WTM / 16, NT = WTN / 16;          // 8 x 4 tiles of 16 x 16
  constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, TILE_ELEMS = A_ELEMS + B_ELEMS;   // 24 KiB
  constexpr int PIECE = 16 * BK;                       // 16 rows of 64 B = 1 KiB
  constexpr int PA = BM / 16, PB = BN / 16, NP = (PA + PB) / 4;   // pieces per wave and K-tile: 6
  static_assert((PA + PB) % 4 == 0 && NP == 6, "six DMA pieces per wave");
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[ST * TILE_ELEMS * 2];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
  static_assert(ST * TILE_ELEMS * 2 >= 4 * 32 * (WTN + 4) * 4, "epilogue slabs fit in the operand ring");

  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  const int GROUP_M = g._p1 > 0 ? g._p1 : 4;
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * BM;
  const int64_t n0 = (int64_t)tn * BN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntiles = (int)g.K / BK;                    // K % 32 == 0 (host)

  const __bf16* Ah = reinterpret_cast<const __bf16*>(g.A);
  const __bf16* Bh = reinterpret_cast<const __bf16*>(g.B);
  const __bf16* src[NP];
  int dst[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = wave + 4 * i;                        // pieces 0..PA-1 = A rows 16p.., then B rows
    const bool isb = p >= PA;
    const int row = (isb ? p - PA : p) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ (3 * ((row >> 3) & 1));
    if (!isb) {
      int64_t gm = m0 + row;
      gm = gm < g.M ? gm : g.M - 1;                    // clamped rows feed accumulator rows that are never stored
      src[i] = Ah + gm * g.lda + chunk * 8;
      dst[i] = p * PIECE;
    } else {
      int64_t gn = n0 + row;
      gn = gn < g.N ? gn : g.N - 1;
      src[i] = Bh + gn * g.ldb + chunk * 8;
      dst[i] = A_ELEMS + (p - PA) * PIECE;
    }
  }
  auto issue_tile = [&](int slot) {
    __bf16* buf = smem + slot * TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      dma16(src[i], buf + dst[i]);
      src[i] += BK;
    }
  };

  f32x4v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int fr = lane & 15, fq = lane >> 4;
  int a_off[MT], b_off[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * WTM + i * 16 + fr;
    a_off[i] = row * BK + ((fq ^ (3 * ((row >> 3) & 1))) << 3);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * WTN + j * 16 + fr;
    b_off[j] = A_ELEMS + row * BK + ((fq ^ (3 * ((row >> 3) & 1))) << 3);
  }
  auto mfma_half = [&](const bf16x8 (&af)[4], const bf16x8 (&bf)[NT], int i0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (i0 == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        else acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[4 + i][j], 0, 0, 0);
      }
  };

  issue_tile(0);
  if (ntiles > 1) issue_tile(1);
  if (ntiles > 2) issue_tile(2);
  if (ntiles > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (ntiles > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 fa[2][4], fb[2][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(smem + b_off[j]);
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(smem + a_off[i]);
  int slot = 0;
  auto ktile = [&](int t, auto parity) {
    constexpr int P = decltype(parity)::value;
    const __bf16* Tc = smem + slot * TILE_ELEMS;
    const int nslot = slot == ST - 1 ? 0 : slot + 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[1][i] = *reinterpret_cast<const bf16x8*>(Tc + a_off[4 + i]);
    mfma_half(fa[0], fb[P], 0);
    if (t + 1 < ntiles) {
      // every fragment read of tile t has returned (its slot is refilled behind the barrier); this wave's pieces of
      // tile t+1 have landed once at most tile t+2's six are outstanding
      if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 3 < ntiles) issue_tile(slot);            // tile t+3 into tile t's slot
      const __bf16* Tn = smem + nslot * TILE_ELEMS;
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[P ^ 1][j] = *reinterpret_cast<const bf16x8*>(Tn + b_off[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(Tn + a_off[i]);
    }
    mfma_half(fa[1], fb[P], 4);
    slot = nslot;
  };
  for (int t = 0; t < ntiles; t += 2) {
    ktile(t, std::integral_constant<int, 0>{});
    if (t + 1 < ntiles) ktile(t + 1, std::integral_constant<int, 1>{});
  }
  __syncthreads();                                      // every wave has left the last K-tile: the ring becomes staging

  constexpr int STG_LD = WTN + 4;
  const DropCtx odc = make_drop(g.out_drop);
  float* stg = reinterpret_cast<float*>(smem_raw) + wave * (32 * STG_LD);
  auto stage_slab = [&](auto ic) {                     // 32 accumulator rows = tile rows 2 i, 2 i + 1
    constexpr int i = decltype(ic)::value;
    static_for<2>([&](auto tc) {
      constexpr int ti = decltype(tc)::value;
      static_for<NT>([&](auto jc) {
        constexpr int jn = decltype(jc)::value;
        static_for<4>([&](auto ec) {
          constexpr int e = decltype(ec)::value;
          stg[(16 * ti + 4 * fq + e) * STG_LD + jn * 16 + fr] = acc[2 * i + ti][jn][e];
        });
      });
    });
  };
  const int fmode = tecm_gemm::epi_fast_mode(g);
  constexpr int LPR = WTN / 4, RPI = 64 / LPR;
  const int lcol = (lane % LPR) * 4, lrow = lane / LPR;
  const EpiCol ecol = epi_col(g, n0 + wn * WTN + lcol);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + (ecol.ok ? ecol.n : 0));   // (lanes outside the matrix: column 0, never stored)
  tecm_gemm::epi_fast_dispatch<WTM / 32, 32 / RPI, RPI, STG_LD, true>(fmode, g, odc, stg, lrow, lcol, m0 + wm * WTM, ecol, bias4,
                                                                  stage_slab);
}

}  // namespace tecm_gemm16

int tecm_gemm16_p8_try(const TecmGemm& g, hipStream_t st);      // gemm_bf16_p8.hip

// Returns the number of K splits (1) when the DMA kernel served the call, 0 when the call is not eligible.
int tecm_gemm16_dma_try(const TecmGemm& g, hipStream_t st) {
  using namespace tecm_gemm16;
  const char* sel = std::getenv("TECM_BF16_DMA");       // "0": always use the register-staged kernel (A/B diagnostics)
  if (sel && sel[0] == '0') return 0;
  const bool both = (g.io_bf16 & TECM_IO_A_BF16) && (g.io_bf16 & TECM_IO_B_BF16);
  // K = 32 (lora_A's d-input) and N = 32 (lora_A's forward) are served by the 32-deep ring geometries only (round 4)
  const bool k32 = g.K < DBK, n_small = g.N < DBN / 2;
  if (!both || !(g.io_bf16 & TECM_P0_VEC4) || g.split_k > 1 || g.K % 32 != 0 || g.K < 32 || g.M < DBM || g.N < 32 ||
      ((k32 || n_small) && g.a_win.enabled) || (n_small && g.N > 32))    // N = 64 (first 1x1 conv): the register-staged
    return 0;                                                               //   kernel's 64-column tile is faster (114 vs 124 us)
  // the straight-line epilogues (gemm_impl.h: epi_fast_mode >= 0): at most one input stream; or a row bias; or an fp32
  // window scatter with nothing but bias / activation / dropout
  const int n_streams = (g.residual ? 1 : 0) + (g.dact_src ? 1 : 0) + (g.accumulate ? 1 : 0);
  const bool fast_epi = g.c_win.enabled ? (n_streams == 0 && !g.rowbias && !g.preact &&
                                           !(g.io_bf16 & (TECM_IO_C_BF16 | TECM_IO_PRE_BF16)))
                                        : (g.rowbias ? n_streams == 0 : n_streams <= 1);
  if (g.a_win.enabled) {
    // a pad-free window view of A whose taps are whole K-tiles (the patch projection's 'b (p l) d -> b p (l d)',
    // modules.py:114): the eight-phase geometry (round 5), else the first geometry -- both move their source pointers one
    // time step on at every tap boundary
    if (!sel && !k32 && !n_small && fast_epi && tecm_gemm16_p8_try(g, st)) return 1;
    const TecmWin& w = g.a_win;
    if (g.b_win.enabled || w.pad != 0 || w.Cw % DBK != 0 || (int64_t)(w.Lout - 1) * w.stride_t + w.taps > w.Lin ||
        g.M % ((int64_t)w.Lout * w.N) != 0)
      return 0;
    const int wtm = (int)((g.M + DBM - 1) / DBM), wtn = (int)((g.N + DBN - 1) / DBN);
    hipLaunchKernelGGL(gemm_bf16_dma_kernel, dim3((unsigned)(wtm * wtn)), dim3(DNTH), 0, st, g, wtm, wtn);
    TECM_CHECK_LAUNCH("tecm_gemm_bf16/dma-window");
    return 1;
  }
  // round 5: the eight-phase geometry (gemm_bf16_p8.hip) wherever it is eligible (it declines n-tiles that are mostly
  // padding, N = 800, unless forced); TECM_BF16_P8 = 0 or any TECM_BF16_DMA selection keeps the older geometries
  if (!sel && fast_epi && tecm_gemm16_p8_try(g, st)) return 1;
  // the 128-column geometry where the 256-column tile would waste more than half of its last n-tile (N = 800);
  // TECM_BF16_DMA = 1 / 2 forces one of the two (A/B diagnostics, tools/dma_ab.sh)
  const int nrem = (int)(g.N % DBN);
  const bool narrow = n_small || (sel && !k32 ? sel[0] == '2' : (nrem >= 1 && nrem <= D2N));
  if (narrow) {
    const int t2m = (int)((g.M + D2M - 1) / D2M), t2n = (int)((g.N + D2N - 1) / D2N);
    hipLaunchKernelGGL(gemm_bf16_dma2_kernel, dim3((unsigned)(t2m * t2n)), dim3(D2TH), 0, st, g, t2m, t2n);
    TECM_CHECK_LAUNCH("tecm_gemm_bf16/dma2");
    return 1;
  }
  const int tiles_m = (int)((g.M + DBM - 1) / DBM);
  const int tiles_n = (int)((g.N + DBN - 1) / DBN);
  // TECM_BF16_DMA = 3 / 4: the four-slot ring, plain and with anti-phase wave groups (A/B diagnostics, tools/dma_ab.sh).
  // Per shape the anti-phase ring measured K = 800: 418 -> 352 us, K = 3072 + residual epilogue: 517 -> 481, N = 3072
  // K = 768: 659 -> 688, 8192^3: 1081 vs 1085 TFLOP/s -- and 396.5 vs 395.9 samples/s in the step, i.e. nothing: it is
  // not dispatched by default.
  // the 16x16x32 ring wherever its straight-line epilogue serves the call: per shape (M = 69 864, tools/dma_ab.sh) K = 800
  // 397 -> 348 us (no zero-filled half K-tile), K = 3072 456 -> 412 us plain and a tie with the residual epilogue, N = 3072
  // K = 768 a tie, K = 768 N = 768 0..-5 %; in the step 392.1 samples/s against 389.0 with it on K = 800 / K >= 2048 only and
  // 386.8 without (one box).  TECM_BF16_DMA = 5 forces it (also for N = 800), 1 forbids it.
  {
    const int streams = (g.residual ? 1 : 0) + (g.dact_src ? 1 : 0) + (g.accumulate ? 1 : 0);
    const bool can16 = !g.c_win.enabled && !g.rowbias && streams <= 1;     // the straight-line epilogues only
    if (can16 && sel && (sel[0] == '6' || sel[0] == '7')) {                // four-wave blocks, two per CU (A/B diagnostics)
      if (sel[0] == '6') {
        const int t6m = (int)((g.M + 255) / 256), t6n = (int)((g.N + 127) / 128);
        hipLaunchKernelGGL((gemm_bf16_dma6_kernel<256, 128, 2, 2>), dim3((unsigned)(t6m * t6n)), dim3(256), 0, st, g, t6m, t6n);
      } else {
        const int t6m = (int)((g.M + 127) / 128), t6n = (int)((g.N + 255) / 256);
        hipLaunchKernelGGL((gemm_bf16_dma6_kernel<128, 256, 1, 4>), dim3((unsigned)(t6m * t6n)), dim3(256), 0, st, g, t6m, t6n);
      }
      TECM_CHECK_LAUNCH("tecm_gemm_bf16/dma6");
      return 1;
    }
    const bool want16 = (sel && !k32) ? (sel[0] == '5' || sel[0] == '8') : true;
    if (k32 && !can16) return 0;                           // the 64-deep geometries below cannot take K = 32
    if (can16 && want16) {
      // tile height: 256 rows, or 288 where that saves a mostly empty last round of blocks on the 256 CUs (N = 768 at
      // M = 69 864: 819 tiles = 3.2 rounds of 256-row tiles, 729 = 2.85 rounds of 288-row ones); TECM_BF16_DMA = 5 / 8 force one
      auto cost = [&](int64_t bm) {
        const int64_t t = ((g.M + bm - 1) / bm) * tiles_n;
        return ((t + 255) / 256) * bm;
      };
      const char* tl = std::getenv("TECM_BF16_TALL");      // "0": never (A/B diagnostics)
      const bool tall = sel && sel[0] == '8' ? true : ((sel && sel[0] == '5') || (tl && tl[0] == '0') ? false : cost(288) * 100 < cost(DBM) * 95);
      if (tall) {
        const int tm288 = (int)((g.M + 287) / 288);
        hipLaunchKernelGGL(gemm_bf16_dma5w_kernel, dim3((unsigned)(tm288 * tiles_n)), dim3(D3TH), 0, st, g, tm288, tiles_n);
        TECM_CHECK_LAUNCH("tecm_gemm_bf16/dma5w");
        return 1;
      }
      hipLaunchKernelGGL(gemm_bf16_dma5_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(D3TH), 0, st, g, tiles_m, tiles_n);
      TECM_CHECK_LAUNCH("tecm_gemm_bf16/dma5");
      return 1;
    }
  }
  const bool ring = sel && sel[0] == '4';
  if (ring) {
    hipLaunchKernelGGL(gemm_bf16_dma4_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(D3TH), 0, st, g, tiles_m, tiles_n);
    TECM_CHECK_LAUNCH("tecm_gemm_bf16/dma4");
    return 1;
  }
  if (sel && sel[0] == '3') {                           // A/B diagnostics: the four-slot ring geometry
    hipLaunchKernelGGL(gemm_bf16_dma3_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(D3TH), 0, st, g, tiles_m, tiles_n);
    TECM_CHECK_LAUNCH("tecm_gemm_bf16/dma3");
    return 1;
  }
  hipLaunchKernelGGL(gemm_bf16_dma_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(DNTH), 0, st, g, tiles_m, tiles_n);
  TECM_CHECK_LAUNCH("tecm_gemm_bf16/dma");
  return 1;
}
