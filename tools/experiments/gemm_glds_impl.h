// fp32 GEMM, direct-to-LDS variant for the plain dense contractions (GPT-2 c_attn / c_proj / c_fc and
// their dX): same 128 x 128 x 32 tile, wave layout, fragment order and epilogue as gemm_impl.h, but the
// operand tiles travel global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction)
// instead of through VGPRs + ds_write.  Measured on this kernel family (profiles/):
// the MFMA loop alone runs at 0.97 of the f32 matrix peak, register staging costs 12 % of that; the loads'
// VGPR write-back and the ds_write pass are what the DMA path removes.
//
// Eligibility (host side, tecm_gemm_glds_try): A in MK layout, no window / dropout prologue on A or B,
// K % 32 == 0 (no zero-filled K tail: LDS-DMA cannot mask), 16-byte friendly operands, N > 64, float4
// epilogue.  Everything else stays on gemm_impl.h's kernel.
//
// LDS image (the DMA writes wave-uniform base + lane*16, so every tile is stored linearly and the
// bank-conflict swizzle is applied to the per-lane SOURCE address and again on the fragment read):
//   [row][k] tiles (A: MK, B: NK): 128 rows x 128 B.  16-byte chunk c of row r sits at position
//       c ^ ((r >> 1) & 7): a ds_read_b128 of one k-chunk over 16 consecutive rows touches 16 distinct
//       4-bank groups.
//   [k][n] tiles (B: KN): 32 rows x 512 B.  Chunk c of k-row kr sits at c ^ (((kr >> 2) & 1) << 3): the two
//       lane halves of a ds_read_b32 (k and k+4) land in opposite halves of the 64 banks.
// Pipeline: two LDS buffers; the DMA of tile t+1 is issued before the MFMAs of tile t and retired by the
// s_waitcnt vmcnt(0) that __syncthreads() emits at the end of the tile (one barrier per K-tile).
#pragma once
#include "gemm_impl.h"

namespace tecm_gemm {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

template <int BLAY>
__global__ __launch_bounds__(512, 4) void gemm_glds_kernel(const TecmGemm g, int tiles_m, int tiles_n, int k_chunk) {
  constexpr int BN = 128, NWAVES = 8, WN = 4, WM = 2, WTM = 64, WTN = 32, MT = 2, NT = 1;
  constexpr int A_FLOATS = BM * BK;                 // 128 x 32, linear
  constexpr int B_FLOATS = BN * BK;
  constexpr int TILE_FLOATS = A_FLOATS + B_FLOATS;  // 32 KiB
  constexpr int STG_LD = WTN + 4;
  constexpr int STG_FLOATS = NWAVES * WTM * STG_LD;
  constexpr int SMEM_FLOATS = 2 * TILE_FLOATS > STG_FLOATS ? 2 * TILE_FLOATS : STG_FLOATS;
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  // block -> tile map: identical to gemm_impl.h (XCD-contiguous runs, 8-m-tile groups)
  const int nwg = tiles_m * tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, local = id >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  const int GROUP_M = g._p1 > 0 ? g._p1 : 8;
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

  // ---- per-lane DMA sources.  Each wave moves two 1 KiB pieces of A and two of B per K-tile.
  // [row][k] piece p (0..15) = tile rows 8p .. 8p+7; lane -> (row 8p + lane/8, stored position lane%8).
  const float* asrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int64_t gm = m0 + row;
    gm = gm < g.M ? gm : g.M - 1;                       // clamped rows feed accumulator rows that are never stored
    asrc[i] = g.A + gm * g.lda + kbeg + chunk * 4;
  }
  const float* bsrc[2];
  int64_t bstep;
  if constexpr (BLAY == TECM_B_NK) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave * 2 + i) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      int64_t gn = n0 + row;
      gn = gn < g.N ? gn : g.N - 1;
      bsrc[i] = g.B + gn * g.ldb + kbeg + chunk * 4;
    }
    bstep = BK;
  } else {
    // [k][n] piece p (0..15) = k-rows 2p, 2p+1; lane -> (k-row 2p + lane/32, stored position lane%32)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = (wave * 2 + i) * 2 + (lane >> 5);
      const int chunk = (lane & 31) ^ (((kr >> 2) & 1) << 3);
      int64_t gn = n0 + chunk * 4;
      gn = gn < g.N ? gn : 0;                           // N % 4 == 0: a chunk is entirely in or out
      bsrc[i] = g.B + (int64_t)(kbeg + kr) * g.ldb + gn;
    }
    bstep = (int64_t)BK * g.ldb;
  }
  auto issue_tile = [&](float* buf) {
    float* a_dst = buf + (wave * 2) * 256;              // 1 KiB = 256 floats per piece
    float* b_dst = buf + A_FLOATS + (wave * 2) * 256;
    dma16(asrc[0], a_dst);
    dma16(asrc[1], a_dst + 256);
    dma16(bsrc[0], b_dst);
    dma16(bsrc[1], b_dst + 256);
    asrc[0] += BK; asrc[1] += BK;
    bsrc[0] += bstep; bsrc[1] += bstep;
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;

  // fragment addresses (floats, within a tile buffer)
  int a_off[MT], a_sw[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * WTM + i * 32 + r;
    a_off[i] = row * BK;
    a_sw[i] = (row >> 1) & 7;
  }
  const int bcol = wn * WTN + r;
  const int b_sw = (bcol >> 1) & 7;                     // NK
  auto read_frags = [&](const float* As, const float* Bs, auto qc, float (&af)[MT][4], float (&bf)[4]) {
    constexpr int q = decltype(qc)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(&As[a_off[i] + (((2 * q + h) ^ a_sw[i]) << 2)]);
      af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
    }
    if constexpr (BLAY == TECM_B_NK) {
      const float4 v = *reinterpret_cast<const float4*>(&Bs[bcol * BK + (((2 * q + h) ^ b_sw) << 2)]);
      bf[0] = v.x; bf[1] = v.y; bf[2] = v.z; bf[3] = v.w;
    } else {
      // k = 8q + 4h + j: (k >> 2) & 1 == h, so the column swizzle of this lane is h * 32 for every j
      const int col = bcol ^ (h << 5);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = Bs[(8 * q + 4 * h + j) * BN + col];
    }
  };
  auto do_mfma = [&](const float (&af)[MT][4], const float (&bf)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j], bf[j], acc[i][0], 0, 0, 0);
  };

  issue_tile(smem);
  __syncthreads();                                      // vmcnt(0) + barrier: tile 0 has landed

  float fa[2][MT][4], fb[2][4];
  int cur = 0;
  for (int32_t k0 = kbeg; k0 < kend; k0 += BK) {
    const float* Ac = smem + cur * TILE_FLOATS;
    const float* Bc = Ac + A_FLOATS;
    if (k0 + BK < kend) issue_tile(smem + (cur ^ 1) * TILE_FLOATS);
    read_frags(Ac, Bc, std::integral_constant<int, 0>{}, fa[0], fb[0]);
    static_for<4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (q < 3) read_frags(Ac, Bc, std::integral_constant<int, q + 1>{}, fa[(q + 1) & 1], fb[(q + 1) & 1]);
      do_mfma(fa[q & 1], fb[q & 1]);
      constexpr int NREAD = MT + (BLAY == TECM_B_NK ? 1 : 4);
#pragma unroll
      for (int m = 0; m < MT * 4; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  // 1 MFMA
        if (q < 3 && m < NREAD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // 1 DS read
      }
    });
    __syncthreads();                                    // DMA of tile t+1 retired, every wave done with tile t
    cur ^= 1;
  }

  block_epilogue<MT, NT, WTM, WTN, STG_LD>(g, acc, smem, wave, lane, wm, wn, m0, n0);
}

template <int BLAY>
int launch_glds(const TecmGemm& g, hipStream_t st) {
  constexpr int BN = 128;
  const int tiles_m = (int)((g.M + BM - 1) / BM);
  const int tiles_n = (int)((g.N + BN - 1) / BN);
  int splits = g.split_k > 1 ? g.split_k : 1;
  int k_chunk = (int)(((g.K + splits - 1) / splits + BK - 1) / BK) * BK;
  splits = (int)((g.K + k_chunk - 1) / k_chunk);
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)splits);
  hipLaunchKernelGGL((gemm_glds_kernel<BLAY>), grid, dim3(512), 0, st, g, tiles_m, tiles_n, k_chunk);
  TECM_CHECK_LAUNCH("tecm_gemm_f32/glds");
  return splits;
}

}  // namespace tecm_gemm

// returns the number of K splits launched (> 0), a negative TECM_E_* code, or 0 when the call is not
// eligible for the direct-to-LDS kernel (caller falls through to gemm_impl.h's kernel)
int tecm_gemm_glds_try(const TecmGemm& g, int avec, int bvec, bool win, bool drop, hipStream_t st);
