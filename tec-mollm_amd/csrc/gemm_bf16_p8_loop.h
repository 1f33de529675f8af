// K loop of the "eight-phase" bf16 GEMM geometry (round 5): C[256 x 256] += A[256 x K] . B[256 x K]^T, both operands bf16
// [row][k] in HBM, fp32 accumulators in registers.  Built to the recipe of /opt/skills/guides/cdna_hip_programming.md
// section 5 ("The 256^2 8-phase template"): 64-deep K tiles cut into four 16 KiB HALF-tiles (A0, B0, B1, A1), one
// C-quadrant (64 x 32 per wave, 16 MFMAs of 16x16x32) per phase, the two wave groups of a CU (one wave of each per SIMD)
// running half a phase apart so that one group's 16 MFMAs cover the other group's fragment reads and DMA issue, operand
// DMAs (global_load_lds_dwordx4) kept in flight ACROSS the raw barriers behind counted s_waitcnt vmcnt.
//
// Geometry
//   * block = 512 threads = 8 waves as 2 (m) x 4 (n); wave (wr, wc) owns C rows wr*128 .. +128, columns wc*64 .. +64
//     (128 accumulator registers) as 2 x 2 quadrants of 64 rows x 32 columns: quadrant (hA, hB) = rows +hA*64, and the
//     columns c of the wave's 64 with (c >> 3) & 1 == hB (see the accumulator map at kloop).
//   * half-tile A_h = the 64-row halves hA = h of BOTH wave rows (128 LDS rows of 64 k = 128 B each), B_h likewise the
//     32-column sets hB = h of the four wave columns: every wave reads 64 (A) / 32 (B) rows of a half-tile, and a
//     half-tile is consumed by exactly one or two phases.  LDS: 2 K-tile buffers x 4 half-tiles x 16 KiB = 128 KiB.
//   * a half-tile is 16 pieces of 8 rows x 128 B, one global_load_lds_dwordx4 wave-instruction each (full 128-B lines
//     from HBM/L2), wave w moves pieces 2w, 2w+1.  The DMA writes wave-uniform base + lane * 16, so the image is linear
//     and the bank swizzle is applied to the per-lane SOURCE address and again on the fragment read: 16-byte chunk c of
//     row r sits at position c ^ ((r >> 1) & 7).  A 16x16x32 operand fragment is one ds_read_b128 (lane l: row l & 15,
//     chunk 4 ks + (l >> 4)); with this swizzle each of its four 16-lane hardware groups ({0-3,12-15,20-27}, ...) touches
//     16 distinct 4-bank groups.
//
// Schedule (global phase g = 4 t + p of K-tile t; half-tile sequence s = 4 t + kind, kinds in consumption order
// A0, B0, B1, A1; S = 4 T half-tiles in all)
//   phase p=0: read B0, A0 (12 ds_read_b128) | issue s = g + 6 | vmcnt -> B1(t) landed | barrier | c[0][0] | barrier
//   phase p=1: read B1 (4)                   | issue           | vmcnt -> A1(t)        | barrier | c[0][1] | barrier
//   phase p=2: read A1 (8)                   | issue           |                       | barrier | c[1][1] | barrier
//   phase p=3:                               | issue           | vmcnt -> A0,B0(t+1)   | barrier | c[1][0] | barrier
//   Every phase issues exactly one half-tile (two DMAs per wave), six half-tiles ahead of the phase: s = g + 6 lands in
//   the slot of s - 8, whose last fragment read was at phase g - 2 (A0) or g - 3 -- two or more phases earlier, which is
//   what the half-phase stagger needs (the later group's reads of phase q are retired by ITS lgkmcnt(0) behind global
//   barrier 2q+1, the earlier group issues phase q+2's DMA behind barrier 2q+3).  A wait of phase w sits before that
//   phase's first barrier and names data first read in phase w + 1 (never the same phase): each wave retires its own
//   DMAs, the barrier publishes them.  In steady state every wait is vmcnt(8) -- four half-tiles stay in flight --, the
//   last K-tile counts down 4 / 2 / 0.  No vmcnt(0) inside the loop, no __syncthreads().
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace tecm_p8 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BM = 256, BN = 256, BK = 64, NTH = 512;
constexpr int HALF_BYTES = 128 * BK * 2;              // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;             // one K-tile: A0, B0, B1, A1
constexpr int LDS_BYTES = 2 * BUF_BYTES;              // 128 KiB
#ifndef P8_MAP
#define P8_MAP 1
#endif
constexpr int BMAP = P8_MAP;                          // which of the wave's 64 columns a B fragment row holds (see kloop)
constexpr int MT = 8, NT = 4;                         // accumulator tiles of 16 x 16 per wave: rows 16 i, columns 16 j

struct Operands {
  const __bf16* A; const __bf16* B;                   // [M][lda], [N][ldb], bf16, 16-byte aligned rows
  int64_t lda, ldb, M, N; int K;                      // K % 32 == 0, K >= 128; byte offsets of both operands < 4 GiB
  // A as a pad-free window view (TecmWin, include/tecmollm.h; the patch projection's 'b (p l) d -> b p (l d)', modules.py:114):
  // row m = (bq, t_out, n) starts at source row (bq*Lin + t_out*stride_t)*wN + n, K index kk = (tap, c) lies `tap` time
  // steps = tap*wN source rows further on.  Cw % 64 == 0: a K-tile never straddles a tap.  wN == 0: the plain view.
  int32_t wN, wLin, wLout, wstride, wCw;
};                                                    // (K % 64 == 32, c_attn with its 32 LoRA columns: the last K-tile is
                                                      //  half deep -- its DMAs re-read the valid half, its MFMAs stop at k = 32)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void bar() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// Accumulator map (the B fragment is the MFMA's row operand, and the B rows are dealt to fragment rows so that):
//   BMAP 1: acc[i][j][e] = C[m0 + wr*128 + 16 i + (lane & 15)][n0 + wc*64 + 16 (lane >> 4) + 4 j + e]
//   BMAP 0: acc[i][j][e] = C[m0 + wr*128 + 16 i + (lane & 15)][n0 + wc*64 + 16 j + 4 (lane >> 4) + e]
// -- lane (fr, fq) holds, for each of its 8 row tiles i, the 16 consecutive columns 16 fq .. 16 fq + 15 of row 16 i + fr:
// the epilogue works row-wise straight from the registers (64 B fp32 / 32 B bf16 per lane and row), no LDS staging.
//
// WR = rows of C a wave row owns: 128, or 112 / 96 -- the SHORT tiles (block tile 2 WR x 256).  A short tile moves the same
// half-tiles (its A1 rows 64 .. 127 of a wave row reach into the next wave row's / tile's rows; they are loaded and not
// used) and runs 12 / 8 MFMAs instead of 16 in the two phases of the lower quadrants: accumulator row tiles WR/16 .. 7
// stay zero and are never stored.  What it buys is the tile COUNT: 273 x 3 tiles of 256 rows (M = 69 864, N = 768) are 3.2
// rounds of 256 CUs = four rounds of work for 3.2 rounds of result; 312 x 3 tiles of 224 rows are 3.66 rounds of 7/8 the work.
template <int WR = 128>
__device__ __forceinline__ void kloop(const Operands& o, int64_t m0, int64_t n0, unsigned char* smem, f32x4 (&acc)[MT][NT]) {
  static_assert(WR == 128 || WR == 112 || WR == 96, "a wave row owns 8, 7 or 6 row tiles of 16");
  constexpr int MTU = WR / 16;                          // row tiles in use
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int T = (o.K + BK - 1) / BK, S = 4 * T;
  const bool ktail = (o.K % BK) != 0;

  // ---- DMA sources: per kind (A0, B0, B1, A1) and piece (2) a 32-bit byte offset from the operand base
  uint32_t soff[4][2], tadj[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {                       // half-deep last tile: chunks 4..7 lie beyond K -> fetch chunk - 4 again
    const int lr = 16 * wave + 8 * i + (lane >> 3);
    tadj[i] = (((lane & 7) ^ ((lr >> 1) & 7)) >= 4) ? 64u : 0u;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int h = k >> 1;
      const bool isA = (k == 0 || k == 3);
      const int lr = 16 * wave + 8 * i + (lane >> 3);
      const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
      if (isA) {
        int64_t gm = m0 + (lr >> 6) * WR + h * 64 + (lr & 63);
        gm = gm < o.M ? gm : o.M - 1;                  // clamped rows feed accumulator rows that are never stored
        if (o.wN > 0) {                                // window view: the source row of tap 0 (32-bit arithmetic, M < 2^31)
          const uint32_t q = (uint32_t)gm / (uint32_t)o.wN, n = (uint32_t)gm - q * (uint32_t)o.wN;
          const uint32_t bq = q / (uint32_t)o.wLout, t_out = q - bq * (uint32_t)o.wLout;
          gm = ((int64_t)bq * o.wLin + (int64_t)t_out * o.wstride) * o.wN + n;
        }
        soff[k][i] = (uint32_t)((gm * o.lda + chunk * 8) * 2);
      } else {
        // LDS row lr of B_h = wave column lr >> 5, MFMA tile jl = (lr >> 4) & 1 of the half, fragment row r16 = lr & 15;
        // it holds column (r16 >> 2) * 16 + (2 h + jl) * 4 + (r16 & 3) of the wave's 64, so that a lane's 4 j x 4 e
        // accumulator values are 16 CONSECUTIVE columns of one row (see the accumulator map above kloop)
        const int r16 = lr & 15, jt = 2 * h + ((lr >> 4) & 1);
        int64_t gn = n0 + (lr >> 5) * 64 + (BMAP == 1 ? (r16 >> 2) * 16 + jt * 4 + (r16 & 3) : jt * 16 + r16);
        gn = gn < o.N ? gn : o.N - 1;
        soff[k][i] = (uint32_t)((gn * o.ldb + chunk * 8) * 2);
      }
    }
  const char* Ab = reinterpret_cast<const char*>(o.A);
  const char* Bb = reinterpret_cast<const char*>(o.B);
  int kA0 = 0, kB0 = 0, kB1 = 0, kA1 = 0;              // bytes along K already issued, per kind (wave-uniform)
  // window view of A: after the last K-tile of a tap the source moves on by one time step (wN source rows) less the tap
  const int a_tpt = o.wN > 0 ? o.wCw / BK : 0x40000000;  // K-tiles per tap
  const int a_jump = o.wN > 0 ? (int)(((int64_t)o.wN * o.lda - o.wCw) * 2) : 0;
  int iA0 = 0, iA1 = 0;                                // K-tiles issued inside the current tap, per A kind
  auto issue = [&](auto kc, int slot_bytes, bool tail_tile = false) {
    constexpr int k = decltype(kc)::value;
    const uint32_t tm = tail_tile ? 0xffffffffu : 0u;
    constexpr bool isA = (k == 0 || k == 3);
    int& kk = k == 0 ? kA0 : (k == 1 ? kB0 : (k == 2 ? kB1 : kA1));
    const char* base = (isA ? Ab : Bb) + kk;
    unsigned char* dst = smem + slot_bytes + k * HALF_BYTES + wave * 2048;
    __builtin_amdgcn_global_load_lds((glb_void*)(base + (soff[k][0] - (tadj[0] & tm))), (lds_void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + (soff[k][1] - (tadj[1] & tm))), (lds_void*)(dst + 1024), 16, 0, 0);
    kk += BK * 2;
    if constexpr (isA) {
      int& it = k == 0 ? iA0 : iA1;
      const bool wrap = ++it == a_tpt;
      kk += wrap ? a_jump : 0;
      it = wrap ? 0 : it;
    }
  };
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;

  // ---- fragment read offsets within a half-tile (bytes): row (w*rows + 16 i + fr) * 128 + swizzled chunk; the swizzle
  // depends on fr only (the other row terms are multiples of 16)
  int aoff[2], boff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int pos = ((4 * ks + fq) ^ ((fr >> 1) & 7)) << 4;
    aoff[ks] = (wr * 64 + fr) * 128 + pos;
    boff[ks] = (wc * 32 + fr) * 128 + pos;
  }
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  auto read_a = [&](int slot_bytes, auto kindc) {
    constexpr int kind = decltype(kindc)::value;         // 0: A0 (row tiles 0-3), 3: A1 (row tiles 4 .. MTU-1)
    constexpr int NI = kind == 0 ? 4 : MTU - 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NI; ++i)
        fa[i][ks] = *reinterpret_cast<const bf16x8*>(smem + slot_bytes + kind * HALF_BYTES + aoff[ks] + i * 2048);
  };
  auto read_b = [&](bf16x8 (&fb)[2][2], int slot_bytes, int kind) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[j][ks] = *reinterpret_cast<const bf16x8*>(smem + slot_bytes + kind * HALF_BYTES + boff[ks] + j * 2048);
  };
  auto quadrant = [&](auto hac, auto hbc, const bf16x8 (&fb)[2][2], bool half_deep) {
    constexpr int HA = decltype(hac)::value, HB = decltype(hbc)::value;
    constexpr int NI = HA == 0 ? 4 : MTU - 4;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1 && half_deep) break;                  // wave-uniform
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 * HA + i][2 * HB + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks], fa[i][ks], acc[4 * HA + i][2 * HB + j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: six half-tiles (K-tile 0, A0/B0 of K-tile 1), wait for A0, B0 of tile 0
  issue(K0{}, 0); issue(K1{}, 0); issue(K2{}, 0); issue(K3{}, 0);
  issue(K0{}, BUF_BYTES, ktail && T == 2); issue(K1{}, BUF_BYTES, ktail && T == 2);
  wait_vm<8>();
  bar();
  if (wr == 1) bar();                                   // the second wave group runs one barrier (half a phase) behind

  auto ktile = [&](int t, auto dc) {
    constexpr int cur = decltype(dc)::value * BUF_BYTES, oth = BUF_BYTES - cur;
    const int g = 4 * t;
    const bool last = t + 1 == T;
    const bool hd = last && ktail;                       // this tile is the half-deep one
    const bool tl1 = ktail && t + 2 == T, tl2 = ktail && t + 3 == T;   // ... or the next / next-but-one is
    // phase 0: c[0][0]
    read_b(fb0, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_a(cur, K0{});
    if (g + 6 < S) issue(K2{}, oth, tl1);
    if (!last) wait_vm<8>(); else wait_vm<2>();
    bar();
    quadrant(K0{}, K0{}, fb0, hd);
    bar();
    // phase 1: c[0][1]
    read_b(fb1, cur, 2);
    if (g + 7 < S) issue(K3{}, oth, tl1);
    if (!last) wait_vm<8>(); else wait_vm<0>();
    bar();
    quadrant(K0{}, K1{}, fb1, hd);
    bar();
    // phase 2: c[1][1]
    read_a(cur, K3{});
    if (g + 8 < S) issue(K0{}, cur, tl2);
    bar();
    quadrant(K1{}, K1{}, fb1, hd);
    bar();
    // phase 3: c[1][0]
    if (g + 9 < S) issue(K1{}, cur, tl2);
    if (t + 2 < T) wait_vm<8>(); else if (!last) wait_vm<4>();
    bar();
    quadrant(K1{}, K0{}, fb0, hd);
    bar();
  };
  for (int t = 0; t < T; t += 2) {
    ktile(t, K0{});
    if (t + 1 < T) ktile(t + 1, K1{});
  }
  if (wr == 0) bar();                                   // pairs with the second group's last barrier
}

}  // namespace tecm_p8
