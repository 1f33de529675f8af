// Multi_Scale_Conv_Block (reference modules.py:43-60): the WEIGHT gradient of the three parallel Conv1d
// (k = 3, 5, 7) as ONE kernel that reads the block input and dy once.
//
//     dw_j[co, ci, tau] = sum_{b, t, n}  dy[b, t, n, j*Cout + co] * inp[b, t + tau - p_j, n, ci],     p_j = (k_j - 1) / 2
//
// Before: three split-K window GEMMs (A = dy^T, B = the window view of inp, K = B*Lc*N up to 1.1 M rows), each staging
// its operands through a transposing register stager and re-reading dy / the taps of inp: 0.54 + 0.60 ms per step at
// 60-250 TFLOP/s for 2 x 0.5 GB of operands -- five times what HBM needs to deliver them once.  Here: a SEQUENCE tile.
//   * a persistent block (one per CU, 8 waves) walks tiles of 4 nodes x up to 24 time steps of one sample; the tile's
//     rows of inp and dy are staged in LDS in their NATURAL layout (row = (t, node), channels contiguous), inp with 3
//     more time steps in front and behind (zeros outside the sequence), so every tap of every kernel size is a ROW
//     offset of the same image and the zero padding of the convolution is the halo;
//   * the contraction runs over rows, which is the k index of the MFMA: both operands are wanted "k-major".
//     ds_read_b64_tr_b16 (gfx950) delivers exactly that from the natural image -- lane 4q+p of a 16-lane group supplies
//     row q (= node q of one time step), lane i receives column i of the 4 rows -- so nothing is transposed in
//     registers and each lane's row address is linear in the k-chunk index;
//   * v_mfma_f32_32x32x16_bf16 with A = 32 input channels x 16 rows, B = 16 rows x 32 output channels.  A PAIR of waves
//     owns one (input-channel block, output-channel block) pair and all 15 taps of it -- kernel size 7 in one wave,
//     5 and 3 in the other: 7-8 accumulator tiles = 128 registers that live across the whole kernel.  Per 16-row
//     k-chunk a wave issues 8 transposed reads of inp (time steps t-3..t+4: the upper half of tap o's fragment is the
//     lower half of tap o+1's) + 4 of dy for 7-8 MFMAs;
//   * the 15 x Cin x Cout accumulators of Cout = 128 / ld_in = 64 are 480 KB -- the whole register file of a CU holds
//     512 KB -- so a block owns only HALF the output channels ("flavor" = blockIdx % 2) and stages only those columns
//     of dy; inp (1/6 of the bytes) is read by both flavors.  Cout = 64 / ld_in = 24 has only two (cib, cob) pairs:
//     the 4 wave pairs then also split the k-chunks two ways;
//   * the next TWO tiles travel from HBM into registers (2 x 7-8 x 16 B per lane) while the current one is multiplied;
//     the loads are issued and awaited by hand (see issue_load);
//   * every wave writes its 15 tiles once, at the end, to a per-block slab; conv_dw_reduce_kernel sums the slabs in a
//     fixed order and writes the three (Cout, Cin, k) gradients.  Bit-reproducible.
// Arithmetic = the bf16 mode's: operands are the bf16 tensors the GEMM path reads (inp16 written by the producer, dy by
// the GroupNorm backward), fp32 accumulation; only the summation order differs.
// The SAME kernel in exact fp32 (template F32, BASELINE configs[1]): fp32 inp and dy, v_mfma_f32_32x32x2_f32 fed by one
// ds_read_b32 per operand from the natural image (no transposition needed: a lane holds one element).  There the kernel
// is bound by the matrix cores, and what it saves over the split-K window GEMMs is their re-staging of both operands.
// Measured (B = 8, N = 2911, round 3, in the step): bf16 block 1 (Lc 48, ld_in 24, Cout 64) 540 -> 120 us (5.1 TB/s of the
// two operands), block 2 (Lc 24, 64, 128) 600 -> 170 us (3.7 TB/s) + 17 us of reduction each; fp32 870 -> 620 us and
// 1450 -> 1150 us (92 / 119 TFLOP/s of MFMA issued; the K loop alone 0.82 of the f32 matrix peak).  DESIGN.md Appendix B.8.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace tecm_convdw {

#ifndef CDW_ABLATE
#define CDW_ABLATE 0         // diagnostics (tools/build_variant.py -DCDW_ABLATE=n): bit0 no HBM loads, bit1 no LDS stores, bit2 no K loop
#endif
constexpr int NTH = 512;     // 8 waves = 4 (channel block pair | k range) x 2 tap groups; 8 accumulator tiles each
constexpr int NB = 4;        // nodes per tile = rows of one transposed-read block
constexpr int HALO = 3;      // zero time steps in front of / behind the inp image (k = 7)
constexpr int NTAP = 15;     // 3 + 5 + 7
constexpr int TCMAX = 24;    // time steps per tile (longer sequences are walked in chunks; the halo rows are re-read)

struct Args {
  const void* x;             // bf16 (B, Lc, N, ld_in)
  const void* dy;            // bf16 (B, Lc, N, 3 * Cout)
  float* ws;                 // slabs [flavor][(gridDim.x / F) * KS][15][32 * NCIB][32 * CPB]
  int B, Lc, N, ntiles, nblk, TC, nchunk;
};

template <int LD_IN, int COUT, bool F32>
struct Geo {
  static constexpr int ES = F32 ? 4 : 2;                     // element size
  static constexpr int NCIB = (LD_IN + 31) / 32;             // input-channel blocks of 32
  static constexpr int NCOB = COUT / 32;
  static constexpr int CPB = NCOB < 4 / NCIB ? NCOB : 4 / NCIB;   // output-channel blocks per thread block
  static constexpr int F = NCOB / CPB;                       // flavors: thread blocks that share a tile's inp, split dy's columns
  static constexpr int R = NCIB * CPB;                       // (cib, cob) pairs = wave roles
  static constexpr int KS = 4 / R;                           // wave pairs per role: they split the k range
  static constexpr int XB = LD_IN * ES, YB = 3 * COUT * ES;  // bytes of one row in HBM
  static constexpr int SEGB = CPB * 32 * ES;                 // bytes of one branch's columns of this flavor
  static constexpr int YI = 3 * SEGB;                        // bytes of one dy row in the image
  // bf16: pitches are odd multiples of 64 B -- the 4 rows x 64 B of a transposed read (one 32-lane half) tile all 64
  // banks.  fp32: a half-wave reads 32 consecutive floats of one row, any pitch is conflict-free.
  static constexpr int XP = F32 ? XB : (NCIB == 1 ? 64 : 192);
  static constexpr int YP = F32 ? YI : YI + 64;
  static constexpr int CPRX = XB / 16;                       // 16-byte chunks per inp row
  static constexpr int CPS = SEGB / 16;                      // chunks per branch segment
  static constexpr int MAXROWS = TCMAX * NB;                 // dy rows of a tile
  static constexpr int MAXXROWS = (TCMAX + 2 * HALO) * NB;   // inp rows of a tile: the halo is staged too
  static constexpr int XBYTES = (TCMAX + 2 * HALO + 1) * NB * XP + 64, YBYTES = TCMAX * NB * YP;   // + slack: see compute
  // staging map, divisions by powers of two only: LX (LY) consecutive lanes share a row of inp (dy); an inp lane moves one
  // chunk per pass, a dy lane three (the same chunk of each branch segment)
  static constexpr int LX = CPRX <= 4 ? 4 : (CPRX <= 8 ? 8 : 16), LY = CPS;
  static constexpr int RPX = NTH / LX, RPY = NTH / LY;       // rows per pass
  static constexpr int SBX = (MAXXROWS + RPX - 1) / RPX, SBY = 3 * ((MAXROWS + RPY - 1) / RPY);
  static constexpr int SLAB = NTAP * 32 * NCIB * 32 * CPB;   // floats
  static_assert(R * KS == 4 && CPB * F == NCOB, "roles must tile the 4 wave pairs");
  static_assert(LD_IN % 8 == 0 && COUT % 32 == 0 && CPRX <= LX && (LY == 8 || LY == 16 || LY == 32), "shape");
  static_assert(F32 || (YP / 64) % 2 == 1, "bf16 dy pitch must be an odd multiple of 64 B");
  static_assert(XBYTES % 16 == 0 && XP % 16 == 0 && YP % 16 == 0, "16-byte rows");
};

typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
__device__ __forceinline__ bf16x4 tr_read(lds_ptr p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(p));
}
// The staging loads are issued and awaited BY HAND.  With compiler-tracked loads the register sets that travel
// round-robin through the tile loop end in `s_waitcnt vmcnt(0)` at the top of every K loop (the wait-count pass merges
// the loop's paths pessimistically): the tile just requested was awaited before the current one was multiplied, and
// staging and MFMA time added up (155 + 89 us at Cout = 128, measured).  vmcnt retires in order, so "at most `newer`
// operations outstanding" is exactly "everything older has landed"; anything the compiler adds in between only makes
// the wait stricter.  land() ties the registers to the wait so no use can move above it.
__device__ __forceinline__ void issue_load(u32x4& dst, const void* src) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src));
}
template <int NEWER>
__device__ __forceinline__ void wait_loads() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER));
}
__device__ __forceinline__ void land(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ bf16x8 join(const bf16x4& lo, const bf16x4& hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct Tile {
  int b, n0, t0, tc;
};
__device__ __forceinline__ Tile decode_tile(const Args& a, int tile) {
  Tile t;
  const int nbk = tile % a.nblk;
  tile /= a.nblk;
  const int chunk = tile % a.nchunk;
  t.b = tile / a.nchunk;
  t.n0 = nbk * NB;
  t.t0 = chunk * a.TC;
  t.tc = min(a.TC, a.Lc - t.t0);                             // multiple of 4
  return t;
}

template <int LD_IN, int COUT, int PD, bool F32>   // PD: tiles in flight from HBM (register sets)
__global__ __launch_bounds__(NTH) void conv_dw_seq_kernel(const Args a) {
  using G = Geo<LD_IN, COUT, F32>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);   // scalar: everything derived from it branches, not masks
  constexpr int xbytes = G::XBYTES, ybytes = G::YBYTES;
  unsigned char* xs = lds;
  unsigned char* ys = lds + xbytes;
  {                                                          // pad bytes are zero for the whole kernel
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid0 * 16; i < xbytes + ybytes; i += NTH * 16) *reinterpret_cast<u32x4*>(lds + i) = z;
  }
  // tap group: 0 = kernel size 7 (7 MFMAs per step), 1 = kernel sizes 5 and 3 (8).  Waves w and w + 4 share a SIMD: the
  // XOR gives every SIMD one wave of each group (15 MFMAs per step everywhere instead of 14 / 16)
  const int tg = (wave ^ (wave >> 2)) & 1;
  const int role = (wave >> 1) % G::R, kq = (wave >> 1) / G::R;
  const int cib = role % G::NCIB, cob = role / G::NCIB;      // cob: local to this block's flavor
  const int flavor = blockIdx.x % G::F, bif = blockIdx.x / G::F, nbf = gridDim.x / G::F;
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const char* xg = static_cast<const char*>(a.x);
  const char* yg = static_cast<const char*>(a.dy);
  u32x4 vx[PD][G::SBX], vy[PD][G::SBY];
  // inp image row r = (time t0 - 3 + (r >> 2), node r & 3), dy image row r = (time t0 + (r >> 2), node r & 3).  Loads are
  // clamped (always in bounds), only real chunks are stored; rows of nodes >= N and of times outside [0, Lc) -- the zero
  // padding of the convolution -- are stored as zeros.
  const int xrow = tid0 / G::LX, xch = tid0 % G::LX, yrow = tid0 / G::LY, ych = tid0 % G::LY;
  auto stage_load = [&](int tile, u32x4 (&wx)[G::SBX], u32x4 (&wy)[G::SBY]) {
    const Tile t = decode_tile(a, tile);
    const int64_t row0 = (int64_t)t.b * a.Lc * a.N;
#pragma unroll
    for (int q = 0; q < G::SBX; ++q) {
      const int row = xrow + q * G::RPX, ch = min(xch, G::CPRX - 1);
      const int ts = min(max(t.t0 - HALO + (row >> 2), 0), a.Lc - 1);
      const int ng = min(t.n0 + (row & 3), a.N - 1);
      issue_load(wx[q], xg + (row0 + (int64_t)ts * a.N + ng) * G::XB + ch * 16);
    }
#pragma unroll
    for (int q = 0; q < G::SBY / 3; ++q) {
      const int row = yrow + q * G::RPY;
      const int ts = min(t.t0 + (row >> 2), a.Lc - 1);
      const int ng = min(t.n0 + (row & 3), a.N - 1);
      const char* src = yg + (row0 + (int64_t)ts * a.N + ng) * G::YB + flavor * G::SEGB + ych * 16;
#pragma unroll
      for (int seg = 0; seg < 3; ++seg) issue_load(wy[3 * q + seg], src + seg * (COUT * G::ES));
    }
  };
  auto stage_store = [&](int tile, const u32x4 (&wx)[G::SBX], const u32x4 (&wy)[G::SBY]) {
    const Tile t = decode_tile(a, tile);
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < G::SBX; ++q) {
      const int row = xrow + q * G::RPX;
      const int ts = t.t0 - HALO + (row >> 2);
      const bool real = ts >= 0 && ts < a.Lc && t.n0 + (row & 3) < a.N;
      if (row < (t.tc + 2 * HALO) * NB && xch < G::CPRX) *reinterpret_cast<u32x4*>(xs + row * G::XP + xch * 16) = real ? wx[q] : z;
    }
#pragma unroll
    for (int q = 0; q < G::SBY / 3; ++q) {
      const int row = yrow + q * G::RPY;
      if (row < t.tc * NB) {
        const bool real = t.n0 + (row & 3) < a.N;
#pragma unroll
        for (int seg = 0; seg < 3; ++seg)
          *reinterpret_cast<u32x4*>(ys + row * G::YP + seg * G::SEGB + ych * 16) = real ? wy[3 * q + seg] : z;
      }
    }
  };

  const lds_ptr base3 = (lds_ptr)lds;
  const int ya = (tg ? 1 : 2) * G::SEGB;                     // the main dy operand: third branch (k = 7) / second (k = 5)
  // One tile's product, both tap groups through the same code.  Operand i of inp = time offset i - 3 + tg (tap group 1
  // starts one time step later: its widest kernel is 5).
  //   tg 0 (kernel size 7): slots 0..6 = operands 0..6 against dy's third branch.
  //   tg 1: slots 0..4 = kernel size 5 (operands 0..4, offsets -2..2, second branch), slots 5..7 = kernel size 3
  //         (operands 1..3, offsets -1..1, first branch).
  auto compute = [&](int tc) {
    if constexpr (!F32) {
      // transposed reads: this lane supplies row q (node q) of time step 2h (+1 for the second read of a fragment),
      // columns 16*g1 + 4p .. +3 of the wave's channel block.  xr[i] = time step t + i - 3 + tg (t = 4*kc + 2h), operand
      // i = (xr[i], xr[i+1]): the upper half of one tap's fragment is the lower half of the next one's.
      const int q = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1, h = lane >> 5;
      const int x0 = ((2 * h + tg) * NB + q) * G::XP + (cib * 32 + 16 * g1 + 4 * p) * 2;   // tap offset -3 = image row t
      const int y0 = xbytes + (2 * h * NB + q) * G::YP + (cob * 32 + 16 * g1 + 4 * p) * 2;
      const int nchunks = tc / 4;                            // 16 rows = 4 time steps per step
      bf16x4 xr[8], yr[4];
      auto fetch = [&](int kc, bf16x4 (&x)[8], bf16x4 (&y)[4]) {
        const lds_ptr xp = base3 + x0 + kc * 16 * G::XP;
        const lds_ptr yp = base3 + y0 + kc * 16 * G::YP;
#pragma unroll
        for (int o = 0; o < 8; ++o) x[o] = tr_read(xp + o * NB * G::XP);
        y[0] = tr_read(yp + ya);
        y[1] = tr_read(yp + ya + NB * G::YP);
        y[2] = tr_read(yp);
        y[3] = tr_read(yp + NB * G::YP);
      };
      if (kq < nchunks) fetch(kq, xr, yr);
      for (int kc = kq; kc < nchunks; kc += G::KS) {
        bf16x4 xn[8], yn[4];                                 // the next step's operands are read while this one multiplies
        fetch(min(kc + G::KS, nchunks - 1), xn, yn);
        const bf16x8 yA = join(yr[0], yr[1]), yB = join(yr[2], yr[3]);
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5)
          acc[s5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(xr[s5], xr[s5 + 1]), yA, acc[s5], 0, 0, 0);
#pragma unroll
        for (int s5 = 5; s5 < 7; ++s5) {
          const bf16x8 xf = tg ? join(xr[s5 - 4], xr[s5 - 3]) : join(xr[s5], xr[s5 + 1]);
          acc[s5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, tg ? yB : yA, acc[s5], 0, 0, 0);
        }
        if (tg) acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(xr[3], xr[4]), yB, acc[7], 0, 0, 0);
#pragma unroll
        for (int o = 0; o < 8; ++o) xr[o] = xn[o];
#pragma unroll
        for (int o = 0; o < 4; ++o) yr[o] = yn[o];
      }
    } else {
      // v_mfma_f32_32x32x2_f32: lane = (channel m, row h of the pair) reads ONE float per operand, straight from the natural
      // image (32 consecutive floats of a row per half-wave).  ld_in = 24: lanes m >= 24 read into the next row -- finite
      // or not, that only reaches accumulator rows the reduction never reads (the image has 64 B of slack at its end).
      const int m = lane & 31, h = lane >> 5;
      const int x0 = (h + tg * NB) * G::XP + (cib * 32 + m) * 4;
      const int y0 = xbytes + h * G::YP + (cob * 32 + m) * 4;
      typedef __attribute__((address_space(3))) const float* lds_f;
      const int npairs = tc * 2;                             // 2 rows per step
      float xr[7], yA, yB;
      auto fetch = [&](int kp, float (&x)[7], float& ya_, float& yb_) {
        const lds_ptr xp = base3 + x0 + kp * 2 * G::XP;
        const lds_ptr yp = base3 + y0 + kp * 2 * G::YP;
#pragma unroll
        for (int o = 0; o < 7; ++o) x[o] = *reinterpret_cast<lds_f>(xp + o * NB * G::XP);
        ya_ = *reinterpret_cast<lds_f>(yp + ya);
        yb_ = *reinterpret_cast<lds_f>(yp);
      };
      if (kq < npairs) fetch(kq, xr, yA, yB);
      for (int kp = kq; kp < npairs; kp += G::KS) {
        float xn[7], yAn, yBn;                               // the next step's operands are read while this one multiplies
        fetch(min(kp + G::KS, npairs - 1), xn, yAn, yBn);
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) acc[s5] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[s5], yA, acc[s5], 0, 0, 0);
#pragma unroll
        for (int s5 = 5; s5 < 7; ++s5)
          acc[s5] = __builtin_amdgcn_mfma_f32_32x32x2f32(tg ? xr[s5 - 4] : xr[s5], tg ? yB : yA, acc[s5], 0, 0, 0);
        if (tg) acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[3], yB, acc[7], 0, 0, 0);
#pragma unroll
        for (int o = 0; o < 7; ++o) xr[o] = xn[o];
        yA = yAn;
        yB = yBn;
      }
    }
  };

  // PD register sets travel round-robin: while tile i is multiplied, tiles i+1 .. i+PD are on their way from HBM
#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (bif + s * nbf < a.ntiles) stage_load(bif + s * nbf, vx[s], vy[s]);
  __syncthreads();                                           // the zero fill is complete
  for (int tile = bif; tile < a.ntiles;) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      if (tile < a.ntiles) {                                 // block-uniform
        // outstanding, oldest first: this tile's loads, then those of the PD - 1 tiles requested after it (if they exist)
        if (PD == 2 && tile + nbf < a.ntiles && !(CDW_ABLATE & 1)) wait_loads<G::SBX + G::SBY>();
        else wait_loads<0>();
#pragma unroll
        for (int q = 0; q < G::SBX; ++q) land(vx[s][q]);
#pragma unroll
        for (int q = 0; q < G::SBY; ++q) land(vy[s][q]);
        if (!(CDW_ABLATE & 2)) stage_store(tile, vx[s], vy[s]);
        __syncthreads();
        if (tile + PD * nbf < a.ntiles && !(CDW_ABLATE & 1)) stage_load(tile + PD * nbf, vx[s], vy[s]);
        if (!(CDW_ABLATE & 4)) compute(min(a.TC, a.Lc - ((tile / a.nblk) % a.nchunk) * a.TC));
        __syncthreads();                                     // every wave is done with the image
      }
      tile += nbf;
    }
  }
  // slab [flavor][(block in flavor, kq)][tap][ci][local co]: lane holds co = lane & 31 of rows ci = 8*(i>>2) + 4*(lane>>5) + (i&3)
  // slab taps 0..2: kernel size 3, 3..7: 5, 8..14: 7
  float* slab = a.ws + (((int64_t)flavor * nbf + bif) * G::KS + kq) * G::SLAB;
#pragma unroll
  for (int t8 = 0; t8 < 8; ++t8) {
    const int tp = tg == 0 ? 8 + t8 : (t8 < 5 ? 3 + t8 : t8 - 5);
    if (tg == 0 && t8 == 7) continue;                        // kernel size 7 has seven taps
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ci = cib * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
      slab[(tp * 32 * G::NCIB + ci) * (32 * G::CPB) + cob * 32 + (lane & 31)] = acc[t8][i];
    }
  }
}

// dw_j[co][ci][tau] = sum over the slabs of co's flavor, fixed order.  Block = 64 consecutive co of one (tap, ci); the
// 4 waves split the slabs, 16 in flight per lane (clamped loads, see splitk_reduce_kernel).
__global__ __launch_bounds__(256) void conv_dw_reduce_kernel(const float* __restrict__ ws, int nslab, int CI32, int CW,
                                                             int Cout, int Cin, float* __restrict__ dw3,
                                                             float* __restrict__ dw5, float* __restrict__ dw7) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpb = Cout / 64;                                 // blocks per (tap, ci)
  const int cb = blockIdx.x % cpb, ci = (blockIdx.x / cpb) % Cin, tp = blockIdx.x / (cpb * Cin);
  const int co = cb * 64 + lane;
  const int flavor = co / CW, lc = co - flavor * CW;         // CW = columns per flavor (a multiple of 64)
  const int64_t slab = (int64_t)NTAP * CI32 * CW;
  const float* src = ws + (int64_t)flavor * nslab * slab + ((int64_t)tp * CI32 + ci) * CW + lc;
  float v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = 0.f;
  for (int s = wave; s < nslab; s += 64) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int sl = s + 4 * u;
      const float t = src[(int64_t)(sl < nslab ? sl : nslab - 1) * slab];
      v[u] += sl < nslab ? t : 0.f;
    }
  }
#pragma unroll
  for (int u = 8; u > 0; u >>= 1)
#pragma unroll
    for (int w = 0; w < u; ++w) v[w] += v[w + u];
  red[wave][lane] = v[0];
  __syncthreads();
  if (wave == 0) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    // taps 0..2: k = 3 (tau = tp), 3..7: k = 5, 8..14: k = 7
    float* dw = tp < 3 ? dw3 : (tp < 8 ? dw5 : dw7);
    const int k = tp < 3 ? 3 : (tp < 8 ? 5 : 7), tau = tp < 3 ? tp : (tp < 8 ? tp - 3 : tp - 8);
    dw[((int64_t)co * Cin + ci) * k + tau] = t;
  }
}

template <int LD_IN, int COUT, int PD, bool F32>
int launch(const TecmConvDw* p, hipStream_t st) {
  using G = Geo<LD_IN, COUT, F32>;
  Args a;
  a.x = p->inp; a.dy = p->dy; a.ws = p->workspace;
  a.B = p->B; a.Lc = p->Lc; a.N = p->N;
  a.nblk = (p->N + NB - 1) / NB;
  a.TC = p->Lc < TCMAX ? p->Lc : TCMAX;
  a.nchunk = (p->Lc + a.TC - 1) / a.TC;
  const int64_t tiles = (int64_t)p->B * a.nchunk * a.nblk;
  TECM_REQUIRE(tiles < ((int64_t)1 << 31), TECM_E_ARG, "tecm_conv_dw_bf16: too many tiles");
  a.ntiles = (int)tiles;
  const size_t lds = (size_t)G::XBYTES + G::YBYTES;
  static_assert(G::XBYTES + G::YBYTES <= 160 * 1024, "one tile must fit the LDS");
  int nbf = p->num_blocks / G::F;                           // blocks per flavor
  if (nbf > tiles) nbf = (int)tiles;
  TECM_REQUIRE(nbf >= 1, TECM_E_ARG, "tecm_conv_dw_bf16: num_blocks must be at least %d", G::F);
  // per launch, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE (see conv_seq.hip)
  TECM_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dw_seq_kernel<LD_IN, COUT, PD, F32>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
               TECM_E_LAUNCH, "tecm_conv_dw: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  hipLaunchKernelGGL((conv_dw_seq_kernel<LD_IN, COUT, PD, F32>), dim3(nbf * G::F), dim3(NTH), lds, st, a);
  TECM_CHECK_LAUNCH("tecm_conv_dw/seq");
  hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3((unsigned)(NTAP * p->Cin * (COUT / 64))), dim3(256), 0, st, p->workspace,
                     nbf * G::KS, 32 * G::NCIB, 32 * G::CPB, COUT, p->Cin, p->dw3, p->dw5, p->dw7);
  TECM_CHECK_LAUNCH("tecm_conv_dw/reduce");
  return TECM_OK;
}

}  // namespace tecm_convdw

extern "C" int64_t tecm_conv_dw_workspace(int32_t Cout, int32_t ld_in, int32_t num_blocks) {
  if (Cout <= 0 || Cout % 32 != 0 || ld_in <= 0 || ld_in > 64 || num_blocks <= 0) return -1;
  const int ncib = (ld_in + 31) / 32, ncob = Cout / 32;
  const int cpb = ncob < 4 / ncib ? ncob : 4 / ncib, f = ncob / cpb, ks = 4 / (ncib * cpb);
  // [f flavors][num_blocks / f blocks x ks k-ranges][15 taps][32 * ncib][32 * cpb] floats
  return (int64_t)f * (num_blocks / f > 0 ? num_blocks / f : 1) * ks * tecm_convdw::NTAP * 32 * ncib * 32 * cpb;
}

static int conv_dw_launch(const TecmConvDw* p, void* stream, bool f32, const char* who) {
  TECM_REQUIRE(p && p->inp && p->dy && p->workspace && p->dw3 && p->dw5 && p->dw7, TECM_E_ARG, "%s: null pointer", who);
  TECM_REQUIRE(p->B > 0 && p->N > 0 && p->Lc > 0 && p->Lc % 4 == 0, TECM_E_ARG,
               "%s: the sequence length must be a multiple of 4 (got %d)", who, p->Lc);
  TECM_REQUIRE(p->Cin > 0 && p->Cin <= p->ld_in && p->num_blocks > 0, TECM_E_ARG, "%s: bad Cin / num_blocks", who);
  TECM_REQUIRE(tecm_aligned(p->inp, 16) && tecm_aligned(p->dy, 16), TECM_E_ALIGN, "%s: 16-byte aligned tensors", who);
  hipStream_t st = (hipStream_t)stream;
  // PD = register sets in flight: two where they fit WITHOUT spilling (a spilled register with a hand-issued load pending
  // would be stored before the data has landed; __graft_entry__.build() checks ScratchSize of every instantiation).  The
  // fp32 kernel is bound by the matrix cores (11 us of MFMA per tile): one set is plenty.
#define TECM_DW_CASE(LD, CO, PD)                                                              \
  if (p->ld_in == LD && p->Cout == CO)                                                        \
    return f32 ? tecm_convdw::launch<LD, CO, 1, true>(p, st) : tecm_convdw::launch<LD, CO, PD, false>(p, st)
  TECM_DW_CASE(64, 128, 2);
  TECM_DW_CASE(24, 64, 2);
  TECM_DW_CASE(64, 64, 2);
  TECM_DW_CASE(24, 128, 1);                                  // (not a shape of the reference model)
#undef TECM_DW_CASE
  TECM_REQUIRE(false, TECM_E_ARG, "%s: built for ld_in in {24, 64} x Cout in {64, 128} (got %d, %d)", who, p->ld_in, p->Cout);
  return TECM_E_ARG;
}
extern "C" int tecm_conv_dw_bf16(const TecmConvDw* p, void* stream) { return conv_dw_launch(p, stream, false, "tecm_conv_dw_bf16"); }
extern "C" int tecm_conv_dw_f32(const TecmConvDw* p, void* stream) { return conv_dw_launch(p, stream, true, "tecm_conv_dw_f32"); }
