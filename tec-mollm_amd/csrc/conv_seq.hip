// Multi_Scale_Conv_Block (reference modules.py:43-60), bf16 mode: the input gradient of the three parallel Conv1d
// (k = 3, 5, 7) as ONE kernel that reads dy once.
//
//     dinp[b, t, n, ci] = sum_j sum_tau sum_co  dy[b, t - tau + p_j, n, j*Cout + co] * w_j[co, ci, tau],   p_j = (k_j - 1) / 2
//
// Before: three window-view GEMMs (K = 3/5/7 x Cout, N = 24 or 64 output columns, the second and third accumulating into
// dinp), each re-reading its taps of dy through a 256 x 128 tile at one block per CU: 0.82 + 0.72 ms per step in bf16 mode
// at 50-200 TFLOP/s -- bound by data movement, not by the matrix cores.  Here: a SEQUENCE tile.  A block owns 4 nodes x
// (up to 48) time steps of one sample and stages exactly those rows of dy (all 3*Cout channels, bf16) in LDS once; every
// tap of every kernel size is then a row offset inside the tile (rows outside [0, Lc) read a shared zero row), so dy
// leaves HBM once (429 MB at B = 8) and the weights (61 / 246 KB, fragment-ordered by tecm_conv_dx_pack) stream from L2.
//   * matrix cores: v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand (32 input channels x 16 k) and 32 data
//     rows (8 time steps x 4 nodes) as the B operand, so an accumulator lane holds runs of 4 consecutive input channels of
//     ONE row: the result leaves in float4 stores straight from registers, 32 rows x 96 (256) B contiguous per tile;
//   * K = 15 tap-blocks x Cout is split over the block's 4 waves (every wave multiplies its k range against ALL rows of
//     the tile); the four partial sums meet in LDS (the dy image is dead by then) and each accumulator tile is finished
//     and stored by one owner wave -- a fixed order, bit-reproducible;
//   * k order = (time offset delta = -3..3, kernel size j active at delta, co): for a fixed delta the active channels of a
//     dy row are contiguous (all 3*Cout for |delta| <= 1, the upper 2*Cout for 2, the upper Cout for 3), so every
//     16-wide k step is one aligned 32-byte run of one LDS row;
//   * 77 KiB of LDS (pitch 3*Cout*2 + 16 B: conflict-free ds_read_b128) and ~150 registers: two blocks per CU, one
//     loading while the other multiplies.
// Measured (B = 8, N = 2911, round 3): block 1 (Lc 48, Cout 64, 24 columns) 823 -> 220-290 us, block 2 (Lc 24, Cout 128,
// 64 columns) 718 -> 290-350 us; whole bf16 step 23.19 -> 22.26 ms on one box.  Phase ablations (-DCDX_ABLATE): staging +
// stores alone 182 us, the K loop alone 135 us, neither 24 us.  A persistent form that requests the NEXT tile's rows into
// registers before multiplying the current one (72 registers of prefetch) measured the same (217 / 322 us): not kept.
// Arithmetic = the bf16 mode's: operands rounded to bf16 (dy by its producer, the weights by the pack kernel, RNE), fp32
// accumulation; the summation order differs from the GEMM path's (four k ranges), the rounding points do not.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // a native vector: arrays of it stay in registers

namespace tecm_convseq {

#ifndef CDX_ABLATE
#define CDX_ABLATE 0      // diagnostics (tools/build_variant.py): bit0 no staging loads, bit1 no K loop, bit2 no exchange
#endif

constexpr int NB = 4;        // nodes per tile (one MFMA column block = 8 time steps x 4 nodes)
constexpr int NTH = 256;     // 4 waves = 4 k ranges
constexpr int MAXT = 6;      // accumulator tiles per wave: (time steps / 8) x (input-channel blocks of 32)

// k segment d = delta + 3: how many kernel sizes are active, first active channel
__host__ __device__ __forceinline__ int seg_nact(int d) {
  const int a = d < 3 ? 3 - d : d - 3;
  return a <= 1 ? 3 : (a == 2 ? 2 : 1);
}

struct Args {
  const void* dy;            // bf16 or fp32 (B, Lc, N, 3 * Cout)
  const void* wpack;         // fragment-ordered weights, the element type of dy
  float* dinp;
  int B, Lc, N, Cout, ld_in, NCI, TC, nchunk, nblk, pitch, zero_off;   // pitch, zero_off in bytes
  int row_bytes, cpr;        // one dy row: bytes, 16-byte chunks
};

struct Tile {
  int b, n0, t0, tc, ts_lo, rows;
};
__device__ __forceinline__ Tile decode_tile(const Args& a, int tile) {
  Tile t;
  const int nbk = tile % a.nblk;
  tile /= a.nblk;
  const int chunk = tile % a.nchunk;
  t.b = tile / a.nchunk;
  t.n0 = nbk * NB;
  t.t0 = chunk * a.TC;
  t.tc = min(a.TC, a.Lc - t.t0);                           // multiple of 8
  t.ts_lo = max(0, t.t0 - 3);
  t.rows = (min(a.Lc, t.t0 + t.tc + 3) - t.ts_lo) * NB;
  return t;
}

// dy rows of a tile: (ts, n) -> LDS row (ts - ts_lo) * 4 + n, in 16-byte chunks; chunk q of this lane is chunk
// tid + q * NTH of the image.  Loads are clamped (never out of bounds) and only real chunks are stored.
template <int SB>
__device__ __forceinline__ void stage_load(const Args& a, const Tile& t, int q0, u32x4 (&v)[SB]) {
  const int cpr = a.cpr, total = t.rows * cpr, nth = (int)blockDim.x;
  const char* src = static_cast<const char*>(a.dy);
#pragma unroll
  for (int q = 0; q < SB; ++q) {
    const int idx = min((int)threadIdx.x + (q0 + q) * nth, total - 1);
    const int row = idx / cpr, ch = idx - row * cpr;
    const int ts = t.ts_lo + (row >> 2);
    const int ng = min(t.n0 + (row & 3), a.N - 1);         // the ragged last node block re-reads node N-1; never stored
    v[q] = *reinterpret_cast<const u32x4*>(src + (((int64_t)t.b * a.Lc + ts) * a.N + ng) * a.row_bytes + ch * 16);
  }
}
template <int SB>
__device__ __forceinline__ void stage_store(const Args& a, const Tile& t, int q0, const u32x4 (&v)[SB], unsigned char* lds) {
  const int cpr = a.cpr, total = t.rows * cpr, nth = (int)blockDim.x;
#pragma unroll
  for (int q = 0; q < SB; ++q) {
    const int idx = (int)threadIdx.x + (q0 + q) * nth;
    const int row = idx / cpr, ch = idx - row * cpr;
    if (idx < total) *reinterpret_cast<u32x4*>(lds + row * a.pitch + ch * 16) = v[q];
  }
}

// The tile's product: this wave's k range against all rows (accumulators), then the exchange and the stores.
template <int NCI>
__device__ __forceinline__ void tile_compute(const Args& a, const Tile& t, unsigned char* lds, int wave, int lane,
                                             bool) {
  const int r = lane & 31, h = lane >> 5;
  const int zero_off = a.zero_off;
  // ---- this wave's k range, in 16-wide steps; (d, sd) = (segment, step inside the segment)
  const int spc = a.Cout / 16;                             // k steps per kernel size and time offset
  const int KS = 15 * spc;
  const int s_beg = wave * (KS / 4), s_end = s_beg + KS / 4;
  int d = 0, sd = s_beg;
  while (sd >= seg_nact(d) * spc) {
    sd -= seg_nact(d) * spc;
    ++d;
  }
  const int ntt = t.tc >> 3;                               // row tiles (8 time steps x 4 nodes) of this chunk
  f32x16 acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // lane r of row tile i is data row (tl = 8 i + r / 4, n = r % 4); its source row at offset delta is LDS row
  // (t0 + tl + delta - ts_lo) * 4 + n = r + 4 * (t0 - ts_lo + 8 i + delta)
  const int tl0 = r >> 2;
  // weight fragments come straight from L2 (fragment-ordered, 1 KiB per wave-instruction): a ring of WD steps in
  // registers, refilled WD steps ahead -- one step (6 MFMAs) is far shorter than an L2 round trip
  constexpr int WD = 3;                                    // KS / 4 = 15 * Cout / 64 is a multiple of 15
  const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wpack) + (int64_t)s_beg * NCI * 64 + lane;
  bf16x8 wf[WD][NCI];
#pragma unroll
  for (int u = 0; u < WD; ++u)
#pragma unroll
    for (int c = 0; c < NCI; ++c) wf[u][c] = wp[(u * NCI + c) * 64];
  for (int s0 = s_beg; s0 < ((CDX_ABLATE & 2) ? s_beg : s_end); s0 += WD) {
#pragma unroll
    for (int u = 0; u < WD; ++u) {
      const int delta = d - 3;
      const int colb = ((3 - seg_nact(d)) * a.Cout + sd * 16 + 8 * h) * 2;    // byte offset inside the dy row
      const int rbase = (t.t0 - t.ts_lo + delta) * 4 + r;
#pragma unroll
      for (int i = 0; i < MAXT / NCI; ++i) {
        if (i < ntt) {
          const int ts = t.t0 + 8 * i + tl0 + delta;
          const bool ok = ts >= 0 && ts < a.Lc;
          const int off = ok ? (rbase + 32 * i) * a.pitch + colb : zero_off + 16 * h;
          const bf16x8 df = *reinterpret_cast<const bf16x8*>(lds + off);
#pragma unroll
          for (int c = 0; c < NCI; ++c)
            acc[i * NCI + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u][c], df, acc[i * NCI + c], 0, 0, 0);
        }
      }
      if (s0 + u + WD < s_end) {
#pragma unroll
        for (int c = 0; c < NCI; ++c) wf[u][c] = wp[((u + WD) * NCI + c) * 64];
      }
      if (++sd == seg_nact(d) * spc) {
        sd = 0;
        ++d;
      }
    }
    wp += WD * NCI * 64;
  }
  __syncthreads();                                         // the dy image is dead: it becomes the exchange buffer

  // ---- the four k ranges meet: tile T is finished by wave T % 4; the other three park their partial sums in
  //      part[T][slot][e][lane] (slot = (wave - owner - 1) & 3 in 0..2), conflict-free both ways
  float* part = reinterpret_cast<float*>(lds);
  const int NT = ntt * NCI;
#pragma unroll
  for (int T = 0; T < MAXT; ++T) {
    if (T < NT && (T & 3) != wave && !(CDX_ABLATE & 4)) {
      const int slot = (wave - (T & 3) - 1) & 3;
#pragma unroll
      for (int e = 0; e < 16; ++e) part[((T * 3 + slot) * 16 + e) * 64 + lane] = acc[T][e];
    }
  }
  __syncthreads();
#pragma unroll
  for (int T = 0; T < MAXT; ++T) {
    if (T < NT && (T & 3) == wave) {
      f32x16 v = acc[T];
#pragma unroll
      for (int slot = 0; slot < ((CDX_ABLATE & 4) ? 0 : 3); ++slot)
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += part[((T * 3 + slot) * 16 + e) * 64 + lane];
      // accumulator register e of lane (r, h): input channel 32 c + (e & 3) + 8 (e >> 2) + 4 h of data row r
      const int i = T / NCI, c = T % NCI;
      const int tt = t.t0 + 8 * i + tl0, n = t.n0 + (r & 3);
      if (n < a.N) {
        float* orow = a.dinp + (((int64_t)t.b * a.Lc + tt) * a.N + n) * a.ld_in;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int ci = 32 * c + 8 * gq + 4 * h;
          if (ci < a.ld_in)
            *reinterpret_cast<float4*>(orow + ci) = make_float4(v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]);
        }
      }
    }
  }
}

// One tile per block (any shape the LDS admits).
template <int NCI>
__global__ __launch_bounds__(NTH, 2) void conv_dx_seq_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Tile t = decode_tile(a, blockIdx.x);
  {
    constexpr int SB = 12;                                 // loads in flight per lane (the accumulators are not live yet)
    const int total = t.rows * a.cpr;
    for (int q0 = 0; q0 * NTH < ((CDX_ABLATE & 1) ? 0 : total); q0 += SB) {
      u32x4 v[SB];
      stage_load<SB>(a, t, q0, v);
      stage_store<SB>(a, t, q0, v, lds);
    }
    if (tid < a.pitch / 16) *reinterpret_cast<uint4*>(lds + a.zero_off + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  tile_compute<NCI>(a, t, lds, wave, lane, false);
}

// ------------------------------------------------------------------------------------------------------------------
// The same kernel in EXACT fp32 (BASELINE configs[1], the headline): dy fp32, v_mfma_f32_32x32x2_f32 (bit-identical to an
// fmaf chain), 8 waves = 8 k ranges, the fp32 image of 4 nodes x Lc steps takes 150 KiB (one block per CU).  Replaces
// three accumulating window GEMMs that ran at 57 (block 1: 24 of 32 tile columns used) and 87 TFLOP/s (block 2).
//   * k runs in steps of 8: a lane reads ONE float4 = k 8s + 4h .. 8s + 4h + 3 of its data row and feeds MFMA j (j = 0..3)
//     with element j, i.e. MFMA j multiplies the k pair (8s + j, 8s + 4 + j); the weights are packed in the same order;
//   * the eight partial sums meet in two rounds: waves 4..7 park theirs for waves 0..3 (96 KiB), then the scheme of the
//     bf16 kernel among waves 0..3.
constexpr int NTH32 = 512;

template <int NCI>
__global__ __launch_bounds__(NTH32, 2) void conv_dx_seq_f32_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const Tile t = decode_tile(a, blockIdx.x);
  {
    constexpr int SB = 10;
    const int total = t.rows * a.cpr;
    for (int q0 = 0; q0 * NTH32 < total; q0 += SB) {
      u32x4 v[SB];
      stage_load<SB>(a, t, q0, v);
      stage_store<SB>(a, t, q0, v, lds);
    }
    if (tid < a.pitch / 16) *reinterpret_cast<uint4*>(lds + a.zero_off + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();

  const int spc = a.Cout / 8;                              // 8-wide k steps per kernel size and time offset
  const int KS = 15 * spc;
  const int s_beg = wave * (KS / 8), s_end = s_beg + KS / 8;
  int d = 0, sd = s_beg;
  while (sd >= seg_nact(d) * spc) {
    sd -= seg_nact(d) * spc;
    ++d;
  }
  const int ntt = t.tc >> 3;
  f32x16 acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const int tl0 = r >> 2;
  constexpr int WD = 3;                                    // KS / 8 = 15 * Cout / 64 is a multiple of 15
  const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack) + (int64_t)s_beg * NCI * 64 + lane;
  f32x4 wf[WD][NCI];
#pragma unroll
  for (int u = 0; u < WD; ++u)
#pragma unroll
    for (int c = 0; c < NCI; ++c) wf[u][c] = wp[(u * NCI + c) * 64];
  for (int s0 = s_beg; s0 < s_end; s0 += WD) {
#pragma unroll
    for (int u = 0; u < WD; ++u) {
      const int delta = d - 3;
      const int colb = ((3 - seg_nact(d)) * a.Cout + sd * 8 + 4 * h) * 4;     // byte offset inside the dy row
      const int rbase = (t.t0 - t.ts_lo + delta) * 4 + r;
#pragma unroll
      for (int i = 0; i < MAXT / NCI; ++i) {
        if (i < ntt) {
          const int ts = t.t0 + 8 * i + tl0 + delta;
          const bool ok = ts >= 0 && ts < a.Lc;
          const int off = ok ? (rbase + 32 * i) * a.pitch + colb : a.zero_off + 16 * h;
          const f32x4 df = *reinterpret_cast<const f32x4*>(lds + off);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < NCI; ++c)
              acc[i * NCI + c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[u][c][j], df[j], acc[i * NCI + c], 0, 0, 0);
        }
      }
      if (s0 + u + WD < s_end) {
#pragma unroll
        for (int c = 0; c < NCI; ++c) wf[u][c] = wp[((u + WD) * NCI + c) * 64];
      }
      if (++sd == seg_nact(d) * spc) {
        sd = 0;
        ++d;
      }
    }
    wp += WD * NCI * 64;
  }
  __syncthreads();                                         // the dy image is dead: it becomes the exchange buffer

  float* part = reinterpret_cast<float*>(lds);
  const int NT = ntt * NCI;
  // round 1: waves 4..7 -> waves 0..3 (wave w adds the sums of wave w + 4): part[w - 4][T][e][lane]
  if (wave >= 4) {
#pragma unroll
    for (int T = 0; T < MAXT; ++T)
      if (T < NT)
#pragma unroll
        for (int e = 0; e < 16; ++e) part[(((wave - 4) * NT + T) * 16 + e) * 64 + lane] = acc[T][e];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int T = 0; T < MAXT; ++T)
      if (T < NT)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[T][e] += part[((wave * NT + T) * 16 + e) * 64 + lane];
  }
  __syncthreads();
  // round 2: among waves 0..3, tile T is finished by wave T % 4 (as in the bf16 kernel)
  if (wave < 4) {
#pragma unroll
    for (int T = 0; T < MAXT; ++T) {
      if (T < NT && (T & 3) != wave) {
        const int slot = (wave - (T & 3) - 1) & 3;
#pragma unroll
        for (int e = 0; e < 16; ++e) part[((T * 3 + slot) * 16 + e) * 64 + lane] = acc[T][e];
      }
    }
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int T = 0; T < MAXT; ++T) {
      if (T < NT && (T & 3) == wave) {
        f32x16 v = acc[T];
#pragma unroll
        for (int slot = 0; slot < 3; ++slot)
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] += part[((T * 3 + slot) * 16 + e) * 64 + lane];
        const int i = T / NCI, c = T % NCI;
        const int tt = t.t0 + 8 * i + tl0, n = t.n0 + (r & 3);
        if (n < a.N) {
          float* orow = a.dinp + (((int64_t)t.b * a.Lc + tt) * a.N + n) * a.ld_in;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int ci = 32 * c + 8 * gq + 4 * h;
            if (ci < a.ld_in)
              *reinterpret_cast<float4*>(orow + ci) = make_float4(v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]);
          }
        }
      }
    }
  }
}

// fp32 weights in the order of conv_dx_seq_f32_kernel: wpack[s][c][lane][j] = w_j[co][ci][tau] for k = 8 s + 4 (lane >> 5) + j
__global__ __launch_bounds__(256) void conv_dx_pack_f32_kernel(const float* __restrict__ w3, const float* __restrict__ w5,
                                                               const float* __restrict__ w7, float* __restrict__ wpack,
                                                               int Cout, int Cin, int NCI) {
  const int total = 15 * Cout * NCI * 32;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int j = i & 3, lane = (i >> 2) & 63;
    const int sc = i >> 8, c = sc % NCI, s = sc / NCI;
    int k = 8 * s + 4 * (lane >> 5) + j;
    const int ci = 32 * c + (lane & 31);
    int d = 0;
    while (k >= seg_nact(d) * Cout) {
      k -= seg_nact(d) * Cout;
      ++d;
    }
    const int col = (3 - seg_nact(d)) * Cout + k;
    const int jj = col / Cout, co = col - jj * Cout;
    const int kj = 3 + 2 * jj, tau = (kj - 1) / 2 - (d - 3);
    const float* w = jj == 0 ? w3 : (jj == 1 ? w5 : w7);
    float v = 0.f;
    if (ci < Cin && tau >= 0 && tau < kj) v = w[((int64_t)co * Cin + ci) * kj + tau];
    wpack[i] = v;
  }
}

// wpack[s][c][lane][e] = w_j[co][ci][tau]  for k = 16 s + 8 (lane >> 5) + e  -> (delta, j, co), tau = p_j - delta,
// ci = 32 c + (lane & 31); zero for ci >= Cin (the 22 -> 24 channel padding and the tail of the last channel block)
__global__ __launch_bounds__(256) void conv_dx_pack_kernel(const float* __restrict__ w3, const float* __restrict__ w5,
                                                           const float* __restrict__ w7, __bf16* __restrict__ wpack,
                                                           int Cout, int Cin, int NCI) {
  const int KTOT = 15 * Cout;
  const int total = KTOT * NCI * 32;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int e = i & 7, lane = (i >> 3) & 63;
    const int sc = i >> 9, c = sc % NCI, s = sc / NCI;
    int k = 16 * s + 8 * (lane >> 5) + e;
    const int ci = 32 * c + (lane & 31);
    int d = 0;
    while (k >= seg_nact(d) * Cout) {
      k -= seg_nact(d) * Cout;
      ++d;
    }
    const int col = (3 - seg_nact(d)) * Cout + k;          // channel of the dy row: kernel size j = col / Cout
    const int j = col / Cout, co = col - j * Cout;
    const int kj = 3 + 2 * j, tau = (kj - 1) / 2 - (d - 3);
    const float* w = j == 0 ? w3 : (j == 1 ? w5 : w7);
    float v = 0.f;
    if (ci < Cin && tau >= 0 && tau < kj) v = w[((int64_t)co * Cin + ci) * kj + tau];
    wpack[i] = (__bf16)v;
  }
}

}  // namespace tecm_convseq

// ------------------------------------------------------------------------------------------------------------------
// Forward of the three parallel Conv1d (k = 3, 5, 7) of a Multi_Scale_Conv_Block, bf16 mode, ONE launch:
//     y[b, t, n, j*Cout + co] = bias_j[co] + sum_tau sum_ci inp[b, t + tau - p_j, n, ci] * w_j[co, ci, tau]
// Before: three window GEMMs, each writing its own 64- (128-) column slice of y (256-B pieces of 768-B rows) and each
// re-reading its taps of the input.  Here a block stages the (4 nodes x Lc + 7 halo steps) input rows once (bf16, 10-17 KiB,
// halo rows zero) and writes whole rows of y; the data rows are the MFMA A operand (8 time steps x 4 nodes = 32 rows), the
// weights (fragment-ordered by tecm_conv_fwd_pack, streamed from L2) the B operand, so a lane holds one output channel and
// a store instruction writes 2 x 128 contiguous bytes.  Work is dealt to the 4 waves by (kernel size, 32-channel block)
// units of 5..28 k steps, balanced on the host; each unit keeps (time steps / 8) accumulator tiles.
//   k order inside a kernel size = (tap, ci); a 16-wide k step is two 8-channel chunks of (possibly) two taps: the chunk a
//   lane reads is q = 2 s + h -> tap q / (ld_in / 8), channel 8 (q % (ld_in / 8)); chunks past the last tap meet zero
//   weights and read the (zeroed) extra halo step.
namespace tecm_convseq {

// Sum over the 64 lanes on the DPP network (quad swaps, row mirrors, row broadcasts; the total arrives in lane 63 and is
// read into a scalar): no LDS crossbar traffic (__shfl_xor is ds_bpermute) in a kernel that lives on its LDS reads.
__device__ __forceinline__ float wave_total_dpp(float v) {
  auto step = [](float x, auto ctrl, auto rmask) {
    constexpr int C = decltype(ctrl)::value, R = decltype(rmask)::value;
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C, R, 0xF, false));
  };
  v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});     // quad_perm [1,0,3,2]
  v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});     // quad_perm [2,3,0,1]
  v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});    // row_half_mirror
  v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});    // row_mirror: every lane = its row's sum
  v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});    // row_bcast15 into rows 1, 3
  v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});    // row_bcast31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

constexpr int FMAXU = 12;      // units: 3 kernel sizes x Cout / 32 channel blocks (Cout <= 128)
#ifndef CFW_PD
#define CFW_PD 4               // weight-prefetch depth of the forward K loops, in k-steps (tools/build_variant.py sweeps it)
#endif
#ifndef CFW_ABLATE
#define CFW_ABLATE 0           // diagnostics (tools/build_variant.py): bit0 no staging loads, bit1 no K loop, bit2 no y stores
#endif
#ifndef CFW_PD32
#define CFW_PD32 1             // the same for the exact-fp32 kernel (a k-step there is 4 NTT MFMAs of 64 cycles: already long)
#endif
struct FArgs {
  const void* inp;             // bf16 or fp32 (B, Lc, N, ld_in)
  const void* wpack;           // fragment-ordered weights, the element type of inp
  const float* bias;           // [3 * Cout]: b3 | b5 | b7
  float* y;                    // fp32 (B, Lc, N, 3 * Cout) -- or bf16 when y16 (the bf16 kernel only)
  int y16;
  float* stats;                // y16 and whole sequences per tile: GroupNorm(1) statistics (B*N, 3, 2) of the rounded y
  float eps;
  int B, Lc, N, Cout, ld_in, TC, nchunk, nblk, pitch;
  int nb32;                    // Cout / 32: unit u = (kernel size j = u / nb32, channel block u % nb32)
  unsigned long long assign;   // 4 waves x 3 unit ids of 4 bits (15 = none): no kernel-argument arrays -- indexing
};                             //   one dynamically makes the compiler copy the whole struct to scratch

#ifdef CFW_STAMPS          // diagnostics (tools/build_variant.py): 10 ns time stamps of wave 0 -- entry, image staged, per unit K loop / stores
__device__ unsigned long long cfw_stamps[16 * 8192];
}  // namespace tecm_convseq
extern "C" int tecm_cfw_stamps_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tecm_convseq::cfw_stamps), sizeof(unsigned long long) * n);
}
namespace tecm_convseq {
#define CFW_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) cfw_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CFW_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(NTH, 3) void conv_fwd_seq_kernel(const FArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  CFW_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int tile = blockIdx.x;
  const int nbk = tile % a.nblk;
  tile /= a.nblk;
  const int chunk = tile % a.nchunk, b = tile / a.nchunk;
  const int n0 = nbk * NB, t0 = chunk * a.TC;
  const int tc = min(a.TC, a.Lc - t0);                     // multiple of 8
  const int steps = tc + 7;                                // image time steps: t0 - 3 .. t0 + tc + 3 (+1: k padding)
  const int CT = 3 * a.Cout;

  // ---- stage: image row (ts - (t0 - 3)) * 4 + n, 16-byte chunks; rows outside [0, Lc) are zeros
  {
    const int cpr = a.ld_in / 8;
    const int total = steps * NB * cpr;
    for (int idx = tid; idx < total; idx += NTH) {
      const int row = idx / cpr, ch = idx - row * cpr;
      const int ts = t0 - 3 + (row >> 2);
      const int ng = min(n0 + (row & 3), a.N - 1);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (!(CFW_ABLATE & 1) && ts >= 0 && ts < a.Lc)
        v = *reinterpret_cast<const u32x4*>(static_cast<const __bf16*>(a.inp) + (((int64_t)b * a.Lc + ts) * a.N + ng) * a.ld_in + ch * 8);
      *reinterpret_cast<u32x4*>(lds + row * a.pitch + ch * 16) = v;
    }
  }
  __syncthreads();
  CFW_STAMP(1);

  const int ntt = tc >> 3;
  const int cpt = a.ld_in / 8;                             // chunks per tap
  const unsigned cpt_magic = (unsigned)((65536 + cpt - 1) / cpt);      // q / cpt for q < 256
  // GroupNorm statistics: every unit leaves (mean, M2) of its 32 channels per node in LDS (no running registers: they cost
  // the kernel its third wave per SIMD); twelve lanes merge them with Chan's rule at the end, in a fixed order
  __shared__ float xst[4][3][2 * NB + 1];                  // [wave][unit slot]: branch (-1: none), 4 means, 4 M2s
  if (a.stats && lane < 3) xst[wave][lane][0] = -1.f;
#pragma unroll 1
  for (int ui = 0; ui < 3; ++ui) {
    const int u = (int)((a.assign >> (12 * wave + 4 * ui)) & 15ull);
    if (u == 15) break;
    const int j = u / a.nb32, kb = u - j * a.nb32;
    const int pad = j + 1, col0 = j * a.Cout + 32 * kb;
    const int st3 = (3 * a.ld_in + 15) / 16, st5 = (5 * a.ld_in + 15) / 16, st7 = (7 * a.ld_in + 15) / 16;
    const int nsteps = j == 0 ? st3 : (j == 1 ? st5 : st7);
    const int woff = (j == 0 ? 0 : (j == 1 ? st3 : st3 + st5)) * a.nb32 + kb * nsteps;
    f32x16 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wpack) + (int64_t)woff * 64 + lane;
    // Operands of step s+1 (weights from L2, data rows from LDS) are requested while step s multiplies.  Every load is
    // UNCONDITIONAL on a clamped index: a wave-uniform `if` around a load is a branch, and the compiler waits for the
    // load inside it (vmcnt(0) / lgkmcnt(0) per load: the first version exposed the L2 latency of the weights at every
    // k-step and the LDS latency in front of every MFMA).
    // (the row-tile count is dispatched to a compile-time constant: a run-time `if (i < ntt)` around each MFMA puts every
    // one of them into its own basic block)
    auto kloop = [&](auto ntc) {
      constexpr int NTT = decltype(ntc)::value;
      auto rd = [&](int s, bf16x8 (&df)[NTT]) {
        const int q = 2 * s + h;
        const int tap = (int)((q * cpt_magic) >> 16), c8 = q - tap * cpt;
        // data row r of row tile i at this tap: image row (8 i + tap - pad + 3) * 4 + r
        const int off0 = ((tap - pad + 3) * 4 + r) * a.pitch + c8 * 16;
#pragma unroll
        for (int i = 0; i < NTT; ++i) df[i] = *reinterpret_cast<const bf16x8*>(lds + off0 + 32 * i * a.pitch);
      };
      // weights: a register ring CFW_PD steps deep (a k-step is NTT MFMAs = 100-200 cycles, an L2 hit under load costs
      // 300-500: one step of prefetch left every step waiting for its weights); data rows: one step ahead (LDS)
      bf16x8 wq[CFW_PD], df[NTT];
#pragma unroll
      for (int p_ = 0; p_ < CFW_PD; ++p_) wq[p_] = wp[min(p_, nsteps - 1) * 64];
      rd(0, df);
      for (int s = 0; s < nsteps; ++s) {
        const int sn = min(s + 1, nsteps - 1);
        const bf16x8 wn = wp[min(s + CFW_PD, nsteps - 1) * 64];
        bf16x8 dn[NTT];
        rd(sn, dn);
#pragma unroll
        for (int i = 0; i < NTT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[i], wq[0], acc[i], 0, 0, 0);
#pragma unroll
        for (int p_ = 0; p_ + 1 < CFW_PD; ++p_) wq[p_] = wq[p_ + 1];
        wq[CFW_PD - 1] = wn;
#pragma unroll
        for (int i = 0; i < NTT; ++i) df[i] = dn[i];
      }
    };
    if (!(CFW_ABLATE & 2)) switch (ntt) {
      case 1: kloop(std::integral_constant<int, 1>{}); break;
      case 2: kloop(std::integral_constant<int, 2>{}); break;
      case 3: kloop(std::integral_constant<int, 3>{}); break;
      case 4: kloop(std::integral_constant<int, 4>{}); break;
      case 5: kloop(std::integral_constant<int, 5>{}); break;
      default: kloop(std::integral_constant<int, MAXT>{}); break;
    }
    CFW_STAMP(2 + 2 * ui);                                 // wave 0: this unit's K loop done
    if (CFW_ABLATE & 4) {                                  // keep the accumulators alive without storing them
      float keep = 0.f;
#pragma unroll
      for (int i = 0; i < MAXT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) keep += acc[i][e];
      if (keep == 12345.678f) a.y[0] = keep;
      continue;
    }
    // ---- this unit's 32 channels of every row of the tile: accumulator register e of lane (co, h) is data row
    //      (e & 3) + 8 (e >> 2) + 4 h of its row tile, i.e. time step 8 i + 2 (e >> 2) + h, node e & 3
    const float bv = a.bias[col0 + r];
    const int tstride = a.N * CT;                          // one time step, in elements (a tile spans < 2^31 of them)
    if (a.stats) {
      // statistics of the values AS STORED: the accumulators are replaced by the rounded y once (the store below converts
      // exactly); the sums themselves are formed AFTER the stores have been issued, under their latency
#pragma unroll
      for (int i = 0; i < MAXT; ++i)
        if (i < ntt) {
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][e] = (float)(__bf16)(acc[i][e] + bv);
        }
    }
    const float bvs = a.stats ? 0.f : bv;                  // the bias is already inside the rounded accumulators
    if (a.y16) {
      // y as the bf16 tensor a bf16 Conv1d produces under autocast (train.py:68): neighbouring lanes (channels c, c + 1)
      // trade one value per register pair, so that the even lane stores (c, c + 1) of node e0 & 3 and the odd lane
      // (c - 1, c) of the next node as ONE 4-byte word each -- half the store instructions of the fp32 form, half the bytes
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      const bool odd = (lane & 1) != 0;
      __bf16* yh = reinterpret_cast<__bf16*>(a.y) + (((int64_t)b * a.Lc + t0 + h) * a.N + n0) * CT + col0 + (r & ~1);
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        if (i < ntt) {
#pragma unroll
          for (int ep = 0; ep < 8; ++ep) {
            const float v0 = acc[i][2 * ep] + bvs, v1 = acc[i][2 * ep + 1] + bvs;
            const float send = odd ? v0 : v1;
            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, false));
            bf16x2 pk;
            pk[0] = (__bf16)(odd ? recv : v0);
            pk[1] = (__bf16)(odd ? v1 : recv);
            const int e = 2 * ep + (odd ? 1 : 0), n = e & 3;
            if (n0 + n < a.N) *reinterpret_cast<bf16x2*>(yh + (8 * i + 2 * (e >> 2)) * tstride + n * CT) = pk;
          }
        }
      }
      CFW_STAMP(3 + 2 * ui);                               // wave 0: this unit's stores issued
      if (a.stats) {
        // one pass of shifted sums -- shift = the wave's first value, so that sum((v - s)^2) - sum(v - s)^2 / n does not
        // cancel when |mean| >> std -- and eight wave reductions per unit
        float s1[NB] = {0.f, 0.f, 0.f, 0.f}, s2[NB] = {0.f, 0.f, 0.f, 0.f};
        const float sh = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[0][0])));
#pragma unroll
        for (int i = 0; i < MAXT; ++i)
          if (i < ntt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float dv = acc[i][e] - sh;
              s1[e & 3] += dv;
              s2[e & 3] += dv * dv;
            }
          }
        const float ucnt = (float)(32 * tc);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const float t1 = wave_total_dpp(s1[n]), t2 = wave_total_dpp(s2[n]);
          if (lane == 0) {
            xst[wave][ui][1 + n] = sh + t1 / ucnt;
            xst[wave][ui][1 + NB + n] = fmaxf(t2 - t1 * t1 / ucnt, 0.f);
          }
        }
        if (lane == 0) xst[wave][ui][0] = (float)j;
      }
      continue;
    }
    float* yb = a.y + (((int64_t)b * a.Lc + t0 + h) * a.N + n0) * CT + col0 + r;       // (time t0 + h, node n0)
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      if (i < ntt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = e & 3;
          if (n0 + n < a.N) yb[(8 * i + 2 * (e >> 2)) * tstride + n * CT] = acc[i][e] + bv;
        }
      }
    }
  }
  CFW_STAMP(8);
  if (a.stats) {
    __syncthreads();
    if (tid < 3 * NB) {
      const int jb = tid / NB, n = tid % NB;
      const float ucnt = (float)(32 * tc);
      float cnt = 0.f, mean = 0.f, m2 = 0.f;
      for (int w = 0; w < 4; ++w)
        for (int q = 0; q < 3; ++q)
          if (xst[w][q][0] == (float)jb) {
            const float tot = cnt + ucnt, fr = ucnt / tot, dl = xst[w][q][1 + n] - mean;
            m2 += xst[w][q][1 + NB + n] + dl * dl * cnt * fr;
            mean += dl * fr;
            cnt = tot;
          }
      if (n0 + n < a.N) {
        float* st = a.stats + (((int64_t)b * a.N + n0 + n) * 3 + jb) * 2;
        st[0] = mean;
        st[1] = 1.0f / sqrtf(m2 / cnt + a.eps);
      }
    }
  }
}

// The forward in EXACT fp32 (BASELINE configs[1]; also the eval / inference path): input fp32, v_mfma_f32_32x32x2_f32, k in
// steps of 8 (a lane reads one float4 = k 8 s + 4 h + 0..3 and feeds MFMA j with element j; the weights are packed in the
// same order).  k_j * ld_in is a multiple of 8 for every kernel size, so there is no k padding and no extra halo step.
__global__ __launch_bounds__(NTH, 2) void conv_fwd_seq_f32_kernel(const FArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int tile = blockIdx.x;
  const int nbk = tile % a.nblk;
  tile /= a.nblk;
  const int chunk = tile % a.nchunk, b = tile / a.nchunk;
  const int n0 = nbk * NB, t0 = chunk * a.TC;
  const int tc = min(a.TC, a.Lc - t0);
  const int steps = tc + 7;
  const int CT = 3 * a.Cout;
  {
    const int cpr = a.ld_in / 4;                           // 16-byte chunks per fp32 row
    const int total = steps * NB * cpr;
    const float* src = static_cast<const float*>(a.inp);
    for (int idx = tid; idx < total; idx += NTH) {
      const int row = idx / cpr, ch = idx - row * cpr;
      const int ts = t0 - 3 + (row >> 2);
      const int ng = min(n0 + (row & 3), a.N - 1);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ts >= 0 && ts < a.Lc)
        v = *reinterpret_cast<const u32x4*>(src + (((int64_t)b * a.Lc + ts) * a.N + ng) * a.ld_in + ch * 4);
      *reinterpret_cast<u32x4*>(lds + row * a.pitch + ch * 16) = v;
    }
  }
  __syncthreads();

  const int ntt = tc >> 3;
  const int cpt = a.ld_in / 4;                             // float4 chunks per tap
  const unsigned cpt_magic = (unsigned)((65536 + cpt - 1) / cpt);
#pragma unroll 1
  for (int ui = 0; ui < 3; ++ui) {
    const int u = (int)((a.assign >> (12 * wave + 4 * ui)) & 15ull);
    if (u == 15) break;
    const int j = u / a.nb32, kb = u - j * a.nb32;
    const int pad = j + 1, col0 = j * a.Cout + 32 * kb;
    const int st3 = 3 * a.ld_in / 8, st5 = 5 * a.ld_in / 8, st7 = 7 * a.ld_in / 8;
    const int nsteps = j == 0 ? st3 : (j == 1 ? st5 : st7);
    const int woff = (j == 0 ? 0 : (j == 1 ? st3 : st3 + st5)) * a.nb32 + kb * nsteps;
    f32x16 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack) + (int64_t)woff * 64 + lane;
    // as the bf16 kernel: unconditional clamped loads one step ahead; the four MFMAs a float4 feeds go to the SAME
    // accumulator, so they are issued tile-interleaved (jj outer, row tile inner) -- back-to-back they would each wait for
    // the previous one's result
    auto kloop = [&](auto ntc) {
      constexpr int NTT = decltype(ntc)::value;
      auto rd = [&](int s, f32x4 (&df)[NTT]) {
        const int q = 2 * s + h;
        const int tap = (int)((q * cpt_magic) >> 16), c4 = q - tap * cpt;
        const int off0 = ((tap - pad + 3) * 4 + r) * a.pitch + c4 * 16;
#pragma unroll
        for (int i = 0; i < NTT; ++i) df[i] = *reinterpret_cast<const f32x4*>(lds + off0 + 32 * i * a.pitch);
      };
      f32x4 wq[CFW_PD32], df[NTT];
#pragma unroll
      for (int p_ = 0; p_ < CFW_PD32; ++p_) wq[p_] = wp[min(p_, nsteps - 1) * 64];
      rd(0, df);
      for (int s = 0; s < nsteps; ++s) {
        const int sn = min(s + 1, nsteps - 1);
        const f32x4 wn = wp[min(s + CFW_PD32, nsteps - 1) * 64];
        f32x4 dn[NTT];
        rd(sn, dn);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int i = 0; i < NTT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(df[i][jj], wq[0][jj], acc[i], 0, 0, 0);
#pragma unroll
        for (int p_ = 0; p_ + 1 < CFW_PD32; ++p_) wq[p_] = wq[p_ + 1];
        wq[CFW_PD32 - 1] = wn;
#pragma unroll
        for (int i = 0; i < NTT; ++i) df[i] = dn[i];
      }
    };
    switch (ntt) {
      case 1: kloop(std::integral_constant<int, 1>{}); break;
      case 2: kloop(std::integral_constant<int, 2>{}); break;
      case 3: kloop(std::integral_constant<int, 3>{}); break;
      case 4: kloop(std::integral_constant<int, 4>{}); break;
      case 5: kloop(std::integral_constant<int, 5>{}); break;
      default: kloop(std::integral_constant<int, MAXT>{}); break;
    }
    const float bv = a.bias[col0 + r];
    float* yb = a.y + (((int64_t)b * a.Lc + t0 + h) * a.N + n0) * CT + col0 + r;
    const int tstride = a.N * CT;
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      if (i < ntt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = e & 3;
          if (n0 + n < a.N) yb[(8 * i + 2 * (e >> 2)) * tstride + n * CT] = acc[i][e] + bv;
        }
      }
    }
  }
}

// fp32 weights in the order of conv_fwd_seq_f32_kernel: k = 8 s + 4 (lane >> 5) + jj = tap * ld_in + ci
__global__ __launch_bounds__(256) void conv_fwd_pack_f32_kernel(const float* __restrict__ w3, const float* __restrict__ w5,
                                                                const float* __restrict__ w7, float* __restrict__ wpack,
                                                                int Cout, int Cin, int ld_in) {
  const int nblk = Cout / 32;
  int steps[3], base[3], tot = 0;
  for (int j = 0; j < 3; ++j) {
    steps[j] = (3 + 2 * j) * ld_in / 8;
    base[j] = tot;
    tot += steps[j] * nblk;
  }
  const int total = tot * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int jj = i & 3, lane = (i >> 2) & 63, f = i >> 8;
    int j = 2;
    if (f < base[1]) j = 0; else if (f < base[2]) j = 1;
    const int fl = f - base[j], blk = fl / steps[j], s = fl - blk * steps[j];
    const int k = 8 * s + 4 * (lane >> 5) + jj;
    const int tap = k / ld_in, ci = k - tap * ld_in, kj = 3 + 2 * j;
    const float* w = j == 0 ? w3 : (j == 1 ? w5 : w7);
    float v = 0.f;
    if (tap < kj && ci < Cin) v = w[((int64_t)(32 * blk + (lane & 31)) * Cin + ci) * kj + tap];
    wpack[i] = v;
  }
}

// wpack[unit][s][lane][e] = w_j[32 blk + (lane & 31)][ci][tap] for k = 16 s + 8 (lane >> 5) + e = tap * ld_in + ci
__global__ __launch_bounds__(256) void conv_fwd_pack_kernel(const float* __restrict__ w3, const float* __restrict__ w5,
                                                            const float* __restrict__ w7, __bf16* __restrict__ wpack,
                                                            int Cout, int Cin, int ld_in) {
  const int nblk = Cout / 32;
  int steps[3], base[3], tot = 0;
  for (int j = 0; j < 3; ++j) {
    steps[j] = ((3 + 2 * j) * ld_in + 15) / 16;
    base[j] = tot;
    tot += steps[j] * nblk;
  }
  const int total = tot * 512;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int e = i & 7, lane = (i >> 3) & 63, f = i >> 9;             // f = fragment index
    int j = 2;
    if (f < base[1]) j = 0; else if (f < base[2]) j = 1;
    const int fl = f - base[j], blk = fl / steps[j], s = fl - blk * steps[j];
    const int k = 16 * s + 8 * (lane >> 5) + e;
    const int tap = k / ld_in, ci = k - tap * ld_in, kj = 3 + 2 * j;
    const float* w = j == 0 ? w3 : (j == 1 ? w5 : w7);
    float v = 0.f;
    if (tap < kj && ci < Cin) v = w[((int64_t)(32 * blk + (lane & 31)) * Cin + ci) * kj + tap];
    wpack[i] = (__bf16)v;
  }
}

}  // namespace tecm_convseq

extern "C" int tecm_conv_fwd_pack(const float* w3, const float* w5, const float* w7, void* wpack, int32_t Cout, int32_t Cin,
                                  int32_t ld_in, void* stream) {
  using namespace tecm_convseq;
  TECM_REQUIRE(w3 && w5 && w7 && wpack, TECM_E_ARG, "tecm_conv_fwd_pack: null pointer");
  TECM_REQUIRE(Cout > 0 && Cout % 32 == 0 && Cout <= 128 && Cin > 0 && Cin <= ld_in && ld_in % 8 == 0, TECM_E_ARG,
               "tecm_conv_fwd_pack: Cout a multiple of 32 up to 128, Cin <= ld_in, ld_in a multiple of 8");
  int tot = 0;
  for (int j = 0; j < 3; ++j) tot += (((3 + 2 * j) * ld_in + 15) / 16) * (Cout / 32);
  const int total = tot * 512;
  hipLaunchKernelGGL(conv_fwd_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w3, w5, w7,
                     static_cast<__bf16*>(wpack), Cout, Cin, ld_in);
  TECM_CHECK_LAUNCH("tecm_conv_fwd_pack");
  return TECM_OK;
}

// LDS bytes of one forward tile: (TC + 7 halo steps) x NB nodes x an odd number of 16-byte slots per row
static size_t conv_fwd_lds(int Lc, int ld_in, bool f32, int* pitch_out, int* tc_out) {
  using namespace tecm_convseq;
  int slots = ld_in * (f32 ? 4 : 2) / 16;                               // row pitch in 16-byte slots, odd: conflict-free ds_read_b128
  if (slots % 2 == 0) ++slots;
  const int pitch = slots * 16;
  const int TC = 8 * MAXT < Lc ? 8 * MAXT : Lc;
  if (pitch_out) *pitch_out = pitch;
  if (tc_out) *tc_out = TC;
  return (size_t)(TC + 7) * NB * pitch;
}

// 1 when tecm_conv_fwd_{bf16,f32} serves the shape (callers fall back to the window-view GEMMs otherwise), else 0
extern "C" int tecm_conv_fwd_supported(int32_t Lc, int32_t Cout, int32_t ld_in, int32_t f32) {
  if (Lc <= 0 || Lc % 8 != 0 || Cout <= 0 || Cout % 32 != 0 || Cout > 128 || ld_in <= 0 || ld_in % 8 != 0 || ld_in > 128) return 0;
  return conv_fwd_lds(Lc, ld_in, f32 != 0, nullptr, nullptr) <= 64 * 1024 ? 1 : 0;
}

static int conv_fwd_launch(const TecmConvFwd* p, void* stream, bool f32, const char* who) {
  using namespace tecm_convseq;
  TECM_REQUIRE(p && p->inp && p->wpack && p->bias && p->y, TECM_E_ARG, "%s: null pointer", who);
  TECM_REQUIRE(p->B > 0 && p->Lc > 0 && p->N > 0 && p->Lc % 8 == 0, TECM_E_ARG,
               "%s: bad shape (the sequence length must be a multiple of 8)", who);
  TECM_REQUIRE(p->Cout % 32 == 0 && p->Cout > 0 && p->Cout <= 128 && p->ld_in % 8 == 0 && p->ld_in > 0 && p->ld_in <= 128,
               TECM_E_ARG, "%s: Cout a multiple of 32 up to 128, ld_in a multiple of 8 up to 128", who);
  TECM_REQUIRE(tecm_aligned(p->inp, 16) && tecm_aligned(p->wpack, 16), TECM_E_ALIGN, "%s: 16-byte aligned pointers", who);
  FArgs a;
  a.inp = p->inp;
  a.wpack = p->wpack;
  a.bias = p->bias;
  a.y = p->y;
  a.y16 = p->y_bf16 != 0;
  a.stats = p->stats;
  a.eps = p->eps;
  TECM_REQUIRE(!(a.y16 && f32), TECM_E_ARG, "%s: a bf16 y is written by the bf16 kernel only", who);
  TECM_REQUIRE(!a.y16 || tecm_aligned(p->y, 4), TECM_E_ALIGN, "%s: y must be 4-byte aligned", who);
  a.B = p->B; a.Lc = p->Lc; a.N = p->N; a.Cout = p->Cout; a.ld_in = p->ld_in;
  const size_t lds = conv_fwd_lds(p->Lc, p->ld_in, f32, &a.pitch, &a.TC);
  a.nchunk = (p->Lc + a.TC - 1) / a.TC;
  a.nblk = (p->N + NB - 1) / NB;
  TECM_REQUIRE(!a.stats || (a.y16 && !f32 && a.nchunk == 1 && p->eps > 0.f), TECM_E_ARG,
               "%s: GroupNorm statistics come with a bf16 y and sequences of at most %d steps (one tile per sequence)", who, a.TC);
  TECM_REQUIRE(lds <= 64 * 1024, TECM_E_LDS, "%s: %zu B of LDS per tile", who, lds);
  // units (kernel size j, 32-channel block), longest first, each to the least loaded wave
  const int nb32 = p->Cout / 32;
  a.nb32 = nb32;
  const int nunits = 3 * nb32;
  int load[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
  unsigned long long assign = ~0ull;
  for (int u = nunits - 1; u >= 0; --u) {                  // kernel size 7 first
    const int j = u / nb32, st = f32 ? (3 + 2 * j) * p->ld_in / 8 : ((3 + 2 * j) * p->ld_in + 15) / 16;
    int best = -1;
    for (int w = 0; w < 4; ++w)
      if (cnt[w] < 3 && (best < 0 || load[w] < load[best])) best = w;
    TECM_REQUIRE(best >= 0, TECM_E_ARG, "%s: more than 12 (kernel size, channel block) units", who);
    const int sh = 12 * best + 4 * cnt[best]++;
    assign = (assign & ~(15ull << sh)) | ((unsigned long long)u << sh);
    load[best] += st;
  }
  a.assign = assign;
  const int64_t tiles = (int64_t)p->B * a.nchunk * a.nblk;
  TECM_REQUIRE(tiles < ((int64_t)1 << 31), TECM_E_ARG, "%s: too many tiles", who);
  if (f32) hipLaunchKernelGGL(conv_fwd_seq_f32_kernel, dim3((unsigned)tiles), dim3(NTH), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(conv_fwd_seq_kernel, dim3((unsigned)tiles), dim3(NTH), lds, (hipStream_t)stream, a);
  TECM_CHECK_LAUNCH("tecm_conv_fwd");
  return TECM_OK;
}

extern "C" int tecm_conv_fwd_bf16(const TecmConvFwd* p, void* stream) { return conv_fwd_launch(p, stream, false, "tecm_conv_fwd_bf16"); }
extern "C" int tecm_conv_fwd_f32(const TecmConvFwd* p, void* stream) { return conv_fwd_launch(p, stream, true, "tecm_conv_fwd_f32"); }

extern "C" int tecm_conv_fwd_pack_f32(const float* w3, const float* w5, const float* w7, float* wpack, int32_t Cout,
                                      int32_t Cin, int32_t ld_in, void* stream) {
  using namespace tecm_convseq;
  TECM_REQUIRE(w3 && w5 && w7 && wpack, TECM_E_ARG, "tecm_conv_fwd_pack_f32: null pointer");
  TECM_REQUIRE(Cout > 0 && Cout % 32 == 0 && Cout <= 128 && Cin > 0 && Cin <= ld_in && ld_in % 8 == 0, TECM_E_ARG,
               "tecm_conv_fwd_pack_f32: Cout a multiple of 32 up to 128, Cin <= ld_in, ld_in a multiple of 8");
  int tot = 0;
  for (int j = 0; j < 3; ++j) tot += ((3 + 2 * j) * ld_in / 8) * (Cout / 32);
  const int total = tot * 256;
  hipLaunchKernelGGL(conv_fwd_pack_f32_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w3, w5, w7, wpack,
                     Cout, Cin, ld_in);
  TECM_CHECK_LAUNCH("tecm_conv_fwd_pack_f32");
  return TECM_OK;
}

extern "C" int tecm_conv_dx_pack(const float* w3, const float* w5, const float* w7, void* wpack, int32_t Cout, int32_t Cin,
                                 int32_t ld_in, void* stream) {
  using namespace tecm_convseq;
  TECM_REQUIRE(w3 && w5 && w7 && wpack, TECM_E_ARG, "tecm_conv_dx_pack: null pointer");
  TECM_REQUIRE(Cout > 0 && Cout % 64 == 0 && Cin > 0 && Cin <= ld_in && ld_in % 4 == 0, TECM_E_ARG,
               "tecm_conv_dx_pack: Cout must be a multiple of 64, Cin <= ld_in, ld_in a multiple of 4");
  const int NCI = (ld_in + 31) / 32;
  const int total = 15 * Cout * NCI * 32;
  hipLaunchKernelGGL(conv_dx_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w3, w5, w7,
                     static_cast<__bf16*>(wpack), Cout, Cin, NCI);
  TECM_CHECK_LAUNCH("tecm_conv_dx_pack");
  return TECM_OK;
}

// Tile plan of the d-input kernel: time steps per tile (at most MAXT accumulator tiles per wave), the offset of the shared
// zero row behind the staged image / the exchange buffer, and the LDS bytes of one tile.
struct ConvDxPlan { int TC, zero_off, row_bytes, pitch, NCI; size_t lds; };
static ConvDxPlan conv_dx_plan(int Lc, int Cout, int ld_in, bool f32) {
  using namespace tecm_convseq;
  ConvDxPlan q;
  q.NCI = (ld_in + 31) / 32;
  q.row_bytes = 3 * Cout * (f32 ? 4 : 2);
  q.pitch = q.row_bytes + 16;
  int TC = 8 * (MAXT / q.NCI);
  if (TC > Lc) TC = Lc;
  auto img_steps = [&](int tc) { return tc + 6 < Lc ? tc + 6 : Lc; };
  auto zero_of = [&](int tc) {                             // the shared zero row sits behind the image AND the exchange buffer
    const size_t img = (size_t)img_steps(tc) * NB * q.pitch;
    const size_t xch = (size_t)(tc / 8) * q.NCI * (f32 ? 4 : 3) * 16 * 64 * 4;   // partial sums of 3 (fp32 round 1: 4) waves
    return img > xch ? img : xch;
  };
  while (TC > 8 && zero_of(TC) + q.pitch > 160 * 1024) TC -= 8;
  q.TC = TC;
  q.zero_off = (int)zero_of(TC);
  q.lds = zero_of(TC) + q.pitch;
  return q;
}

// 1 when tecm_conv_dx_{bf16,f32} serves the shape (callers fall back to the window-view GEMMs otherwise), else 0
extern "C" int tecm_conv_dx_supported(int32_t Lc, int32_t Cout, int32_t ld_in, int32_t f32) {
  if (Lc <= 0 || Lc % 8 != 0 || Cout <= 0 || Cout % 64 != 0 || ld_in <= 0 || ld_in % 4 != 0 || ld_in > 64) return 0;
  return conv_dx_plan(Lc, Cout, ld_in, f32 != 0).lds <= 160 * 1024 ? 1 : 0;
}

static int conv_dx_launch(const TecmConvDx* p, void* stream, bool f32, const char* who) {
  using namespace tecm_convseq;
  TECM_REQUIRE(p && p->dy && p->wpack && p->dinp, TECM_E_ARG, "%s: null pointer", who);
  TECM_REQUIRE(p->B > 0 && p->Lc > 0 && p->N > 0, TECM_E_ARG, "%s: bad shape", who);
  TECM_REQUIRE(p->Cout % 64 == 0 && p->Cout > 0 && p->ld_in % 4 == 0 && p->ld_in > 0 && p->ld_in <= 64, TECM_E_ARG,
               "%s: Cout must be a multiple of 64 and ld_in a multiple of 4 up to 64", who);
  TECM_REQUIRE(p->Lc % 8 == 0, TECM_E_ARG, "%s: the sequence length must be a multiple of 8", who);
  TECM_REQUIRE(tecm_aligned(p->dy, 16) && tecm_aligned(p->wpack, 16) && tecm_aligned(p->dinp, 16), TECM_E_ALIGN,
               "%s: 16-byte aligned pointers", who);
  Args a;
  a.dy = p->dy;
  a.wpack = p->wpack;
  a.dinp = p->dinp;
  a.B = p->B; a.Lc = p->Lc; a.N = p->N; a.Cout = p->Cout; a.ld_in = p->ld_in;
  // time steps per tile: at most MAXT accumulator tiles per wave, and the staged rows (+ 3 halo steps each side) must
  // fit: two blocks per CU when that costs nothing (Lc <= 48 at Cout = 64, Lc <= 24 at Cout = 128), else one
  const ConvDxPlan q = conv_dx_plan(p->Lc, p->Cout, p->ld_in, f32);
  a.NCI = q.NCI;
  a.row_bytes = q.row_bytes;
  a.cpr = a.row_bytes / 16;
  a.pitch = q.pitch;
  const size_t lds = q.lds;
  TECM_REQUIRE(lds <= 160 * 1024, TECM_E_LDS, "%s: %d channels need %zu B of LDS per tile", who, 3 * p->Cout, lds);
  const int TC = q.TC;
  a.TC = TC;
  a.zero_off = q.zero_off;
  a.nchunk = (p->Lc + TC - 1) / TC;
  a.nblk = (p->N + NB - 1) / NB;
  const int64_t tiles = (int64_t)p->B * a.nchunk * a.nblk;
  TECM_REQUIRE(tiles < ((int64_t)1 << 31), TECM_E_ARG, "%s: too many tiles", who);
  // more than 64 KiB of dynamic LDS needs the attribute on the function of THIS device: set per launch (cheap, and
  // correct for a process that drives several GPUs or calls from several threads), return code checked
  const void* fn = f32 ? (a.NCI == 1 ? reinterpret_cast<const void*>(&conv_dx_seq_f32_kernel<1>)
                                     : reinterpret_cast<const void*>(&conv_dx_seq_f32_kernel<2>))
                       : (a.NCI == 1 ? reinterpret_cast<const void*>(&conv_dx_seq_kernel<1>)
                                     : reinterpret_cast<const void*>(&conv_dx_seq_kernel<2>));
  TECM_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, TECM_E_LAUNCH,
               "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", who);
  if (f32) {
    if (a.NCI == 1)
      hipLaunchKernelGGL(conv_dx_seq_f32_kernel<1>, dim3((unsigned)tiles), dim3(NTH32), lds, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL(conv_dx_seq_f32_kernel<2>, dim3((unsigned)tiles), dim3(NTH32), lds, (hipStream_t)stream, a);
  } else if (a.NCI == 1) {
    hipLaunchKernelGGL(conv_dx_seq_kernel<1>, dim3((unsigned)tiles), dim3(NTH), lds, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(conv_dx_seq_kernel<2>, dim3((unsigned)tiles), dim3(NTH), lds, (hipStream_t)stream, a);
  }
  TECM_CHECK_LAUNCH("tecm_conv_dx");
  return TECM_OK;
}

extern "C" int tecm_conv_dx_bf16(const TecmConvDx* p, void* stream) { return conv_dx_launch(p, stream, false, "tecm_conv_dx_bf16"); }
extern "C" int tecm_conv_dx_f32(const TecmConvDx* p, void* stream) { return conv_dx_launch(p, stream, true, "tecm_conv_dx_f32"); }

extern "C" int tecm_conv_dx_pack_f32(const float* w3, const float* w5, const float* w7, float* wpack, int32_t Cout,
                                     int32_t Cin, int32_t ld_in, void* stream) {
  using namespace tecm_convseq;
  TECM_REQUIRE(w3 && w5 && w7 && wpack, TECM_E_ARG, "tecm_conv_dx_pack_f32: null pointer");
  TECM_REQUIRE(Cout > 0 && Cout % 64 == 0 && Cin > 0 && Cin <= ld_in && ld_in % 4 == 0, TECM_E_ARG,
               "tecm_conv_dx_pack_f32: Cout must be a multiple of 64, Cin <= ld_in, ld_in a multiple of 4");
  const int NCI = (ld_in + 31) / 32;
  const int total = 15 * Cout * NCI * 32;
  hipLaunchKernelGGL(conv_dx_pack_f32_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w3, w5, w7, wpack,
                     Cout, Cin, NCI);
  TECM_CHECK_LAUNCH("tecm_conv_dx_pack_f32");
  return TECM_OK;
}
