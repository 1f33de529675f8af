// Multi_Scale_Conv_Block (reference modules.py:43-60), bf16 mode: the WEIGHT gradient of the three parallel Conv1d
// (k = 3, 5, 7) as ONE kernel that reads the block input and dy once.
//
//     dw_j[co, ci, tau] = sum_{b, t, n}  dy[b, t, n, j*Cout + co] * inp[b, t + tau - p_j, n, ci],     p_j = (k_j - 1) / 2
//
// Before: three split-K window GEMMs (A = dy^T, B = the window view of inp, K = B*Lc*N up to 1.1 M rows), each staging
// its operands through a transposing register stager and re-reading dy / the taps of inp: 0.54 + 0.60 ms per step at
// 60-250 TFLOP/s for 2 x 0.5 GB of operands -- five times what HBM needs to deliver them once.  Here: a SEQUENCE tile.
//   * a persistent block (one per CU, 4 waves = one per SIMD, 512 registers each) walks tiles of 4 nodes x the whole
//     sequence (Lc <= 48 time steps) of one
//     sample; the tile's rows of inp and dy are staged in LDS in their NATURAL layout (row = (t, node), channels
//     contiguous), inp with 3 zero time steps in front and behind, so every tap of every kernel size is a ROW offset of
//     the same image and the zero padding of the convolution is the halo;
//   * the contraction runs over rows, which is the k index of the MFMA: both operands are wanted "k-major".
//     ds_read_b64_tr_b16 (gfx950) delivers exactly that from the natural image -- lane 4q+p of a 16-lane group supplies
//     row q (= node q of one time step), lane i receives column i of the 4 rows -- so nothing is transposed in
//     registers and each lane's row address is linear in the k-chunk index;
//   * v_mfma_f32_32x32x16_bf16 with A = 32 input channels x 16 rows, B = 16 rows x 32 output channels.  A wave owns one
//     (input-channel block, output-channel block) pair and ALL 15 taps of it: 15 accumulator tiles = 240 registers
//     that live across the whole kernel.  Per 16-row k-chunk it issues 8 transposed reads of inp (time steps t-3..t+4:
//     the upper half of tap o's fragment is the lower half of tap o+1's) + 6 of dy for 15 MFMAs;
//   * the 15 x Cin x Cout accumulators of Cout = 128 / ld_in = 64 are 480 KB -- the whole register file of a CU holds
//     512 KB -- so a block owns only HALF the output channels ("flavor" = blockIdx % 2) and stages only those columns
//     of dy; inp (1/6 of the bytes) is read by both flavors.  Cout = 64 / ld_in = 24 has only two (cib, cob) pairs:
//     the 4 waves then also split the k-chunks two ways;
//   * the NEXT tile travels from HBM into registers (12-21 x 16 B per lane) while the current one is multiplied;
//   * every wave writes its 15 tiles once, at the end, to a per-block slab; conv_dw_reduce_kernel sums the slabs in a
//     fixed order and writes the three (Cout, Cin, k) gradients.  Bit-reproducible.
// Arithmetic = the bf16 mode's: operands are the bf16 tensors the GEMM path reads (inp16 written by the producer, dy by
// the GroupNorm backward), fp32 accumulation; only the summation order differs.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace tecm_convdw {

constexpr int NTH = 256;     // 4 waves: one per SIMD, the 240 accumulator registers of a wave need a 512-register budget
constexpr int NB = 4;        // nodes per tile = rows of one transposed-read block
constexpr int HALO = 3;      // zero time steps in front of / behind the inp image (k = 7)
constexpr int NTAP = 15;     // 3 + 5 + 7

struct Args {
  const void* x;             // bf16 (B, Lc, N, ld_in)
  const void* dy;            // bf16 (B, Lc, N, 3 * Cout)
  float* ws;                 // slabs [flavor][(gridDim.x / F) * KS][15][32 * NCIB][32 * CPB]
  int B, Lc, N, ntiles, nblk;
};

template <int LD_IN, int COUT>
struct Geo {
  static constexpr int NCIB = (LD_IN + 31) / 32;             // input-channel blocks of 32
  static constexpr int NCOB = COUT / 32;
  static constexpr int CPB = NCOB < 4 / NCIB ? NCOB : 4 / NCIB;   // output-channel blocks per thread block
  static constexpr int F = NCOB / CPB;                       // flavors: thread blocks that share a tile's inp, split dy's columns
  static constexpr int R = NCIB * CPB;                       // (cib, cob) pairs = wave roles
  static constexpr int KS = 4 / R;                           // waves per role: they split the k-chunks
  static constexpr int XB = LD_IN * 2, YB = 3 * COUT * 2;    // bytes of one row in HBM
  static constexpr int YI = 3 * CPB * 64;                    // bytes of one dy row in the image: 3 branches x CPB*32 columns
  // pitches are odd multiples of 64 B: the 4 rows x 64 B of a transposed read (one 32-lane half) tile all 64 banks
  static constexpr int XP = NCIB == 1 ? 64 : 192;
  static constexpr int YP = YI + 64;
  static constexpr int CPRX = XB / 16, CPRY = YI / 16;       // 16-byte chunks per row
  static constexpr int CPS = CPB * 4;                        // chunks per branch segment
  static constexpr int MAXROWS = 48 * NB;
  static constexpr int SBX = (MAXROWS * CPRX + NTH - 1) / NTH, SBY = (MAXROWS * CPRY + NTH - 1) / NTH;
  static constexpr int SLAB = NTAP * 32 * NCIB * 32 * CPB;   // floats
  static_assert(R * KS == 4 && CPB * F == NCOB, "roles must tile the 4 waves");
  static_assert(LD_IN % 8 == 0 && COUT % 32 == 0 && XB <= (NCIB == 1 ? 64 : 128), "shape");
  static_assert((YP / 64) % 2 == 1, "dy pitch must be an odd multiple of 64 B");
};

typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
__device__ __forceinline__ bf16x4 tr_read(lds_ptr p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(p));
}
__device__ __forceinline__ bf16x8 join(const bf16x4& lo, const bf16x4& hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int LD_IN, int COUT>
__global__ __launch_bounds__(NTH) void conv_dw_seq_kernel(const Args a) {
  using G = Geo<LD_IN, COUT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = a.Lc * NB;                                // rows of the dy image; multiple of 16
  const int xbytes = (a.Lc + 2 * HALO) * NB * G::XP, ybytes = rows * G::YP;
  unsigned char* xs = lds;
  unsigned char* ys = lds + xbytes;
  {                                                          // halo rows (and every pad byte) are zero for the whole kernel
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid * 16; i < xbytes + ybytes; i += NTH * 16) *reinterpret_cast<u32x4*>(lds + i) = z;
  }
  const int role = wave % G::R, kq = wave / G::R;
  const int cib = role % G::NCIB, cob = role / G::NCIB;      // cob: local to this block's flavor
  const int flavor = blockIdx.x % G::F, bif = blockIdx.x / G::F, nbf = gridDim.x / G::F;
  f32x16 acc[NTAP];
#pragma unroll
  for (int i = 0; i < NTAP; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int nx = rows * G::CPRX, ny = rows * G::CPRY;
  const char* xg = static_cast<const char*>(a.x);
  const char* yg = static_cast<const char*>(a.dy);
  u32x4 vx[G::SBX], vy[G::SBY];
  // chunk idx of the x (y) image: row = idx / CPR = (t, node), 16-byte chunk idx % CPR.  Loads are clamped (always in
  // bounds), only real chunks are stored, the rows of nodes >= N are stored as zeros.
  auto stage_load = [&](int tile) {
    const int b = tile / a.nblk, n0 = (tile - b * a.nblk) * NB;
    const int64_t row0 = (int64_t)b * a.Lc * a.N;
#pragma unroll
    for (int q = 0; q < G::SBX; ++q) {
      const int idx = min(tid + q * NTH, nx - 1);
      const int row = idx / G::CPRX, ch = idx - row * G::CPRX;
      const int ng = min(n0 + (row & 3), a.N - 1);
      vx[q] = *reinterpret_cast<const u32x4*>(xg + (row0 + (int64_t)(row >> 2) * a.N + ng) * G::XB + ch * 16);
    }
#pragma unroll
    for (int q = 0; q < G::SBY; ++q) {
      const int idx = min(tid + q * NTH, ny - 1);
      const int row = idx / G::CPRY, ch = idx - row * G::CPRY;
      const int ng = min(n0 + (row & 3), a.N - 1);
      const int seg = ch / G::CPS, c = ch - seg * G::CPS;    // branch, chunk inside this flavor's columns of it
      vy[q] = *reinterpret_cast<const u32x4*>(yg + (row0 + (int64_t)(row >> 2) * a.N + ng) * G::YB + seg * (COUT * 2) +
                                              flavor * (G::CPB * 64) + c * 16);
    }
  };
  auto stage_store = [&](int tile) {
    const int b = tile / a.nblk, n0 = (tile - b * a.nblk) * NB;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < G::SBX; ++q) {
      const int idx = tid + q * NTH;
      const int row = idx / G::CPRX, ch = idx - row * G::CPRX;
      if (idx < nx)
        *reinterpret_cast<u32x4*>(xs + (row + HALO * NB) * G::XP + ch * 16) = (n0 + (row & 3) < a.N) ? vx[q] : z;
    }
#pragma unroll
    for (int q = 0; q < G::SBY; ++q) {
      const int idx = tid + q * NTH;
      const int row = idx / G::CPRY, ch = idx - row * G::CPRY;
      if (idx < ny) *reinterpret_cast<u32x4*>(ys + row * G::YP + ch * 16) = (n0 + (row & 3) < a.N) ? vy[q] : z;
    }
  };

  // this lane's addresses for the transposed reads of k-chunk 0: it supplies row q (node q) of time step 2h (+1 for the
  // second read of a fragment), columns 16*g1 + 4p .. +3 of the wave's channel block
  const int q = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1, h = lane >> 5;
  const lds_ptr base3 = (lds_ptr)lds;
  const int x0 = (2 * h * NB + q) * G::XP + (cib * 32 + 16 * g1 + 4 * p) * 2;            // tap offset -3 = image row t
  const int y0 = xbytes + (2 * h * NB + q) * G::YP + (cob * 32 + 16 * g1 + 4 * p) * 2;
  const int nchunks = rows / 16;

  int tile = bif;
  if (tile < a.ntiles) stage_load(tile);
  __syncthreads();                                           // the zero fill is complete
  for (; tile < a.ntiles; tile += nbf) {
    stage_store(tile);
    __syncthreads();
    if (tile + nbf < a.ntiles) stage_load(tile + nbf);
    for (int kc = kq; kc < nchunks; kc += G::KS) {
      const lds_ptr xp = base3 + x0 + kc * 16 * G::XP;
      const lds_ptr yp = base3 + y0 + kc * 16 * G::YP;
      bf16x4 xr[8];                                          // time steps t-3 .. t+4 (t = 4*kc + 2h)
#pragma unroll
      for (int o = 0; o < 8; ++o) xr[o] = tr_read(xp + o * NB * G::XP);
      bf16x8 yf[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) yf[j] = join(tr_read(yp + j * (G::CPB * 64)), tr_read(yp + NB * G::YP + j * (G::CPB * 64)));
      // tap o (time offset o - 3): kernel size 7 uses o = 0..6, 5 uses 1..5, 3 uses 2..4
#pragma unroll
      for (int o = 0; o < 7; ++o) {
        const bf16x8 xf = join(xr[o], xr[o + 1]);
        acc[8 + o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf[2], acc[8 + o], 0, 0, 0);
        if (o >= 1 && o <= 5) acc[3 + o - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf[1], acc[3 + o - 1], 0, 0, 0);
        if (o >= 2 && o <= 4) acc[o - 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf[0], acc[o - 2], 0, 0, 0);
      }
    }
    __syncthreads();                                         // every wave is done with the image
  }
  // slab [flavor][(block in flavor, kq)][tap][ci][local co]: lane holds co = lane & 31 of rows ci = 8*(i>>2) + 4*(lane>>5) + (i&3)
  float* slab = a.ws + (((int64_t)flavor * nbf + bif) * G::KS + kq) * G::SLAB;
#pragma unroll
  for (int tp = 0; tp < NTAP; ++tp)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ci = cib * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
      slab[(tp * 32 * G::NCIB + ci) * (32 * G::CPB) + cob * 32 + (lane & 31)] = acc[tp][i];
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

template <int LD_IN, int COUT>
int launch(const TecmConvDw* p, hipStream_t st) {
  using G = Geo<LD_IN, COUT>;
  Args a;
  a.x = p->inp; a.dy = p->dy; a.ws = p->workspace;
  a.B = p->B; a.Lc = p->Lc; a.N = p->N;
  a.nblk = (p->N + NB - 1) / NB;
  const int64_t tiles = (int64_t)p->B * a.nblk;
  TECM_REQUIRE(tiles < ((int64_t)1 << 31), TECM_E_ARG, "tecm_conv_dw_bf16: too many tiles");
  a.ntiles = (int)tiles;
  const size_t lds = (size_t)(p->Lc + 2 * HALO) * NB * G::XP + (size_t)p->Lc * NB * G::YP;
  TECM_REQUIRE(lds <= 160 * 1024, TECM_E_LDS, "tecm_conv_dw_bf16: %zu B of LDS per tile", lds);
  int nbf = p->num_blocks / G::F;                           // blocks per flavor
  if (nbf > tiles) nbf = (int)tiles;
  TECM_REQUIRE(nbf >= 1, TECM_E_ARG, "tecm_conv_dw_bf16: num_blocks must be at least %d", G::F);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dw_seq_kernel<LD_IN, COUT>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_dw_seq_kernel<LD_IN, COUT>), dim3(nbf * G::F), dim3(NTH), lds, st, a);
  TECM_CHECK_LAUNCH("tecm_conv_dw_bf16/seq");
  hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3((unsigned)(NTAP * p->Cin * (COUT / 64))), dim3(256), 0, st, p->workspace,
                     nbf * G::KS, 32 * G::NCIB, 32 * G::CPB, COUT, p->Cin, p->dw3, p->dw5, p->dw7);
  TECM_CHECK_LAUNCH("tecm_conv_dw_bf16/reduce");
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

extern "C" int tecm_conv_dw_bf16(const TecmConvDw* p, void* stream) {
  TECM_REQUIRE(p && p->inp && p->dy && p->workspace && p->dw3 && p->dw5 && p->dw7, TECM_E_ARG, "tecm_conv_dw_bf16: null pointer");
  TECM_REQUIRE(p->B > 0 && p->N > 0 && p->Lc > 0 && p->Lc % 4 == 0 && p->Lc <= 48, TECM_E_ARG,
               "tecm_conv_dw_bf16: the sequence length must be a multiple of 4 up to 48 (got %d)", p->Lc);
  TECM_REQUIRE(p->Cin > 0 && p->Cin <= p->ld_in && p->num_blocks > 0, TECM_E_ARG, "tecm_conv_dw_bf16: bad Cin / num_blocks");
  TECM_REQUIRE(tecm_aligned(p->inp, 16) && tecm_aligned(p->dy, 16), TECM_E_ALIGN, "tecm_conv_dw_bf16: 16-byte aligned tensors");
  hipStream_t st = (hipStream_t)stream;
  if (p->ld_in == 64 && p->Cout == 128) return tecm_convdw::launch<64, 128>(p, st);
  if (p->ld_in == 24 && p->Cout == 64) return tecm_convdw::launch<24, 64>(p, st);
  if (p->ld_in == 64 && p->Cout == 64) return tecm_convdw::launch<64, 64>(p, st);
  if (p->ld_in == 24 && p->Cout == 128) return tecm_convdw::launch<24, 128>(p, st);
  TECM_REQUIRE(false, TECM_E_ARG, "tecm_conv_dw_bf16: built for ld_in in {24, 64} x Cout in {64, 128} (got %d, %d)", p->ld_in,
               p->Cout);
  return TECM_E_ARG;
}
