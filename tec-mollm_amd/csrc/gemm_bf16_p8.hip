// bf16 GEMM, eighth geometry (round 5): the eight-phase K loop of gemm_bf16_p8_loop.h (256 x 256 x 64 tile, half-tile
// LDS-DMA ring kept in flight across raw barriers, two wave groups half a phase apart) behind the straight-line
// epilogue of the ring kernels (gemm_bf16_dma.hip, fifth geometry: same wave tile 128 x 64 = 8 x 4 tiles of 16 x 16,
// same C/D map, so the parking of the accumulators and epi_fast_dispatch are used as they are).  Serves the plain
// MK x NK contractions with both operands bf16 tensors: 7 of the 8 big GPT-2 GEMMs of a layer in bf16 mode (BASELINE
// configs[2]; reference call sites modules.py:205-209, arithmetic train.py:68).
//
// Measured (round 5, M = 69 864, random operands; fifth geometry -> this): K loop alone N=3072 K=768 362 -> 252-257 us,
// N=768 K=3072 398 -> 284 us; 8192^3 with a bare store epilogue 1.11 -> 1.31 PFLOP/s (K loop alone 1.53).
// The summation order is the fifth geometry's per MFMA but K is walked 64 at a time in quadrant order: results agree
// with it to fp32 rounding (test_bf16_p8_geometry_agrees_to_fp32_rounding), and launches are bit-reproducible.
#include <cstdlib>
#ifdef P8_STAMPS
#include <hip/hip_runtime.h>
__shared__ unsigned long long p8_epi_ts[16];
#define TECM_EPI_STAMP(i) do { if (threadIdx.x == 0) p8_epi_ts[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#include "gemm_bf16_impl.h"
#include "gemm_bf16_p8_loop.h"

#ifdef P8_STAMPS                                         // diagnostics (tools/build_variant.py): per-block time stamps, 10 ns ticks
__device__ unsigned long long p8_stamps[8 * 8192];       // [block][t_entry, t_kloop_done, t_stores_issued, t_stores_acked, hw_id, xcc_id, -, -]
__device__ unsigned long long p8_epi_stamps[16 * 8192];  // [block][slab][before park, parked, rows done]
extern "C" int tecm_p8_stamps_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(p8_stamps), sizeof(unsigned long long) * n);
}
extern "C" int tecm_p8_epi_stamps_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(p8_epi_stamps), sizeof(unsigned long long) * n);
}
#define P8_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) p8_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define P8_STAMP(i) do { } while (0)
#endif

#ifndef P8_SPECIAL                                       // false: every epilogue through the run-time feature tests (A/B)
#define P8_SPECIAL true
#endif

namespace tecm_gemm16 {

// WR: rows per wave row -- 128 (the 256 x 256 tile) or 112 / 96 (short tiles of 224 / 192 rows, gemm_bf16_p8_loop.h): the
// launcher picks the height whose tile count wastes least of the last round of CUs.
template <int WR>
__global__ __launch_bounds__(tecm_p8::NTH, 1) void gemm_bf16_p8_kernel(const TecmGemm g, int tiles_m, int tiles_n) {
  // (names qualified on purpose: tecm_gemm16 has its own BM / BN / BK, which would hide tecm_p8's behind a using-directive)
  constexpr int BM = 2 * WR, BN = tecm_p8::BN, MT = tecm_p8::MT, NT = tecm_p8::NT, LDS_BYTES = tecm_p8::LDS_BYTES;
  constexpr int WTM = WR, WTN = 64;
  constexpr int SLABS = (WR + 31) / 32;                  // 32-row slabs of a wave's rows; the last one is half used at WR = 112
  static_assert(LDS_BYTES >= 8 * 32 * (WTN + 4) * 4 && tecm_p8::BMAP == 1, "epilogue slabs fit in the operand ring");
  static_assert(tecm_p8::BM == 256 && BN == 256 && MT * 16 == 128 && NT * 16 == WTN, "wave tile <= 128 x 64, block tile <= 256 x 256");
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[LDS_BYTES];

  // block -> tile map: XCD-contiguous runs, GROUP_M m-tiles per L2 super-tile (as the other geometries)
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

  P8_STAMP(0);
#ifdef P8_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    p8_stamps[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
    p8_stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
  }
#endif
  tecm_p8::f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  const tecm_p8::Operands o{reinterpret_cast<const __bf16*>(g.A), reinterpret_cast<const __bf16*>(g.B), g.lda, g.ldb, g.M, g.N, (int)g.K,
                            g.a_win.enabled ? g.a_win.N : 0, g.a_win.Lin, g.a_win.Lout, g.a_win.stride_t, g.a_win.Cw};
  tecm_p8::kloop<WR>(o, m0, n0, smem_raw, acc);
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
  P8_STAMP(1);
  __syncthreads();                                      // every wave has left the K loop: the ring becomes staging
  // ---- epilogue: gemm_impl.h's straight-line form over this wave's private staging rows.  Accumulator map (BMAP 1,
  // gemm_bf16_p8_loop.h): lane (fr, fq) holds 16 consecutive columns 16 fq .. of row 16 i + fr, so a slab of 32 rows is
  // parked with 8 ds_write_b128 per lane (conflict-free: 8 consecutive lanes = 8 rows at a pitch of 68 floats).
  // (Round 5, measured and not kept: storing row-wise straight from the registers, no LDS -- 16-byte pieces at a 32- or
  //  64-byte stride per instruction: fp32 C 392 -> 431..1367 us, bf16 C 356 -> 344..503 us over both column maps and
  //  store policies at N = 3072 / K = 768; the coalescer wants whole contiguous row segments per instruction.)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int STG_LD = WTN + 4;
  const DropCtx odc = make_drop(g.out_drop);
  float* stg = reinterpret_cast<float*>(smem_raw) + wave * (32 * STG_LD);
  auto stage_slab = [&](auto ic) {                     // 32 accumulator rows = row tiles 2 i, 2 i + 1
    constexpr int i = decltype(ic)::value;
    static_for<2>([&](auto tc) {
      constexpr int ti = decltype(tc)::value;
      static_for<NT>([&](auto jc) {
        constexpr int jn = decltype(jc)::value;
        // (written as the float4 the straight-line reader loads: a store through another vector type is a different
        //  TBAA type, and the compiler may then move the reader's loads across it)
        const tecm_p8::f32x4 v = acc[2 * i + ti][jn];
        *reinterpret_cast<float4*>(&stg[(16 * ti + fr) * STG_LD + 16 * fq + 4 * jn]) = make_float4(v[0], v[1], v[2], v[3]);
      });
    });
  };
  // a short tile's wave stops at its own WR rows: the rows behind them belong to the next wave row / tile (the epilogue
  // masks rows by M only)
  TecmGemm gw = g;
  if constexpr (WR != 128) gw.M = min(g.M, m0 + (int64_t)(wm + 1) * WTM);
  const int fmode = tecm_gemm::epi_fast_mode(g);         // >= 0: checked on the host (tecm_gemm16_dma_try's can16)
  constexpr int LPR = WTN / 4, RPI = 64 / LPR;
  const int lcol = (lane % LPR) * 4, lrow = lane / LPR;
  const EpiCol ecol = epi_col(g, n0 + wn * WTN + lcol);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + (ecol.ok ? ecol.n : 0));   // (lanes outside the matrix: column 0, never stored)
  tecm_gemm::epi_fast_dispatch<SLABS, 32 / RPI, RPI, STG_LD, true, P8_SPECIAL>(fmode, gw, odc, stg, lrow, lcol, m0 + wm * WTM, ecol, bias4,
                                                                     stage_slab);
#ifdef P8_STAMPS
  P8_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P8_STAMP(3);
  if (threadIdx.x == 0 && blockIdx.x < 8192)
    for (int i = 0; i < 12; ++i) p8_epi_stamps[blockIdx.x * 16 + i] = p8_epi_ts[i];
#endif
}

}  // namespace tecm_gemm16

// Rows per wave row (tile height / 2) the launcher uses for an M x N result.  The short tile is chosen where it was
// MEASURED to win (M = 69 864, same box, 256-row -> 224-row tiles): results at most three tiles wide whose tile count it
// brings closer to whole rounds of CUs -- N = 768 / K = 3072 352 -> 332 us, N = 768 / K = 768 131 -> 128, head d-input
// (M = 23 288, N = 2304) 94 -> 92.  The rounds-of-CUs model alone over-promises (it says -12.5 % for all N <= 800 shapes):
// blocks are re-issued as CUs free up and the last, partly filled round runs faster per tile -- the launch is bound by what
// the CUs share, not by the slowest CU -- so wider results (N = 800 / 2304: +2 %; N = 3072: +9 %) and 192-row tiles
// (+0 .. +13 %) stay on 256 rows; with a residual + dropout epilogue the two heights tie.
// TECM_P8_ROWS = 128 | 112 | 96 pins it (A/B diagnostics).  Mirrored by tecmollm/ops.py (kernel names of the roofline).
extern "C" int tecm_p8_rows(int64_t M, int64_t N) {
  if (const char* e = std::getenv("TECM_P8_ROWS")) {     // read per call: the tests pin one height after another
    const int v = atoi(e);
    if (v == 128 || v == 112 || v == 96) return v;
  }
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  const int64_t tn = (N + tecm_p8::BN - 1) / tecm_p8::BN;
  auto cost = [&](int wr) { return ((((M + 2 * wr - 1) / (2 * wr)) * tn + cus - 1) / cus) * wr; };
  if (tn <= 3 && cost(112) < cost(128)) return 112;
  if (M <= 32768 && cost(112) < cost(128)) return 112;      // few m-tiles (the head's d-input): the last round weighs more
  return 128;
}

// 1 when this geometry served the call, 0 when the call is not eligible (the caller falls through to the ring kernels).
// Eligibility beyond the ring kernels' (checked by the caller: both operands bf16, plain views, float4 epilogue, one
// optional input stream): K >= 128 in whole 32-column halves, tiles of 256 rows, both operands addressable with 32-bit
// byte offsets, and no mostly empty last n-tile below N = 768.
int tecm_gemm16_p8_try(const TecmGemm& g, hipStream_t st) {
  using namespace tecm_gemm16;
  const char* sel = std::getenv("TECM_BF16_P8");         // "0": never (A/B diagnostics), "1": also where the default declines
  if (sel && sel[0] == '0') return 0;
  const bool force = sel && sel[0] == '1';
  if (g.K < 128 || g.K % 32 != 0 || g.M < tecm_p8::BM || g.N < 128) return 0;
  int64_t a_rows = g.M;                                  // source rows of A the kernel may address
  if (g.a_win.enabled) {
    // a pad-free window view whose taps are whole K-tiles and stay inside the sequence (the patch projection, modules.py:114)
    const TecmWin& w = g.a_win;
    if (g.b_win.enabled || w.pad != 0 || w.Cw % tecm_p8::BK != 0 || (int64_t)(w.Lout - 1) * w.stride_t + w.taps > w.Lin ||
        g.M % ((int64_t)w.Lout * w.N) != 0)
      return 0;
    a_rows = g.M / ((int64_t)w.Lout * w.N) * w.Lin * w.N;
  }
  if ((a_rows * g.lda + 64) * 2 >= (int64_t(1) << 32) || (g.N * g.ldb + 64) * 2 >= (int64_t(1) << 32)) return 0;
  // a last n-tile that is mostly padding is only worth it from N = 768 on (N = 800, d c_attn: 290 us here against 344 on
  // the 128-column geometry; N = 576 / 384 / 128, the head and the 1x1 convs: 174 / 120 / 128 us against 147 / 88 / 102)
  const int nrem = (int)(g.N % tecm_p8::BN);
  if (!force && nrem >= 1 && nrem <= 128 && g.N < 768) return 0;
  const int tiles_n = (int)((g.N + tecm_p8::BN - 1) / tecm_p8::BN);
  const int wr = tecm_p8_rows(g.M, g.N);
  const int tiles_m = (int)((g.M + 2 * wr - 1) / (2 * wr));
  const dim3 grid((unsigned)(tiles_m * tiles_n)), block(tecm_p8::NTH);
  if (wr == 128) hipLaunchKernelGGL(gemm_bf16_p8_kernel<128>, grid, block, 0, st, g, tiles_m, tiles_n);
  else if (wr == 112) hipLaunchKernelGGL(gemm_bf16_p8_kernel<112>, grid, block, 0, st, g, tiles_m, tiles_n);
  else hipLaunchKernelGGL(gemm_bf16_p8_kernel<96>, grid, block, 0, st, g, tiles_m, tiles_n);
  TECM_CHECK_LAUNCH("tecm_gemm_bf16/p8");
  return 1;
}
