// extern "C" entry point of the fp32 MFMA GEMM (kernel template: gemm_impl.h; instantiations:
// gemm_mk_nk.hip, gemm_mk_kn.hip, gemm_km_kn.hip -- one translation unit per operand-layout pair).
#include "gemm_bf16_impl.h"
int tecm_gemm_x3_dispatch(const TecmGemm& g, int products, hipStream_t st);      // gemm_x3.hip

namespace {

// Lane = output element (coalesced over each slab), the block's 4 waves take the slabs in turn with four loads in
// flight each: a dW GEMM of this path has up to 256 slabs of a few thousand elements, so one thread walking all
// slabs of its element is a serial chain of ~64 dependent-latency steps (36 us per call, 17 calls per step).
template <int U>   // slabs in flight per lane; the four waves cover 4*U splits per round of memory latency
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const TecmGemm g, int splits) {
  __shared__ float red[4][64];
  const int64_t total = g.M * g.N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const tecm_gemm::DropCtx odc = tecm_gemm::make_drop(g.out_drop);
  for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < total; i0 += (int64_t)gridDim.x * 64) {
    const int64_t i = i0 + lane;
    float v16[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v16[u] = 0.f;
    if (i < total) {
      for (int s = wave; s < splits; s += 4 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {                  // clamped slab + select: the bound is wave-uniform, an `if` would be
          const int sl = s + 4 * u;                     // a branch (and a wait) per load
          const float v = g.workspace[(int64_t)(sl < splits ? sl : splits - 1) * total + i];
          v16[u] += sl < splits ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = U / 2; u > 0; u >>= 1)
#pragma unroll
      for (int v = 0; v < u; ++v) v16[v] += v16[v + u];
    red[wave][lane] = v16[0];
    __syncthreads();
    if (wave == 0 && i < total) {
      const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
      const int64_t m = i / g.N;
      tecm_gemm::epilogue_store(g, odc, m, (int32_t)(i - m * g.N), v);
    }
    __syncthreads();
  }
}

// erf-GELU (nn.GELU() of the prediction head, modules.py:288) is applied by this small elementwise pass after
// the GEMM has written alpha*acc + bias: keeping the erf polynomial out of the GEMM epilogue keeps that
// kernel's code small (it is streamed through the instruction cache once per block).
__global__ __launch_bounds__(256) void erf_post_kernel(float* __restrict__ C, int64_t ldc, const float* __restrict__ src,
                                                       int64_t lds, int64_t M, int32_t N, tecm_gemm::DropCtx odc) {
  const int64_t total = M * N;
  odc.seed = tecm_seed_now(odc.seed, odc.sdev);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N;
    const int32_t n = (int32_t)(i - m * N);
    float v = C[m * ldc + n];
    v = src ? v * dgelu_erf(src[m * lds + n]) : gelu_erf(v);
    if (odc.thresh) v *= tecm_drop_mult(odc.seed, (uint64_t)(m * odc.ld + n), odc.thresh, odc.inv);
    C[m * ldc + n] = v;
  }
}

// widest vector (4, 2 or 1 floats) the operand's address pattern allows
int pick_vec(const float* p, int64_t ld, const TecmWin& w, bool inner_is_k, int64_t K) {
  int v = 4;
  while (v > 1) {
    bool ok = tecm_aligned(p, 4 * v) && (ld % v == 0);
    if (w.enabled) ok = ok && (w.Cw % v == 0);
    if (inner_is_k) ok = ok && (K % v == 0);      // zero-fill past K must be exact
    if (ok) break;
    v >>= 1;
  }
  return v;
}

bool win_ok(const TecmWin& w) {
  return !w.enabled || (w.N > 0 && w.Lin > 0 && w.Lout > 0 && w.stride_t > 0 && w.taps > 0 && w.Cw > 0 && w.pad >= 0);
}

}  // namespace

enum { MODE_F32 = 0, MODE_BF16 = 1, MODE_X3 = 2, MODE_X6 = 3 };
static int gemm_entry(const TecmGemm* d, void* stream, int mode);

extern "C" int tecm_gemm_f32(const TecmGemm* d, void* stream) { return gemm_entry(d, stream, MODE_F32); }

// Same contract as tecm_gemm_f32, operands rounded to bf16 on the way into LDS, fp32 accumulate / epilogue /
// outputs.  Needs 16-byte friendly operands (the float4 loader); returns TECM_E_ALIGN otherwise so that the
// caller can decide to use tecm_gemm_f32 for that call.
extern "C" int tecm_gemm_bf16(const TecmGemm* d, void* stream) { return gemm_entry(d, stream, MODE_BF16); }

// Same contract again, every product evaluated as three bf16 matrix-core products of the hi/lo bf16 split of its
// fp32 factors (gemm_x3_impl.h): ~1e-5 relative error at 3/16 of the exact kernel's matrix-core time.  Serves the
// plain MK x NK contraction only (no a_win / b_win, no a_drop / b_drop, 16-byte friendly operands); anything else
// returns TECM_E_ARG so that the caller runs tecm_gemm_f32 for that call.
extern "C" int tecm_gemm_bf16x3(const TecmGemm* d, void* stream) { return gemm_entry(d, stream, MODE_X3); }
// Three-way split (hi + mid + lo = all 24 mantissa bits), six products: fp32-grade accuracy at 6/16 of the exact time.
extern "C" int tecm_gemm_bf16x6(const TecmGemm* d, void* stream) { return gemm_entry(d, stream, MODE_X6); }

static int gemm_entry(const TecmGemm* d, void* stream, int mode) {
  const bool bf16 = mode == MODE_BF16;
  TECM_REQUIRE(d != nullptr, TECM_E_ARG, "tecm_gemm_f32: null descriptor");
  const TecmGemm& g = *d;
  TECM_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, TECM_E_ARG, "tecm_gemm_f32: M,N,K must be positive (%lld,%lld,%lld)",
               (long long)g.M, (long long)g.N, (long long)g.K);
  TECM_REQUIRE(g.M < (1ll << 31) && g.K < (1ll << 30) && g.N < (1ll << 30), TECM_E_ARG,
               "tecm_gemm_f32: M/N/K too large for 32-bit row arithmetic");
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
  int avec = pick_vec(g.A, g.lda, g.a_win, g.a_layout == TECM_A_MK, g.K);
  int bvec = pick_vec(g.B, g.ldb, g.b_win, g.b_layout == TECM_B_NK, g.K);
  if (g.io_bf16 & TECM_IO_A_BF16) avec = 4;            // bf16 operands are checked separately below
  if (g.io_bf16 & TECM_IO_B_BF16) bvec = 4;
  const bool win = g.a_win.enabled || g.b_win.enabled;
  const bool drop = g.a_drop.p > 0.f || g.b_drop.p > 0.f;
  // float4 epilogue only when everything it touches is 16-byte friendly
  auto ok4 = [](const void* p, int64_t ld) { return p == nullptr || (tecm_aligned(p, 16) && ld % 4 == 0); };
  const bool c16 = (g.io_bf16 & TECM_IO_C_BF16) != 0;
  const bool p16 = (g.io_bf16 & TECM_IO_PRE_BF16) != 0;
  auto ok4h = [](const void* p, int64_t ld) { return p == nullptr || (tecm_aligned(p, 8) && ld % 4 == 0); };   // bf16 quads
  const bool vec4 = g.N % 4 == 0 && (c16 ? ok4h(g.C, g.ldc) : ok4(g.C, g.ldc)) && ok4(g.bias, 4) && ok4(g.rowbias, g.rb_ld) &&
                    (p16 ? ok4h(g.preact, g.ldp) && ok4h(g.dact_src, g.ldd) : ok4(g.preact, g.ldp) && ok4(g.dact_src, g.ldd)) &&
                    ok4(g.residual, g.ldr) &&
                    (!g.c_win.enabled || g.c_win.Cw % 4 == 0) &&
                    (g.split_k <= 1 || tecm_aligned(g.workspace, 16));
  if (c16) TECM_REQUIRE(vec4, TECM_E_ALIGN, "tecm_gemm_bf16: bf16 C needs 16-byte friendly epilogue operands");
  if (bf16)
    TECM_REQUIRE(avec == 4 && bvec == 4, TECM_E_ALIGN,
                 "tecm_gemm_bf16: operands must be 16-byte aligned with leading dims / K / Cw multiples of 4");
  if (mode == MODE_X3 || mode == MODE_X6) {
    TECM_REQUIRE(g.a_layout == TECM_A_MK && g.b_layout == TECM_B_NK && !win && !drop, TECM_E_ARG,
                 "tecm_gemm_bf16x3: serves the plain MK x NK contraction only (no operand windows / prologue dropout)");
    TECM_REQUIRE(avec == 4 && bvec == 4, TECM_E_ALIGN,
                 "tecm_gemm_bf16x3: operands must be 16-byte aligned with leading dims / K multiples of 4");
  }
  const int io = g.io_bf16 & (TECM_IO_A_BF16 | TECM_IO_B_BF16 | TECM_IO_C_BF16 | TECM_IO_PRE_BF16);
  TECM_REQUIRE(io == g.io_bf16, TECM_E_ARG, "tecm_gemm: unknown io_bf16 bits");
  if (io) {
    TECM_REQUIRE(bf16, TECM_E_ARG, "tecm_gemm: bf16 tensors in HBM are served by tecm_gemm_bf16 only");
    if (io & (TECM_IO_A_BF16 | TECM_IO_B_BF16))
      TECM_REQUIRE(!drop, TECM_E_ARG, "tecm_gemm_bf16: bf16 operands take no prologue dropout");
    // a 16-byte vector of 8 bf16 runs along the operand's contiguous dimension: it must not straddle the end of that
    // dimension or a window tap
    if (io & TECM_IO_A_BF16) {
      TECM_REQUIRE(tecm_aligned(g.A, 16) && g.lda % 8 == 0, TECM_E_ALIGN, "tecm_gemm_bf16: bf16 A must be 16-byte friendly");
      TECM_REQUIRE((g.a_layout == TECM_A_MK ? g.K : g.M) % 8 == 0 && (!g.a_win.enabled || g.a_win.Cw % 8 == 0), TECM_E_ARG,
                   "tecm_gemm_bf16: bf16 A needs its contiguous extent (K for MK, M for KM) and a_win.Cw to be multiples of 8");
    }
    if (io & TECM_IO_B_BF16) {
      TECM_REQUIRE(tecm_aligned(g.B, 16) && g.ldb % 8 == 0, TECM_E_ALIGN, "tecm_gemm_bf16: bf16 B must be 16-byte friendly");
      TECM_REQUIRE((g.b_layout == TECM_B_NK ? g.K : g.N) % 8 == 0 && (!g.b_win.enabled || g.b_win.Cw % 8 == 0), TECM_E_ARG,
                   "tecm_gemm_bf16: bf16 B needs its contiguous extent (K for NK, N for KN) and b_win.Cw to be multiples of 8");
    }
    if (io & TECM_IO_PRE_BF16)
      TECM_REQUIRE(vec4 && (g.preact || g.dact_src) && g.split_k <= 1 && g.act != TECM_ACT_GELU_ERF && !g.c_win.enabled &&
                       !g.rowbias,
                   TECM_E_ARG, "tecm_gemm_bf16: a bf16 preact / dact_src needs the 16-byte friendly plain tanh-GELU epilogue");
    if (io & TECM_IO_C_BF16)
      TECM_REQUIRE(vec4 && !g.residual && !g.accumulate && !g.c_win.enabled && g.split_k <= 1 &&
                       g.act != TECM_ACT_GELU_ERF,
                   TECM_E_ARG, "tecm_gemm_bf16: bf16 C needs a 16-byte friendly plain epilogue");
  }
  TecmGemm gk = g;                 // private copy: io_bf16 also carries the epilogue-vectorisation flag to the kernel
  gk.io_bf16 = io | (vec4 ? TECM_P0_VEC4 : 0);
  const bool erf = g.act == TECM_ACT_GELU_ERF;
  if (erf) {                       // GEMM writes the pre-activation, erf_post_kernel finishes (see above)
    TECM_REQUIRE(!g.residual && !g.accumulate && !g.c_win.enabled, TECM_E_ARG,
                 "tecm_gemm_f32: GELU_ERF cannot be combined with residual / accumulate / c_win");
    gk.act = TECM_ACT_NONE;
    gk.dact_src = nullptr;
    gk.out_drop.p = 0.f;
  }
  int splits;
  if (mode == MODE_X3 || mode == MODE_X6) {
    splits = tecm_gemm_x3_dispatch(gk, mode == MODE_X6 ? 6 : 3, st);
  } else if (bf16) {
    if (g.a_layout == TECM_A_MK && g.b_layout == TECM_B_NK)
      splits = tecm_gemm16_dispatch_mk_nk(gk, win, drop, st);
    else if (g.a_layout == TECM_A_MK)
      splits = tecm_gemm16_dispatch_mk_kn(gk, win, drop, st);
    else
      splits = tecm_gemm16_dispatch_km_kn(gk, win, drop, st);
  } else if (g.a_layout == TECM_A_MK && g.b_layout == TECM_B_NK)
    splits = tecm_gemm_dispatch_mk_nk(gk, avec, bvec, win, drop, st);
  else if (g.a_layout == TECM_A_MK)
    splits = tecm_gemm_dispatch_mk_kn(gk, avec, bvec, win, drop, st);
  else
    splits = tecm_gemm_dispatch_km_kn(gk, avec, bvec, win, drop, st);
  if (splits < 0) return splits;
  if (splits > 1) {
    const int64_t total = g.M * g.N;
    const int64_t want = (total + 63) / 64;
    const int blocks = (int)(want < 4096 ? want : 4096);
    if (splits > 32)
      hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3(blocks), dim3(256), 0, st, gk, splits);
    else if (splits > 16)
      hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3(blocks), dim3(256), 0, st, gk, splits);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, st, gk, splits);
    TECM_CHECK_LAUNCH("tecm_gemm_f32/splitk_reduce");
  }
  if (erf) {
    const int64_t want = (g.M * g.N + 255) / 256;
    hipLaunchKernelGGL(erf_post_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, st, g.C, g.ldc,
                       g.dact_src, g.ldd, g.M, (int32_t)g.N, tecm_gemm::make_drop_host(g.out_drop));
    TECM_CHECK_LAUNCH("tecm_gemm_f32/erf_post");
  }
  return TECM_OK;
}
