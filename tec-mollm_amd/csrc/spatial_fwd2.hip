// Forward of the fused stage a-1..a-3, second formulation (round 5): SpatioTemporalEmbedding (modules.py:230-266) +
// GATv2Conv (modules.py:329-336, :356; torch_geometric semantics restated in oracle/ref_cpu.py:gatv2_conv) + residual
// (tec_mollm.py:94) for the configuration TEC_MoLLM.forward runs: block-uniform time features (train.py:65 expands
// them over the nodes with stride 0), embedding tables present, residual on, 24-float output rows.
//
// Why a second kernel.  The first one (spatial_fwd.hip) is a persistent 512-thread block per CU with 138 KiB of LDS --
// a k-major window image as the f32-MFMA operand, two item slots -- and every phase of it runs 2-3x above its
// instruction count because nothing else is resident to cover its latencies (DESIGN B.2, B.5: 0.077 of the HBM
// roofline for three rounds).  This one removes what needed the LDS:
//   * the 22 x 22 input transforms are split by what their input depends on:
//         x_m[g, n] = W_m [x | node_emb[n] + temb_g] + b_m
//                   = A_m x[g, n]  +  P_m[n]  +  c_m(g),     A_m = W_m[:, :Cin],  P_m[n] = W_m[:, Cin:] node_emb[n],
//                                                            c_m(g) = W_m[:, Cin:] temb_g + b_m
//     P_m (N x 24, graph-independent) and c_m (one vector per graph) are computed ONCE per call by a small set-up
//     kernel; per (graph, node) only the Cin-wide part is left -- 10 multiply-adds per output instead of 22 -- as
//     packed FMAs with the weights as SCALAR operands (uniform loads): no operand image, no matrix cores.  (Measured
//     and not kept: the same part on the exact-f32 matrix cores straight from global memory, 32 rows per wave task --
//     219 us against 121: three dependent chains per task, x loads -> 5 MFMAs on one accumulator -> 16 gathers of P.)
//     The two logit helpers u_m = att . x_m per head are formed from the finished row (22 more multiply-adds);
//   * a block is ONE (tile of <= 128 targets, graph) item: 256 threads, 37 KiB of LDS (the x_l rows of the tile's
//     neighbour window and the x_r rows of the tile, later its output) -> four blocks = 16 waves per CU, and the
//     hardware overlaps one block's edge phase with another's transforms and stores;
//   * the CSR slice is read where it lies (L2-resident: consecutive blocks share the tile).
// Edge phase, softmax in base 2, |.|-form of LeakyReLU, dropout of the attention coefficients keyed by the logical
// (row, head, slot) index and the 16-byte output stores are the first kernel's, expression for expression.
//
// Everything the first kernel alone serves (per-node time features, the stand-alone module forwards, odd Cin, output
// pitches other than 24) stays there: tecm_spatial_fwd2_ws_floats() returns 0 and the caller uses tecm_spatial_fwd.
#include "spatial2_common.h"

using namespace tecm_spatial;
using namespace tecm_spatial2;

namespace {

// One thread per (target node, head HH): the first kernel's edge phase reading the CSR from global memory.
template <int HH, int CIN>
__device__ __forceinline__ void edge_phase2(const TecmSpatial& d, const float* __restrict__ xg, const float* __restrict__ emb_row,
                                            const float* __restrict__ temb, const float* xl, float* xr_row, int e0, int deg,
                                            int lo, int wi, uint64_t dbase, const float (&att4)[CH], const float (&bias)[CH],
                                            uint32_t dth, float dinv) {
  float xr[CH + 1];                                          // xr[11] = u_r
  {
    const float4* p = reinterpret_cast<const float4*>(xr_row + HH * 12);
    const float4 a = p[0], b = p[1], c = p[2];
    xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w; xr[4] = b.x; xr[5] = b.y; xr[6] = b.z; xr[7] = b.w;
    xr[8] = c.x; xr[9] = c.y; xr[10] = c.z; xr[11] = c.w;
  }
  float hres[CH];                                            // residual input h[i]: issued now, consumed at the very end
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = HH * CH + c;                              // compile time
    hres[c] = ch < CIN ? xg[ch] : emb_row[ch - CIN] + temb[ch - CIN];
  }
  const int* __restrict__ col = d.colidx + e0;
  const uint64_t dseed = tecm_seed_now(d.alpha_drop.seed, d.alpha_drop.seed_dev);
  const float base = (0.6f * LOG2E) * xr[11];
  float m = -INFINITY, z = 0.f;
  float acc[CH + 1];
#pragma unroll
  for (int c = 0; c <= CH; ++c) acc[c] = 0.f;
  for (int s = 0; s <= deg; s += EU) {                       // slot deg is the implicit self loop
    float a[EU][CH + 1], ev[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int sl = s + u;
      const int j = sl < deg ? col[sl] - lo : wi;            // slots past the self loop re-read it and get weight 0
      const float4* p = reinterpret_cast<const float4*>(xl + j * CP + HH * 12);
      const float4 q0 = p[0], q1 = p[1], q2 = p[2];
      a[u][0] = q0.x; a[u][1] = q0.y; a[u][2] = q0.z; a[u][3] = q0.w; a[u][4] = q1.x; a[u][5] = q1.y;
      a[u][6] = q1.z; a[u][7] = q1.w; a[u][8] = q2.x; a[u][9] = q2.y; a[u][10] = q2.z; a[u][11] = q2.w;
      float e = fmaf(0.6f * LOG2E, q2.w, base);
#pragma unroll
      for (int c = 0; c + 1 < CH; c += 2) {                  // s = x_l[j] + x_r[i] two channels at a time (v_pk_add_f32)
        const f32x2 sv = f32x2{a[u][c], a[u][c + 1]} + f32x2{xr[c], xr[c + 1]};
        e = fmaf(att4[c], fabsf(sv.x), e);
        e = fmaf(att4[c + 1], fabsf(sv.y), e);
      }
      e = fmaf(att4[CH - 1], fabsf(a[u][CH - 1] + xr[CH - 1]), e);
      ev[u] = sl <= deg ? e : -INFINITY;
    }
    float mn = m;                                            // slot s is always valid: mn is finite
#pragma unroll
    for (int u = 0; u < EU; ++u) mn = fmaxf(mn, ev[u]);
    const float corr = __builtin_amdgcn_exp2f(m - mn);       // exp2(-inf) = 0 on the first step
    float pm[EU], zs = 0.f;
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const float p = __builtin_amdgcn_exp2f(ev[u] - mn);
      zs += p;
      pm[u] = p;
      if (dth) pm[u] = p * tecm_drop_mult(dseed, dbase + s + u, dth, dinv);
    }
    z = z * corr + zs;
    m = mn;
#pragma unroll
    for (int c = 0; c < CH + 1; c += 2) {                    // channel pairs (the 12th lane of the pair is the unused u)
      f32x2 t = f32x2{acc[c], acc[c + 1]} * corr;
#pragma unroll
      for (int u = 0; u < EU; ++u) t = f32x2{a[u][c], a[u][c + 1]} * pm[u] + t;
      acc[c] = t.x;
      acc[c + 1] = t.y;
    }
  }
  const float inv = 1.0f / (z + 1e-16f);
  float* o = xr_row + HH * 12;                               // this thread's x_r slice is dead: it becomes the output
#pragma unroll
  for (int c = 0; c < CH; ++c) o[c] = hres[c] + (acc[c] * inv + bias[c]);
}

template <int CIN>
__global__ __launch_bounds__(T2, 4) void spatial_fwd2_kernel(const TecmSpatial d, const float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) float xl[T2 * CP];   // x_l rows of the neighbour window (head-sliced, see slot_of)
  __shared__ __attribute__((aligned(16))) float xr[TN2 * CP];  // x_r rows of the tile, then its output rows
  const int tid = threadIdx.x;
  const int N = d.N, G = d.B * d.L;
  const int item = blockIdx.x;
  const int tile = item / G, gm = item - tile * G;           // tile-major: consecutive blocks share the tile's CSR and P rows
  const int b = gm / d.L, t = gm - b * d.L;
  const bool use_edges = (t * d.B + b) < d.graphs_with_edges; // the reference's flattening is (L*B): g = t*B + b
  const int n0 = tile * d.tile_nodes, n1 = min(N, n0 + d.tile_nodes);
  const int lo = d.tile_lo[tile], hi = d.tile_hi[tile];
  const int wa = use_edges ? 0 : n0 - lo, wb = use_edges ? hi - lo : n1 - lo;   // window rows this graph reads
  const int64_t grow = (int64_t)gm * N;                      // first row of this graph in the (B, L, N, *) tensors
  const float* __restrict__ A = ws + ws_A();
  const float* __restrict__ P = ws + ws_P(N);
  const float* __restrict__ gv = ws + ws_G(N) + (int64_t)gm * GV;

  // ---- phase 1: x_l of the window rows, x_r of the tile rows (thread = window row)
  if (tid >= wa && tid < wb) {
    const int node = lo + tid;
    float x[CIN];
    const float2* xp = reinterpret_cast<const float2*>(d.x + (grow + node) * CIN);
#pragma unroll
    for (int k = 0; k < CIN / 2; ++k) {
      const float2 v = xp[k];
      x[2 * k] = v.x;
      x[2 * k + 1] = v.y;
    }
    transform_row<CIN>(A, P + (int64_t)node * CP, gv, d.att, x, xl + tid * CP);
    if (node >= n0 && node < n1)
      transform_row<CIN>(A + 24 * 16, P + ((int64_t)N + node) * CP, gv + 24, d.att, x, xr + (node - n0) * CP);
  }
  __syncthreads();

  // ---- phase 2: one thread per (target node, head); the head is wave-uniform
  {
    const int tn = tid & (TN2 - 1), hh = tid >> 7;
    const int i = n0 + tn;
    if (i < n1) {
      const uint32_t dth = d.alpha_drop.p > 0.f ? tecm_drop_thresh(d.alpha_drop.p) : 0u;
      const float dinv = d.alpha_drop.p > 0.f ? 1.0f / (1.0f - d.alpha_drop.p) : 1.0f;
      float att4[CH], bias[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        att4[c] = (0.4f * LOG2E) * d.att[hh * CH + c];
        bias[c] = d.bias[hh * CH + c];
      }
      const int e0 = d.rowptr[i];
      const int deg = use_edges ? d.rowptr[i + 1] - e0 : 0;
      const int64_t rowi = (int64_t)(t * d.B + b) * N + i;   // row in the reference's (L*B*N) flattening
      const float* xg = d.x + (grow + i) * CIN;
      const float* emb_row = d.node_tab + (int64_t)i * d.Demb;
      if (hh == 0)
        edge_phase2<0, CIN>(d, xg, emb_row, gv + 48, xl, xr + tn * CP, e0, deg, lo, i - lo,
                            (uint64_t)((rowi * H + 0) * d.alpha_drop.ld), att4, bias, dth, dinv);
      else
        edge_phase2<1, CIN>(d, xg, emb_row, gv + 48, xl, xr + tn * CP, e0, deg, lo, i - lo,
                            (uint64_t)((rowi * H + 1) * d.alpha_drop.ld), att4, bias, dth, dinv);
    }
  }
  __syncthreads();

  // ---- phase 3: the tile leaves in contiguous 16-byte stores; LDS slot = channel + (channel >= 11), columns 22 and 23
  //      of every output row are the zero padding
  {
    const int nf4 = (n1 - n0) * (CP / 4);
    float4* dst = reinterpret_cast<float4*>(d.out + (grow + n0) * CP);
    for (int f = tid; f < nf4; f += T2) {
      const int r = f / (CP / 4), c = 4 * (f - r * (CP / 4));
      const float* src = xr + r * CP;
      float4 v;
      v.x = src[slot_of(c)];
      v.y = src[slot_of(c + 1)];
      v.z = c + 2 < C ? src[slot_of(c + 2)] : 0.f;
      v.w = c + 3 < C ? src[slot_of(c + 3)] : 0.f;
      dst[f] = v;
    }
  }
}

}  // namespace

// Floats of workspace tecm_spatial_fwd2 needs for `d`, or 0 when this formulation does not serve the call (the caller
// then uses tecm_spatial_fwd).  Pure host function.
extern "C" int64_t tecm_spatial_fwd2_ws_floats(const TecmSpatial* dp) {
  if (dp == nullptr || check_common("tecm_spatial_fwd2_ws_floats", *dp) != TECM_OK) return 0;
  if (!v2_eligible(*dp)) return 0;
  return ws_floats(*dp);
}

extern "C" int tecm_spatial_fwd2(const TecmSpatial* dp, float* ws, void* stream) {
  TECM_REQUIRE(dp != nullptr && ws != nullptr, TECM_E_ARG, "tecm_spatial_fwd2: null descriptor / workspace");
  const TecmSpatial& d = *dp;
  const int rc = check_common("tecm_spatial_fwd2", d);
  if (rc) return rc;
  TECM_REQUIRE(v2_eligible(d) && d.out != nullptr, TECM_E_ARG, "tecm_spatial_fwd2: not served (tecm_spatial_fwd2_ws_floats returned 0)");
  TECM_REQUIRE(tecm_aligned(d.x, 8) && tecm_aligned(d.out, 16) && tecm_aligned(ws, 16), TECM_E_ALIGN,
               "tecm_spatial_fwd2: x must be 8-byte, out and the workspace 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nprep = prep_threads(d);
  hipLaunchKernelGGL(spatial_prep_kernel, dim3((unsigned)((nprep + 255) / 256)), dim3(256), 0, st, d, ws);
  TECM_CHECK_LAUNCH("tecm_spatial_fwd2(prep)");
  const int64_t total = (int64_t)d.B * d.L * d.num_tiles;
  if (d.Cin == 10) hipLaunchKernelGGL(spatial_fwd2_kernel<10>, dim3((unsigned)total), dim3(T2), 0, st, d, ws);
  else hipLaunchKernelGGL(spatial_fwd2_kernel<6>, dim3((unsigned)total), dim3(T2), 0, st, d, ws);
  TECM_CHECK_LAUNCH("tecm_spatial_fwd2");
  return TECM_OK;
}
