// Backward of the fused stage a-1..a-3, second formulation (round 5; the forward: spatial_fwd2.hip; reference
// SpatioTemporalEmbedding modules.py:230-266 + GATv2Conv modules.py:329-336, :356 + residual tec_mollm.py:94).  Nothing
// was saved by the forward: x_l / x_r are recomputed, and the kernel reduces straight to PARAMETER gradients -- x needs
// none.  Served: the configuration TEC_MoLLM.forward runs (block-uniform time features, flags = 0, 24-float rows);
// everything else stays on spatial_bwd.hip.
//
// Against the first formulation (one persistent 512-thread block per CU, 146 KiB of LDS, every phase latency-bound):
//   * a block is (tile of <= 128 targets, chunk of graphs): 256 threads, ~45 KiB of LDS -> three blocks per CU;
//   * x_m = A_m x + P_m[n] + c_m(g) as in the forward (set-up launch, spatial2_common.h): no window image, no matrix cores
//     in the recomputation;
//   * two LDS regions are reused phase by phase instead of one array per tensor:
//       R1 (256 x 24): x_l rows of the window -> [x_r | dout] rows of the tile -> d x_l rows
//       R2           : x_r rows (hand-over) -> per-edge (e, dalpha) -> (alpha~, de) -> d x_r rows
//   * what the weight gradients need of the embedding half of h is linear in per-row sums, so it is taken from REGISTER
//     sums over the block's graphs: dW_m[:, Cin:] = sum_n S_m[n] (x) node_emb[n] + sum_g R_m(g) (x) temb_g with
//     S_m[n] = sum_g d x_m[g, n] (per thread) and R_m(g) = sum_n d x_m[g, n] (per item column sums); only the Cin-wide part
//     dW_m[:, :Cin] = sum d x_m (x) x (and the bias, a ones column) is an outer product per item, on the f32 matrix cores
//     (v_mfma_f32_16x16x4_f32, x rows straight from global memory, accumulators live across the block's items).
// Per item (eight block barriers):
//   P1  x_l (window) -> R1, x_r (tile) -> R2                       thread = window row
//   B1  by target, thread (tile node, head): sweep 1 logits e and dalpha = <dout_i, x_l[j]> with online-softmax statistics,
//       (e, dalpha) parked in R2; sweep 2 turns them into (alpha~, de), accumulates d x_r and d att
//   B2  by source, thread = window row, both heads: d x_l[w] = sum_i alpha~_iw dout_i + de_iw att (.) lrelu'(x_l[w] + x_r[i])
//       over the tile's edges that leave w (host-built by-source lists) -- no float atomics
//   P4  d x_l -> R1, d x_r -> R2; outer products with [x | 1] on the matrix cores; per-item column sums; the temporal
//       tables take d temb_g = W_l[:, Cin:]^T R_l + W_r[:, Cin:]^T R_r + sum_n dout[n, Cin:] through four atomics per column.
// Block end: node-embedding halves of dW_l / dW_r (one more outer-product pass over the register sums), node-table rows by
// atomics, everything else into this block's row of `partials` in the first formulation's layout
// [dWl (C*C) | dbl (C) | dWr (C*C) | dbr (C) | datt (C) | dbias (C, unused)] -- the caller's column sum is unchanged.
#include <cstdlib>
#include "spatial2_common.h"

using namespace tecm_spatial;
using namespace tecm_spatial2;

namespace {

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct Bwd2Args {
  TecmSpatial d;
  TecmSpatialGrads g;
  const float* ws;
  int nch, gc;                 // chunks of graphs per tile, graphs per chunk
};

constexpr int PLD = 2 * C * C + 4 * C;

__device__ __forceinline__ void ld12(const float* p, float (&v)[CH + 1]) {
  const float4* q = reinterpret_cast<const float4*>(p);
  const float4 a = q[0], b = q[1], c = q[2];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ __forceinline__ void st12(float* p, const float (&v)[CH], float last) {
  float4* q = reinterpret_cast<float4*>(p);
  q[0] = make_float4(v[0], v[1], v[2], v[3]);
  q[1] = make_float4(v[4], v[5], v[6], v[7]);
  q[2] = make_float4(v[8], v[9], v[10], last);
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// LDS map (floats)
struct Map2 {
  int r1, r2, colp, sums, gsp, ecol, scol, we, xs, total;
};
__host__ __device__ inline Map2 make_map2(int tile_edges_max) {
  Map2 m;
  m.r1 = 0;                                                   // 256 x 24
  m.r2 = T2 * CP;                                             // max(edge array (E + 128) x 2 heads x 2, 128 x 24)
  int r2f = (tile_edges_max + TN2) * 4;
  if (r2f < TN2 * CP) r2f = TN2 * CP;
  m.colp = m.r2 + ((r2f + 3) & ~3);                           // [8][64] partial column sums (d x_l at 0, d x_r at 32)
  m.sums = m.colp + 8 * 64;                                   // [96]: R_l (24) | pad | R_r (24) | pad | G (24)
  m.gsp = m.sums + 96;                                        // [4][12] per-wave sums of dout rows
  m.ecol = m.gsp + 48;                                        // [E] window-relative source of every edge of the tile (by target)
  m.scol = m.ecol + ((tile_edges_max + 3) & ~3);              // [E] the same edges grouped by source: (target << 16) | position
  m.we = m.scol + ((tile_edges_max + 3) & ~3);                // [2][22][16] W_m[ch][Cin + e]: the embedding halves of W_l, W_r
  m.xs = m.we + 2 * C * 16;                                   // [256][12] x rows of the window (B operand of the outer products)
  m.total = m.xs + T2 * 12;
  return m;
}

#ifndef SPB2_OCC
#define SPB2_OCC 2
#endif
#ifndef SPB2_SKIP
#define SPB2_SKIP 0            // diagnostics (tools/build_variant.py): bit0 no B1 sweeps, bit1 no B2 gather, bit2 no outer products /
#endif                         // column sums, bit3 no per-item consumers (temporal atomics, temb part), bit4 no transforms
template <int CIN>
__global__ __launch_bounds__(T2, SPB2_OCC) void spatial_bwd2_kernel(const Bwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TecmSpatial& d = a.d;
  const TecmSpatialGrads& gr = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = d.N, G = d.B * d.L, Demb = d.Demb;
  const int tile = blockIdx.x / a.nch, chunk = blockIdx.x - tile * a.nch;
  const int g0 = chunk * a.gc, g1 = min(G, g0 + a.gc);
  const int n0 = tile * d.tile_nodes, n1 = min(N, n0 + d.tile_nodes);
  const int lo = d.tile_lo[tile], hi = d.tile_hi[tile];
  const int ebase = d.rowptr[n0], E = d.rowptr[n1] - ebase;  // the tile's edge segment of the by-target CSR
  const Map2 m = make_map2(d.tile_edges_max);
  float* R1 = smem + m.r1;
  float* R2 = smem + m.r2;
  const float* __restrict__ A = a.ws + ws_A();
  const float* __restrict__ P = a.ws + ws_P(N);
  // the tile's CSR slices, once per block (the tile is the same for every graph of the chunk): sources by target and the
  // by-source lists in LDS, this thread's own list bounds in registers
  int* ecol = reinterpret_cast<int*>(smem + m.ecol);
  int* scol = reinterpret_cast<int*>(smem + m.scol);
  for (int e = tid; e < E; e += T2) {
    ecol[e] = d.colidx[ebase + e] - lo;
    scol[e] = gr.src_col[ebase + e];
  }
  for (int k = tid; k < 2 * C * 16; k += T2) {
    const int mm = k / (C * 16), r = k - mm * C * 16, ch = r >> 4, e = r & 15;
    smem[m.we + k] = e < Demb ? (mm ? d.Wr : d.Wl)[ch * C + CIN + e] : 0.f;
  }
  const int* __restrict__ sptr = gr.src_ptr + gr.src_ptr_off[tile];   // by-source lists of this tile (window-row indexed)
  const int sq0 = tid < hi - lo ? sptr[tid] : 0, sq1 = tid < hi - lo ? sptr[tid + 1] : 0;

  // thread roles: B1 / tile rows: (tn, hh); B2 / window rows: w = tid
  const int tn = tid & (TN2 - 1), hh = tid >> 7;
  const int i = n0 + tn;
  const bool tgt = i < n1;
  const int te0 = tgt ? d.rowptr[i] - ebase : 0, tdeg = tgt ? d.rowptr[i + 1] - d.rowptr[i] : 0;
  float att[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) att[c] = d.att[hh * CH + c];
  const uint32_t dth = d.alpha_drop.p > 0.f ? tecm_drop_thresh(d.alpha_drop.p) : 0u;
  const float dinv = d.alpha_drop.p > 0.f ? 1.0f / (1.0f - d.alpha_drop.p) : 1.0f;
  const uint64_t dseed = tecm_seed_now(d.alpha_drop.seed, d.alpha_drop.seed_dev);

  // ---- sums that live across the block's graphs
  float SL[C], SR[CH], SG[CH], datt[CH];
#pragma unroll
  for (int c = 0; c < C; ++c) SL[c] = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) SR[c] = SG[c] = datt[c] = 0.f;
  f32x4 accL[2] = {zero4(), zero4()}, accR[2] = {zero4(), zero4()};
  float tw[3] = {0.f, 0.f, 0.f};                            // dW_m[ch][Cin + e] temb part: index tid + 256 j -> (m, ch, e)
  int tw_sum[3], tw_e[3];                                    // ... where its R_m[ch] sits in the column sums, and e (-1: none)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int q = tid + 256 * j;
    const bool on = q < 2 * C * Demb;
    const int mm = on ? q / (C * Demb) : 0, r = q - mm * C * Demb, ch = on ? r / Demb : 0;
    tw_sum[j] = mm * 32 + slot_of(ch);
    tw_e[j] = on ? r - ch * Demb : -1;
  }

  if (g0 >= g1) return;
  // this thread's x row of the NEXT graph is requested one item ahead (registers; the load's latency is an item long)
  float xpre[CIN];
  auto fetch_x = [&](int gmx) {
    const float2* xp = reinterpret_cast<const float2*>(d.x + ((int64_t)gmx * N + min(lo + tid, N - 1)) * CIN);
#pragma unroll
    for (int k = 0; k < CIN / 2; ++k) {
      const float2 v = xp[k];
      xpre[2 * k] = v.x;
      xpre[2 * k + 1] = v.y;
    }
  };
  fetch_x(g0);
  for (int gm = g0; gm < g1; ++gm) {
    const int b = gm / d.L, t = gm - b * d.L;
    const bool use_edges = (t * d.B + b) < d.graphs_with_edges;
    const int wa = use_edges ? 0 : n0 - lo, wb = use_edges ? hi - lo : n1 - lo;
    const int64_t grow = (int64_t)gm * N;
    const float* __restrict__ gv = a.ws + ws_G(N) + (int64_t)gm * GV;
    const bool inwin = tid >= wa && tid < wb;

    // ---- P1: x_l of the window rows -> R1, x_r of the tile rows -> R2; the x row itself (and the ones column) -> XS
    {
      float4* xs4 = reinterpret_cast<float4*>(smem + m.xs + tid * 12);
      const bool on = inwin;
      xs4[0] = make_float4(on ? xpre[0] : 0.f, on ? xpre[1] : 0.f, on ? xpre[2] : 0.f, on ? xpre[3] : 0.f);
      if constexpr (CIN == 10) {
        xs4[1] = make_float4(on ? xpre[4] : 0.f, on ? xpre[5] : 0.f, on ? xpre[6] : 0.f, on ? xpre[7] : 0.f);
        xs4[2] = make_float4(on ? xpre[8] : 0.f, on ? xpre[9] : 0.f, on ? 1.f : 0.f, 0.f);
      } else {
        xs4[1] = make_float4(on ? xpre[4] : 0.f, on ? xpre[5] : 0.f, on ? 1.f : 0.f, 0.f);
        xs4[2] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (inwin && !(SPB2_SKIP & 16)) {
      const int node = lo + tid;
      float x[CIN];
#pragma unroll
      for (int k = 0; k < CIN; ++k) x[k] = xpre[k];
      transform_row<CIN>(A, P + (int64_t)node * CP, gv, d.att, x, R1 + tid * CP);
      if (node >= n0 && node < n1)
        transform_row<CIN>(A + 24 * 16, P + ((int64_t)N + node) * CP, gv + 24, d.att, x, R2 + (node - n0) * CP);
    }
    __syncthreads();
    fetch_x(min(gm + 1, g1 - 1));

    // ---- B1, by target: thread (tile node tn, head hh)
    float xr[CH + 1], gvv[CH + 1], dxr[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) dxr[c] = 0.f;
#pragma unroll
    for (int c = 0; c <= CH; ++c) xr[c] = gvv[c] = 0.f;
    if (tgt) {
      ld12(R2 + tn * CP + hh * 12, xr);
      const float* gp = gr.dout + (grow + i) * CP + hh * CH;
#pragma unroll
      for (int c = 0; c < CH; ++c) gvv[c] = gp[c];
    }
    __syncthreads();                                         // R2 becomes the edge array
    if (tgt && !(SPB2_SKIP & 1)) {
      const int wi = i - lo;
      const int e0 = te0;
      const int deg = use_edges ? tdeg : 0;
      const int* col = ecol + e0;
      const int64_t rowi = (int64_t)(t * d.B + b) * N + i;   // row in the reference's (L*B*N) flattening
      const uint64_t dbase = (uint64_t)((rowi * H + hh) * d.alpha_drop.ld);
      const float base = (0.6f * LOG2E) * xr[CH];
      // sweep 1: logits, dalpha, online softmax statistics (slot deg = the implicit self loop), three slots per step: the
      // LDS round trips of a step overlap (col -> x_l row -> arithmetic is a dependent chain per slot)
      constexpr int BU = 1;                                // (three slots per step measured slower: 745 vs 696 us -- 21 more spilled registers)
      float mx = -INFINITY, z = 0.f, num = 0.f;
      for (int s = 0; s <= deg; s += BU) {
        float ev[BU], dav[BU];
#pragma unroll
        for (int u = 0; u < BU; ++u) {
          const int sl = s + u;
          const bool valid = sl <= deg;
          const int j = sl < deg ? col[sl] : wi;
          const int pos = sl < deg ? e0 + sl : E + tn;
          float al[CH + 1];
          ld12(R1 + j * CP + hh * 12, al);
          float t0 = 0.f, u0 = 0.f, da = 0.f, db = 0.f;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            if (c & 1) {
              u0 = fmaf(att[c], fabsf(al[c] + xr[c]), u0);
              db = fmaf(gvv[c], al[c], db);
            } else {
              t0 = fmaf(att[c], fabsf(al[c] + xr[c]), t0);
              da = fmaf(gvv[c], al[c], da);
            }
          }
          const float e = fmaf(0.4f * LOG2E, t0 + u0, fmaf(0.6f * LOG2E, al[CH], base));
          da += db;
          if (dth) da *= tecm_drop_mult(dseed, dbase + sl, dth, dinv);
          if (valid) *reinterpret_cast<float2*>(R2 + (pos * 2 + hh) * 2) = make_float2(e, da);
          ev[u] = valid ? e : -INFINITY;
          dav[u] = valid ? da : 0.f;
        }
        float mn = mx;
#pragma unroll
        for (int u = 0; u < BU; ++u) mn = fmaxf(mn, ev[u]);  // slot s is always valid: mn is finite
        const float corr = __builtin_amdgcn_exp2f(mx - mn);
        float zs = 0.f, ns = 0.f;
#pragma unroll
        for (int u = 0; u < BU; ++u) {
          const float pw = __builtin_amdgcn_exp2f(ev[u] - mn);
          zs += pw;
          ns = fmaf(pw, dav[u], ns);
        }
        z = z * corr + zs;
        num = num * corr + ns;
        mx = mn;
      }
      const float zinv = 1.0f / (z + 1e-16f);
      const float dot = num * zinv;
      // sweep 2: (alpha~, de) per edge into the edge array; d x_r and d att
      for (int s = 0; s <= deg; s += BU) {
#pragma unroll
        for (int u = 0; u < BU; ++u) {
          const int sl = s + u;
          const bool valid = sl <= deg;
          const int j = sl < deg ? col[sl] : wi;
          const int pos = sl < deg ? e0 + sl : E + tn;
          float2* slot = reinterpret_cast<float2*>(R2 + (pos * 2 + hh) * 2);
          const float2 ed = *slot;                           // (e, dalpha * mult)
          float al[CH + 1];
          ld12(R1 + j * CP + hh * 12, al);
          const float alpha = __builtin_amdgcn_exp2f(ed.x - mx) * zinv;
          float mult = 1.0f;
          if (dth) mult = tecm_drop_mult(dseed, dbase + sl, dth, dinv);
          const float de = valid ? alpha * (ed.y - dot) : 0.f;
          if (valid) *slot = make_float2(alpha * mult, de);
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const float sv = al[c] + xr[c];
            const bool pos_ = sv > 0.f;
            dxr[c] = fmaf(de, pos_ ? att[c] : NEG_SLOPE * att[c], dxr[c]);
            datt[c] = fmaf(de, pos_ ? sv : NEG_SLOPE * sv, datt[c]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        SR[c] += dxr[c];
        SG[c] += gvv[c];
      }
    }
    // per-wave sums of the dout rows (the temporal tables need sum_n dout[n, Cin:]); waves 0, 1 = head 0, waves 2, 3 = head 1
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float s = wave_sum(gvv[c]);
      if (lane == 0) smem[m.gsp + wave * 12 + c] = s;
    }
    // this thread's own x_l row (window row w = tid) for B2, before R1 is reused
    float xlw[CP];
    {
      const float4* q = reinterpret_cast<const float4*>(R1 + tid * CP);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float4 v = inwin ? q[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        xlw[4 * k] = v.x; xlw[4 * k + 1] = v.y; xlw[4 * k + 2] = v.z; xlw[4 * k + 3] = v.w;
      }
    }
    __syncthreads();                                         // sweeps done, own rows read: R1 becomes [x_r | dout] of the tile
    if (tgt) {
      float xr11[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) xr11[c] = xr[c];
      st12(R1 + tn * CP + hh * 12, xr11, 0.f);
      float g11[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) g11[c] = gvv[c];
      st12(R1 + TN2 * CP + tn * CP + hh * 12, g11, 0.f);
    }
    __syncthreads();

    // ---- B2, by source: thread = window row, both heads
    float DXL[C];
#pragma unroll
    for (int c = 0; c < C; ++c) DXL[c] = 0.f;
    if (inwin && !(SPB2_SKIP & 2)) {
      const int w = tid;
      const bool self = w >= n0 - lo && w < n1 - lo;
      const int q0 = use_edges ? sq0 : 0;
      const int q1 = use_edges ? sq1 : 0;
      const int selfc = ((w - (n0 - lo)) << 16) | (E + w - (n0 - lo));   // the implicit self loop of a tile row: last "edge"
      for (int qq = q0; qq < q1 + (self ? 1 : 0); ++qq) {
        const int code = qq < q1 ? scol[qq] : selfc;         // (tile target << 16) | position in the edge array
        const int tt = code >> 16;
        const float4 ad = *reinterpret_cast<const float4*>(R2 + (code & 0xffff) * 4);   // (alpha~, de) of head 0, head 1
#pragma unroll
        for (int h2 = 0; h2 < H; ++h2) {
          if (h2) __builtin_amdgcn_sched_barrier(0);         // one head at a time: the two would double the live registers
          float xt[CH + 1], gt[CH + 1];
          ld12(R1 + tt * CP + h2 * 12, xt);
          ld12(R1 + TN2 * CP + tt * CP + h2 * 12, gt);
          const float al = h2 ? ad.z : ad.x, de = h2 ? ad.w : ad.y;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const float sv = xlw[h2 * 12 + c] + xt[c];
            const float ac = d.att[h2 * CH + c];
            DXL[h2 * CH + c] = fmaf(al, gt[c], fmaf(de, sv > 0.f ? ac : NEG_SLOPE * ac, DXL[h2 * CH + c]));
          }
        }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) SL[c] += DXL[c];
    }
    __syncthreads();                                         // [x_r | dout] and the edge array are dead

    // ---- P4: d x_l rows -> R1, d x_r rows -> R2 (slot order, u slots zero; rows outside the window zero)
    {
      float4* q = reinterpret_cast<float4*>(R1 + tid * CP);
      q[0] = make_float4(DXL[0], DXL[1], DXL[2], DXL[3]);
      q[1] = make_float4(DXL[4], DXL[5], DXL[6], DXL[7]);
      q[2] = make_float4(DXL[8], DXL[9], DXL[10], 0.f);
      q[3] = make_float4(DXL[11], DXL[12], DXL[13], DXL[14]);
      q[4] = make_float4(DXL[15], DXL[16], DXL[17], DXL[18]);
      q[5] = make_float4(DXL[19], DXL[20], DXL[21], 0.f);
      st12(R2 + tn * CP + hh * 12, dxr, 0.f);                // zeros for tn >= tile size
    }
    __syncthreads();
    if (!(SPB2_SKIP & 4)) {
      // outer products with [x | 1]: wave v takes window rows 64 v .. and tile rows 32 v ..; 4 rows per MFMA.
      // A[i = slot][k = row] from LDS, B[k = row][j = input column] from global memory (x, then the ones column)
      const int i0 = lane & 15, k4 = lane >> 4;
      const float* XS = smem + m.xs;
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const int row = 64 * wave + 4 * ks + k4;
        const float bx = i0 < 12 ? XS[row * 12 + i0] : 0.f;   // x | 1 | 0 (zero rows outside the window)
        const float a0 = R1[row * CP + i0];
        const float a1 = i0 < CP - 16 ? R1[row * CP + 16 + i0] : 0.f;
        accL[0] = MFMA16(a0, bx, accL[0]);
        accL[1] = MFMA16(a1, bx, accL[1]);
      }
#pragma unroll 4
      for (int ks = 0; ks < 8; ++ks) {
        const int tr = 32 * wave + 4 * ks + k4;
        const int wrow = min(n0 - lo + tr, T2 - 1);           // the tile row inside the window (d x_r rows beyond the tile are zero)
        const float bx = i0 < 12 ? XS[wrow * 12 + i0] : 0.f;
        const float a0 = R2[tr * CP + i0];
        const float a1 = i0 < CP - 16 ? R2[tr * CP + 16 + i0] : 0.f;
        accR[0] = MFMA16(a0, bx, accR[0]);
        accR[1] = MFMA16(a1, bx, accR[1]);
      }
      // partial column sums of d x_l (rows 32 rq ..) and d x_r (rows 16 rq ..): thread (column, row group)
      const int colc = tid & 31, rq = tid >> 5;
      if (colc < CP) {
        float sl = 0.f, sr = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) sl += R1[(32 * rq + r) * CP + colc];
#pragma unroll 8
        for (int r = 0; r < 16; ++r) sr += R2[(16 * rq + r) * CP + colc];
        smem[m.colp + rq * 64 + colc] = sl;
        smem[m.colp + rq * 64 + 32 + colc] = sr;
      }
    }
    __syncthreads();                                         // R1 / R2 are free for the next item from here on
    if (tid < 96) {
      const int which = tid >> 5, colc = tid & 31;           // 0: R_l, 1: R_r, 2: G (sums of the dout rows)
      float v = 0.f;
      if (colc < CP) {
        if (which < 2) {
#pragma unroll
          for (int rq = 0; rq < 8; ++rq) v += smem[m.colp + rq * 64 + which * 32 + colc];
        } else if (colc != CH && colc != 2 * CH + 1) {
          const int h2 = colc > CH ? 1 : 0, c = colc - h2 * 12;
          v = smem[m.gsp + (2 * h2) * 12 + c] + smem[m.gsp + (2 * h2 + 1) * 12 + c];
        }
      }
      smem[m.sums + which * 32 + colc] = v;
    }
    __syncthreads();
    // ---- per-item consumers of the column sums
    if (!(SPB2_SKIP & 8)) {
      // dW_m[ch][Cin + e] += R_m[ch] temb_g[e]
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (tw_e[j] >= 0) tw[j] = fmaf(smem[m.sums + tw_sum[j]], gv[48 + tw_e[j]], tw[j]);
      // temporal tables: d temb_g[e] = sum_ch W_l[ch][Cin+e] R_l[ch] + W_r[ch][Cin+e] R_r[ch] + sum_n dout[n][Cin+e]; four lanes
      // per column e: lane part p sums 11 of the 44 terms (matrix p >> 1, half p & 1)
      if (tid < 4 * Demb) {
        const int e = tid >> 2, p4 = tid & 3;
        const float* cs = smem + m.sums + (p4 >> 1) * 32;
        const float* W = smem + m.we + (p4 >> 1) * C * 16 + e;
        const int abeg = (p4 & 1) * CH;
        float v = p4 == 0 ? smem[m.sums + 2 * 32 + slot_of(CIN + e)] : 0.f;
#pragma unroll
        for (int q = 0; q < CH; ++q) v = fmaf(cs[slot_of(abeg + q)], W[(abeg + q) * 16], v);
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        if (p4 == 0) {
        const TimeIdx ti = load_time_idx(d, b, t, 0);
        atomicAdd(&gr.d_tod_tab[ti.tod * Demb + e], v);
        atomicAdd(&gr.d_doy_tab[ti.doy * Demb + e], v);
        atomicAdd(&gr.d_year_tab[ti.year * Demb + e], v);
        atomicAdd(&gr.d_season_tab[ti.season * Demb + e], v);
        }
      }
    }
  }

  // ================================================================= block results
  __syncthreads();
  // ---- node-embedding halves: dW_m[:, Cin + e] += sum_rows S_m[row] (x) node_emb[node(row)][e]; node-table rows
  {
    float4* q = reinterpret_cast<float4*>(R1 + tid * CP);
    q[0] = make_float4(SL[0], SL[1], SL[2], SL[3]);
    q[1] = make_float4(SL[4], SL[5], SL[6], SL[7]);
    q[2] = make_float4(SL[8], SL[9], SL[10], 0.f);
    q[3] = make_float4(SL[11], SL[12], SL[13], SL[14]);
    q[4] = make_float4(SL[15], SL[16], SL[17], SL[18]);
    q[5] = make_float4(SL[19], SL[20], SL[21], 0.f);
    st12(R2 + tn * CP + hh * 12, SR, 0.f);
  }
  // node table: d node_emb[n][e] = sum_ch W_l[ch][Cin+e] S_l[n][ch]  (window rows)
  //                              + sum_ch W_r[ch][Cin+e] S_r[n][ch] + sum_g dout[g, n, Cin+e]  (tile rows, per head)
  if (lo + tid < hi) {
    for (int e = 0; e < Demb; ++e) {
      float v = 0.f;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) v = fmaf(d.Wl[ch * C + CIN + e], SL[ch], v);
      if (v != 0.f) atomicAdd(&gr.d_node_tab[(int64_t)(lo + tid) * Demb + e], v);
    }
  }
  if (tgt) {
    for (int e = 0; e < Demb; ++e) {
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c) v = fmaf(d.Wr[(hh * CH + c) * C + CIN + e], SR[c], v);
      const int cg = CIN + e - hh * CH;                      // the residual's column Cin + e, if it belongs to this head
      if (cg >= 0 && cg < CH) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (c == cg) v += SG[c];
      }
      atomicAdd(&gr.d_node_tab[(int64_t)i * Demb + e], v);
    }
  }
  __syncthreads();
  f32x4 accLe[2] = {zero4(), zero4()}, accRe[2] = {zero4(), zero4()};
  {
    const int i0 = lane & 15, k4 = lane >> 4;
#pragma unroll 2
    for (int ks = 0; ks < 16; ++ks) {
      const int row = 64 * wave + 4 * ks + k4;
      const bool ok = lo + row < hi;
      const int node = min(lo + row, N - 1);
      const float bx = (ok && i0 < Demb) ? d.node_tab[(int64_t)node * Demb + i0] : 0.f;
      const float a0 = R1[row * CP + i0];
      const float a1 = i0 < CP - 16 ? R1[row * CP + 16 + i0] : 0.f;
      accLe[0] = MFMA16(a0, bx, accLe[0]);
      accLe[1] = MFMA16(a1, bx, accLe[1]);
    }
#pragma unroll 2
    for (int ks = 0; ks < 8; ++ks) {
      const int tr = 32 * wave + 4 * ks + k4;
      const bool ok = n0 + tr < n1;
      const int node = min(n0 + tr, N - 1);
      const float bx = (ok && i0 < Demb) ? d.node_tab[(int64_t)node * Demb + i0] : 0.f;
      const float a0 = R2[tr * CP + i0];
      const float a1 = i0 < CP - 16 ? R2[tr * CP + 16 + i0] : 0.f;
      accRe[0] = MFMA16(a0, bx, accRe[0]);
      accRe[1] = MFMA16(a1, bx, accRe[1]);
    }
  }
  __syncthreads();                                           // R1 becomes the reduction array: red[wave][PLD]
  {
    float* red = R1;                                         // 4 x 1056 floats = 16.5 KiB <= 24 KiB
    for (int k = tid; k < 4 * PLD; k += T2) red[k] = 0.f;
    __syncthreads();
    float* mine = red + wave * PLD;
    // D[i = slot 4 (lane >> 4) + r (+16)][j = lane & 15]: j < Cin -> dW[ch][j], j == Cin -> db[ch]; the embedding pass: dW[ch][Cin + j]
    const int j = lane & 15;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int sl = 16 * ct + 4 * (lane >> 4) + r;
        const int ch = sl < CP ? chan_of(sl) : -1;
        if (ch >= 0) {
          if (j < CIN) {
            mine[ch * C + j] = accL[ct][r];
            mine[C * C + C + ch * C + j] = accR[ct][r];
          } else if (j == CIN) {
            mine[C * C + ch] = accL[ct][r];
            mine[2 * C * C + C + ch] = accR[ct][r];
          }
          if (j < Demb) {
            mine[ch * C + CIN + j] = accLe[ct][r];
            mine[C * C + C + ch * C + CIN + j] = accRe[ct][r];
          }
        }
      }
    // d att: waves 0, 1 hold head 0, waves 2, 3 head 1
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float s = wave_sum(datt[c]);
      if (lane == 0) mine[2 * C * C + 2 * C + hh * CH + c] = s;
    }
    __syncthreads();
    // the temb parts join wave 0's row (their entries are written by nobody else there ... they ARE: add, in thread order)
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int q = tid + 256 * jj;
      if (q < 2 * C * Demb) {
        const int mm = q / (C * Demb), r = q - mm * C * Demb, ch = r / Demb, e = r - ch * Demb;
        red[3 * PLD + mm * (C * C + C) + ch * C + CIN + e] += tw[jj];   // row 3: one writer per entry here
      }
    }
    __syncthreads();
    float* out = gr.partials + (int64_t)blockIdx.x * gr.partial_ld;
    for (int k = tid; k < PLD; k += T2) out[k] = (red[k] + red[PLD + k]) + (red[2 * PLD + k] + red[3 * PLD + k]);
  }
}

}  // namespace

// Rows of `partials` (= blocks) tecm_spatial_bwd2 launches for `d`, or 0 when this formulation does not serve the call.
extern "C" int tecm_spatial_bwd2_blocks(const TecmSpatial* dp) {
  if (dp == nullptr || check_common("tecm_spatial_bwd2_blocks", *dp) != TECM_OK) return 0;
  const TecmSpatial& d = *dp;
  if (!v2_eligible(d)) return 0;
  const Map2 m = make_map2(d.tile_edges_max);
  if ((size_t)m.total * sizeof(float) > 80 * 1024) return 0;   // two blocks per CU; a tile with more edges: first formulation
  const int G = d.B * d.L;
  int nch = (SPB2_OCC * 256) / d.num_tiles;                   // one round of SPB2_OCC blocks per CU
  if (const char* e = std::getenv("TECM_SPB2_NCH")) {         // diagnostics: chunks of graphs per tile
    const int v = atoi(e);
    if (v > 0) nch = v;
  }
  if (nch < 1) nch = 1;
  if (nch > G) nch = G;
  const int gc = (G + nch - 1) / nch;
  nch = (G + gc - 1) / gc;
  return d.num_tiles * nch;
}

extern "C" int tecm_spatial_bwd2(const TecmSpatial* dp, const TecmSpatialGrads* gp, float* ws, void* stream) {
  TECM_REQUIRE(dp != nullptr && gp != nullptr && ws != nullptr, TECM_E_ARG, "tecm_spatial_bwd2: null descriptor / workspace");
  const TecmSpatial& d = *dp;
  const TecmSpatialGrads& g = *gp;
  const int rc = check_common("tecm_spatial_bwd2", d);
  if (rc) return rc;
  const int nblk = tecm_spatial_bwd2_blocks(dp);
  TECM_REQUIRE(nblk > 0, TECM_E_ARG, "tecm_spatial_bwd2: not served (tecm_spatial_bwd2_blocks returned 0)");
  TECM_REQUIRE(g.dout && g.partials && g.d_node_tab && g.d_tod_tab && g.d_doy_tab && g.d_year_tab && g.d_season_tab &&
                   g.src_ptr && g.src_col && g.src_ptr_off,
               TECM_E_ARG, "tecm_spatial_bwd2: null pointer");
  TECM_REQUIRE(g.num_blocks == nblk && g.partial_ld >= PLD, TECM_E_ARG,
               "tecm_spatial_bwd2: num_blocks must be %d = tecm_spatial_bwd2_blocks(), partial_ld >= %d", nblk, PLD);
  TECM_REQUIRE(tecm_aligned(g.dout, 16) && tecm_aligned(d.x, 8) && tecm_aligned(ws, 16), TECM_E_ALIGN,
               "tecm_spatial_bwd2: dout and the workspace must be 16-byte aligned, x 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(spatial_prep_kernel, dim3((unsigned)((prep_threads(d) + 255) / 256)), dim3(256), 0, st, d, ws);
  TECM_CHECK_LAUNCH("tecm_spatial_bwd2(prep)");
  Bwd2Args a;
  a.d = d;
  a.g = g;
  a.ws = ws;
  const int G = d.B * d.L;
  a.nch = nblk / d.num_tiles;
  a.gc = (G + a.nch - 1) / a.nch;
  const size_t lds = (size_t)make_map2(d.tile_edges_max).total * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_bwd2_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_bwd2_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set = true;
  }
  if (d.Cin == 10) hipLaunchKernelGGL(spatial_bwd2_kernel<10>, dim3((unsigned)nblk), dim3(T2), lds, st, a);
  else hipLaunchKernelGGL(spatial_bwd2_kernel<6>, dim3((unsigned)nblk), dim3(T2), lds, st, a);
  TECM_CHECK_LAUNCH("tecm_spatial_bwd2");
  return TECM_OK;
}
