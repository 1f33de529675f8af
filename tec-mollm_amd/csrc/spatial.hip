// Fused stage a-1..a-3: SpatioTemporalEmbedding (modules.py:230-266) + GATv2Conv (modules.py:329-336,
// :356; torch_geometric semantics restated in oracle/ref_cpu.py:gatv2_conv) + residual (tec_mollm.py:94).
//
// HBM-bound (about 15 flop/byte): the only full-size traffic is one coalesced read of the
// (B,L,N,Cin) input slab and one coalesced write of the (B,L,N,Cp) output slab -- the reference's
// 4 gathers + 4 adds + cat + 2 permute copies + PyG's per-edge tensors never exist.
//
// A block owns one graph (b,t) and a tile of `tile_nodes` target nodes.  The host computes, from the
// CSR-by-target of the graph, the window [lo,hi) of node ids that covers the tile and all of its
// sources; x_l = lin_l(h) for the whole window is staged in LDS (neighbour features), lin_l / lin_r
// live in LDS as 24-float rows read with broadcast ds_read_b128, the attention vector sits in
// registers, the per-node logits / online softmax live in registers, and the output tile goes back
// through LDS so the global store is contiguous float4.  Graphs with g = t*B + b >= graphs_with_edges
// see only their self loop (the reference's literal behaviour for everything but graph 0).
//
// Backward recomputes the forward from x (no activations are saved) and reduces straight to parameter
// gradients; x needs no gradient.  By linearity every edge's contribution to d x_l[j] is folded into
// LDS accumulators of the block that owns the *target*, so no cross-block scatter of dx_l is needed.
// The 22x22 weight gradients are outer products sum_rows dxl^T [h,1]: they run on the f32 matrix cores
// (one 32x32 accumulator per wave, kept in registers across the block's timesteps); only the (N, Demb)
// node table sees float atomics (contiguous 256 B per wave instruction).
#include "spatial_common.h"

#ifndef SP_SKIP
#define SP_SKIP 0      // experiments only: bit0 skip dense x_l/x_r, bit1 edge phase, bit2 outer products, bit3 embedding grads
#endif

using namespace tecm_spatial;

namespace {

constexpr int DXP = 25;    // row pitch of the d x_l accumulators: odd, so the per-edge LDS float atomics of a wave (same
                           // column, 64 different rows) hit 64 different banks instead of 8 (measured: the edge phase
                           // of the backward is bound by ds_add_f32 under bank conflicts)

// h = cat([x, node_emb + temporal_emb])  (modules.py:261-264)
__device__ __forceinline__ void build_h(const TecmSpatial& d, const float* xrow, int node, const float* temb_lds,
                                        int b, int t, float (&h)[C]) {
  TimeIdx ti;
  if (!temb_lds) ti = load_time_idx(d, b, t, node);
#pragma unroll
  for (int k = 0; k < C; ++k) {
    if (k < d.Cin) {
      h[k] = xrow[k];
    } else {
      const int e = k - d.Cin;
      const float te = temb_lds ? temb_lds[e] : temporal_emb(d, ti, e);
      h[k] = d.node_tab[(int64_t)node * d.Demb + e] + te;
    }
  }
}

// ------------------------------------------------------------------------------------------ dense math
// All 22x22 products run on the f32 matrix cores (v_mfma_f32_32x32x2_f32) over LDS-resident operands:
//   rows x [h | 1 | 0] (24 cols)  times  a 24 x 32 matrix stored k-major  ->  32-row x 32-col accumulator.
// The constant-1 column 22 of `h` folds the bias in (row 22 of the k-major matrix holds it).
// A-operand reads have an 8-way bank conflict (row pitch 24 words) -- 16 LDS cycles against a 64-cycle MFMA.
constexpr int KM_FLOATS = CP * 32;          // one k-major 24 x 32 matrix

struct MatsLds {
  float* WlT;     // [k][c]: Wl[c][k], row 22 = bl              (x_l = [h|1] . WlT)
  float* WrT;     // [k][c]: Wr[c][k], row 22 = br
  float* WlE;     // [a][e]: Wl[a][Cin+e]                        (d emb += dxl . WlE)      backward only
  float* WrE;     // [a][e]: Wr[a][Cin+e]
  float* Sel;     // [k][e]: k == Cin+e                          (d emb += dout . Sel)
};

__device__ __forceinline__ void stage_mats(const TecmSpatial& d, float* base, MatsLds& m, bool bwd) {
  m.WlT = base;
  m.WrT = base + KM_FLOATS;
  m.WlE = base + 2 * KM_FLOATS;
  m.WrE = base + 3 * KM_FLOATS;
  m.Sel = base + 4 * KM_FLOATS;
  for (int i = threadIdx.x; i < KM_FLOATS; i += blockDim.x) {
    const int k = i >> 5, c = i & 31;
    float wl = 0.f, wr = 0.f;
    if (c < C) {
      if (k < C) { wl = d.Wl[c * C + k]; wr = d.Wr[c * C + k]; }
      else if (k == C) { wl = d.bl[c]; wr = d.br[c]; }
    }
    m.WlT[i] = wl;
    m.WrT[i] = wr;
    if (bwd) {
      const bool ok = k < C && c < d.Demb;
      m.WlE[i] = ok ? d.Wl[k * C + d.Cin + c] : 0.f;
      m.WrE[i] = ok ? d.Wr[k * C + d.Cin + c] : 0.f;
      m.Sel[i] = (c < d.Demb && k == d.Cin + c) ? 1.f : 0.f;
    }
  }
}

// acc(32 rows x 32 cols) += A[row0 .. row0+31][0..23] . KM   (rows clamped to [0, nrows): duplicates are discarded)
template <int PA = CP>
__device__ __forceinline__ void mfma_rows(f32x16& acc, const float* A, int row0, int nrows, const float* KM, int lane) {
  const int i = lane & 31, kq = lane >> 5;
  int row = row0 + i;
  row = row < nrows ? row : nrows - 1;
  const float* ar = A + row * PA + kq;
  const float* br = KM + kq * 32 + i;
#pragma unroll
  for (int s = 0; s < CP / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * s], br[2 * s * 32], acc, 0, 0, 0);
}

// OUT[rows][0..23] = A[rows][0..23] . KM for rows [beg, end) of an LDS array; the 4 waves take 32-row blocks in turn.
__device__ __forceinline__ void dense_rows(float* OUT, const float* A, int beg, int end, const float* KM, int wave,
                                           int lane) {
  const int j = lane & 31, kq = lane >> 5;
  for (int r0 = beg + 32 * wave; r0 < end; r0 += 128) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    mfma_rows(acc, A, r0, end, KM, lane);
    if (j < CP) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * kq;
        if (row < end) OUT[row * CP + j] = acc[e];
      }
    }
  }
}

__device__ __forceinline__ void load_row(const float* row, float (&v)[C]) {
  const float4* p = reinterpret_cast<const float4*>(row);
#pragma unroll
  for (int q = 0; q < CP / 4; ++q) {
    const float4 t = p[q];
    v[4 * q] = t.x;
    v[4 * q + 1] = t.y;
    if (4 * q + 2 < C) v[4 * q + 2] = t.z;
    if (4 * q + 3 < C) v[4 * q + 3] = t.w;
  }
}
__device__ __forceinline__ void store_row(float* row, const float (&v)[C], float c22, float c23) {
  float4* p = reinterpret_cast<float4*>(row);
#pragma unroll
  for (int q = 0; q < CP / 4; ++q) {
    float4 t;
    t.x = v[4 * q];
    t.y = v[4 * q + 1];
    t.z = 4 * q + 2 < C ? v[4 * q + 2] : c22;
    t.w = 4 * q + 3 < C ? v[4 * q + 3] : c23;
    p[q] = t;
  }
}

__device__ __forceinline__ void logits(const float (&xlj)[C], const float (&xr)[C], const float (&att)[C],
                                       float (&e)[H]) {
#pragma unroll
  for (int hh = 0; hh < H; ++hh) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) a = fmaf(att[hh * CH + c], lrelu(xlj[hh * CH + c] + xr[hh * CH + c]), a);
    e[hh] = a;
  }
}

// h rows of the window -> LDS ([h | 1 | 0]); a thread per row
__device__ __forceinline__ void build_window(const TecmSpatial& d, float* hw, int wa, int wb, int lo, int64_t grow,
                                             const float* temb_p, int b, int t) {
  for (int w = wa + threadIdx.x; w < wb; w += blockDim.x) {
    float h[C];
    build_h(d, d.x + (grow + lo + w) * d.Cin, lo + w, temb_p, b, t, h);
    store_row(hw + w * CP, h, 1.0f, 0.f);
  }
}

// ------------------------------------------------------------------------------------------ backward
// acc(32x32) += sum over `rows` of  A[row][0..23]^T  (x)  B[row][0..23]   on the f32 matrix core.
// Rows are dealt to the 4 waves in pairs (one MFMA consumes k = 2 rows).
template <int PA = CP>
__device__ __forceinline__ void outer_accumulate(f32x16& acc, const float* A, const float* Bm, int row_beg,
                                                 int row_end, int wave, int lane) {
  const int i = lane & 31, kq = lane >> 5;
  const bool col_ok = i < CP;
  for (int r0 = row_beg + 2 * wave; r0 < row_end; r0 += 8) {
    const int row = r0 + kq;
    const bool ok = col_ok && row < row_end;
    const float a = ok ? A[row * PA + i] : 0.f;
    const float bv = ok ? Bm[row * CP + i] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void spatial_bwd_kernel(const TecmSpatial d, const TecmSpatialGrads gr,
                                                          int nchunks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x % d.num_tiles;
  const int rest = blockIdx.x / d.num_tiles;
  const int tch = rest % nchunks, b = rest / nchunks;
  const int n0 = tile * d.tile_nodes;
  const int n1 = min(d.N, n0 + d.tile_nodes);
  const int lo = d.tile_lo[tile], hi = d.tile_hi[tile];
  const int W = hi - lo;
  const int wm4 = (d.win_max + 3) & ~3;
  const int Demb = d.Demb, Cin = d.Cin;
  float* hw = smem;                        // [wm4][CP]   [h | 1 | 0]
  float* xlw = hw + wm4 * CP;              // [wm4][CP]
  float* dxlw = xlw + wm4 * CP;            // [wm4][DXP]
  float* dxr = dxlw + wm4 * DXP;           // [tile_nodes][CP]  x_r of the tile first, then d x_r   (wm4 % 4 == 0: aligned)
  float* gt = dxr + d.tile_nodes * CP;     // [tile_nodes][CP]  dout rows of the tile
  float* dnode = gt + d.tile_nodes * CP;   // [wm4][Demb]
  float* temb = dnode + wm4 * Demb;        // [32]
  float* tsum = temb + 32;                 // [32]
  float* vec = tsum + 32;                  // [CP]      datt block reduction
  MatsLds mats;
  stage_mats(d, vec + CP, mats, true);
  int* eptr = reinterpret_cast<int*>(vec + CP + 5 * KM_FLOATS);    // [tile_nodes + 1]  CSR slice of the tile
  int* ecol = eptr + d.tile_nodes + 1;                             // [tile_edges_max]  window-relative sources
  int* sptr = ecol + d.tile_edges_max;                             // [wm4 + 1]  the tile's edges grouped by SOURCE row
  int* scol = sptr + wm4 + 1;                                      // [tile_edges_max]  (tile target << 16) | slot
  float* tstat = reinterpret_cast<float*>(scol + d.tile_edges_max);   // [tile_nodes][H][3]  softmax m, 1/z, dot
  float* attl = tstat + d.tile_nodes * H * 3;                      // [CP]  attention vector (both heads)
  {
    const int ebase = d.rowptr[n0];
    for (int q = tid; q <= n1 - n0; q += 256) eptr[q] = d.rowptr[n0 + q] - ebase;
    const int ne = d.rowptr[n1] - ebase;
    for (int q = tid; q < ne; q += 256) {
      ecol[q] = d.colidx[ebase + q] - lo;
      scol[q] = gr.src_col[ebase + q];
    }
    const int pbase = gr.src_ptr_off[tile];
    for (int q = tid; q <= W; q += 256) sptr[q] = gr.src_ptr[pbase + q];
    if (tid < C) attl[tid] = d.att[tid];
  }
  const bool tf_uniform = d.tf_sn == 0;

  for (int q = tid; q < W * Demb; q += 256) dnode[q] = 0.f;
  if (tid < CP) vec[tid] = 0.f;
  float att[CH], datt_acc[CH];                      // this thread's head (tid >> 7) only
#pragma unroll
  for (int c = 0; c < CH; ++c) { att[c] = d.att[(tid >> 7) * CH + c]; datt_acc[c] = 0.f; }
  f32x16 accL, accR;
#pragma unroll
  for (int e = 0; e < 16; ++e) { accL[e] = 0.f; accR[e] = 0.f; }

  const uint32_t dth = d.alpha_drop.p > 0.f ? tecm_drop_thresh(d.alpha_drop.p) : 0u;
  const float dinv = d.alpha_drop.p > 0.f ? 1.0f / (1.0f - d.alpha_drop.p) : 1.0f;
  const int tbeg = tch * gr.t_chunk;
  const int tend = min(d.L, tbeg + gr.t_chunk);

  for (int t = tbeg; t < tend; ++t) {
    const bool use_edges = (t * d.B + b) < d.graphs_with_edges;
    const int wa = use_edges ? 0 : n0 - lo;
    const int wb = use_edges ? W : n1 - lo;
    const int ta = n0 - lo, tb = n1 - lo;             // tile rows inside the window
    const int64_t grow = ((int64_t)b * d.L + t) * d.N;
    __syncthreads();                                  // previous timestep fully consumed
    TimeIdx tiu;
    if (tf_uniform) tiu = load_time_idx(d, b, t, 0);
    if (tf_uniform && tid < Demb) temb[tid] = temporal_emb(d, tiu, tid);
    if (tid < 32) tsum[tid] = 0.f;
    __syncthreads();

    // ---- A: h for the window, then x_l (window) and x_r (tile) on the matrix cores; clear d x_l
    build_window(d, hw, wa, wb, lo, grow, tf_uniform ? temb : nullptr, b, t);
    __syncthreads();
    if (!(SP_SKIP & 1)) {
      dense_rows(xlw, hw, wa, wb, mats.WlT, wave, lane);
      dense_rows(dxr - ta * CP, hw, ta, tb, mats.WrT, wave, lane);
    }
    __syncthreads();

    // ---- B: per (target node, head), two passes over the node's edges.  Threads 0..127 take head 0 of tile node
    //         tid, threads 128..255 head 1 of node tid-128: the head is wave-uniform, all four waves work, and the
    //         two threads of a node only ever touch their own 11 columns of the shared rows.
    const int tn = tid & 127, hh = tid >> 7;
    const int i = n0 + tn;
    const bool tgt = tn < d.tile_nodes && i < n1;
    float dxr_acc[CH];
    if (tgt) {
      const int wi = i - lo;
      const int c0 = hh * CH;
      float xr[CH], g[CH];
      load_head(dxr + tn * CP, hh, xr);
      load_head(gr.dout + (grow + i) * CP, hh, g);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        gt[tn * CP + c0 + c] = g[c];
        dxr_acc[c] = 0.f;
      }
      if (hh == 1) { gt[tn * CP + C] = 0.f; gt[tn * CP + C + 1] = 0.f; }
      const int64_t rowi = (int64_t)(t * d.B + b) * d.N + i;
      const uint64_t dbase = (uint64_t)((rowi * H + hh) * d.alpha_drop.ld);
      const int e0 = use_edges ? eptr[tn] : 0;
      const int deg = use_edges ? eptr[tn + 1] - e0 : 0;
      // Both passes walk the edge list two edges at a time (even / odd positions): the two chains are
      // independent, so their LDS round trips and exponentials overlap -- with one wave per SIMD there is no other
      // wave to hide that latency behind.  Slot s == deg is the implicit self loop.
      auto edge_terms = [&](int s_, float (&xlj)[CH], float& e, float& da) {
        const int j = s_ >= deg ? wi : ecol[e0 + s_];
        load_head(xlw + j * CP, hh, xlj);
        e = 0.f;
        da = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          e = fmaf(att[c], lrelu(xlj[c] + xr[c]), e);
          da = fmaf(g[c], xlj[c], da);
        }
        return j;
      };
      // pass 1: online softmax statistics and the numerator of dot = sum_j alpha_ij dalpha_ij
      float m2[2] = {-INFINITY, -INFINITY}, z2[2] = {0.f, 0.f}, num2[2] = {0.f, 0.f};
      for (int s = 0; s <= deg; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (s + u <= deg) {
            float xlj[CH], e, da;
            edge_terms(s + u, xlj, e, da);
            if (dth) da *= tecm_drop_mult(d.alpha_drop.seed, dbase + s + u, dth, dinv);
            const float mn = fmaxf(m2[u], e);
            const float corr = __expf(m2[u] - mn), pw = __expf(e - mn);
            z2[u] = z2[u] * corr + pw;
            num2[u] = num2[u] * corr + pw * da;
            m2[u] = mn;
          }
        }
      }
      const float m = fmaxf(m2[0], m2[1]);                 // chain 0 always holds at least slot 0
      const float k0 = __expf(m2[0] - m), k1 = __expf(m2[1] - m);   // exp(-inf) = 0 for an empty chain 1
      const float z = z2[0] * k0 + z2[1] * k1;
      const float num = num2[0] * k0 + num2[1] * k1;
      const float zinv = 1.0f / (z + 1e-16f);
      const float dot = num * zinv;
      tstat[(tn * H + hh) * 3 + 0] = m;                    // the by-source pass recomputes alpha from these
      tstat[(tn * H + hh) * 3 + 1] = zinv;
      tstat[(tn * H + hh) * 3 + 2] = dot;
      // pass 2: gradients that accumulate per TARGET (d x_r, d att); d x_l is gathered per source below
      for (int s = 0; s <= deg; s += 2) {
        float xl2[2][CH], e_[2], da_[2];
        int j2[2];
        bool on[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          on[u] = s + u <= deg;
          j2[u] = edge_terms(on[u] ? s + u : deg, xl2[u], e_[u], da_[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (on[u]) {
            const float alpha = __expf(e_[u] - m) * zinv;
            float mult = 1.0f;
            if (dth) mult = tecm_drop_mult(d.alpha_drop.seed, dbase + s + u, dth, dinv);
            const float de = alpha * (da_[u] * mult - dot);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const float sv = xl2[u][c] + xr[c];
              const float ds = de * att[c] * (sv > 0.f ? 1.0f : NEG_SLOPE);
              datt_acc[c] += de * lrelu(sv);
              dxr_acc[c] += ds;
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- B': d x_l gathered per SOURCE row of the window (no LDS float atomics: ds_add_f32 runs at about one
    //          lane per 3.5 cycles on this part and was 40 % of the kernel).  Item = (window row w, head); it walks
    //          the tile's edges that leave w (host-built by-source lists) plus w's own self loop when w is a tile row,
    //          recomputes alpha / d e from the per-target statistics and sums into registers.
    {
      const int Wn = wb - wa;
      for (int it = tid; it < 2 * Wn; it += 256) {
        const int h2 = it >= Wn ? 1 : 0;
        const int w = wa + it - h2 * Wn;
        const int c0 = h2 * CH;
        float xl[CH], acc[CH], at[CH];
        load_head(xlw + w * CP, h2, xl);
#pragma unroll
        for (int c = 0; c < CH; ++c) { acc[c] = 0.f; at[c] = attl[c0 + c]; }
        const bool self = w >= ta && w < tb;
        const int q0 = use_edges ? sptr[w] : 0;
        const int q1 = use_edges ? sptr[w + 1] : 0;
        for (int q = q0; q < q1 + (self ? 1 : 0); ++q) {
          int tt, sl;
          if (q < q1) {
            const int code = scol[q];
            tt = code >> 16;
            sl = code & 0xffff;
          } else {                                          // the implicit self loop: last slot of its target
            tt = w - ta;
            sl = use_edges ? eptr[tt + 1] - eptr[tt] : 0;
          }
          float xr[CH], g[CH];
          load_head(dxr + tt * CP, h2, xr);                 // still x_r: d x_r is written after this pass
          load_head(gt + tt * CP, h2, g);
          float e = 0.f, da = 0.f;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            e = fmaf(at[c], lrelu(xl[c] + xr[c]), e);
            da = fmaf(g[c], xl[c], da);
          }
          const float* st = tstat + (tt * H + h2) * 3;
          const float alpha = __expf(e - st[0]) * st[1];
          float mult = 1.0f;
          if (dth) {
            const int64_t rowt = (int64_t)(t * d.B + b) * d.N + n0 + tt;
            mult = tecm_drop_mult(d.alpha_drop.seed, (uint64_t)((rowt * H + h2) * d.alpha_drop.ld) + sl, dth, dinv);
          }
          const float de = alpha * (da * mult - st[2]);
          const float am = alpha * mult;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const float sv = xl[c] + xr[c];
            acc[c] += am * g[c] + de * at[c] * (sv > 0.f ? 1.0f : NEG_SLOPE);
          }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) dxlw[w * DXP + c0 + c] = acc[c];
        if (h2 == 1) { dxlw[w * DXP + C] = 0.f; dxlw[w * DXP + C + 1] = 0.f; dxlw[w * DXP + C + 2] = 0.f; }
      }
    }
    __syncthreads();
    if (tgt) {
      const int c0 = hh * CH;
#pragma unroll
      for (int c = 0; c < CH; ++c) dxr[tn * CP + c0 + c] = dxr_acc[c];
      if (hh == 1) { dxr[tn * CP + C] = 0.f; dxr[tn * CP + C + 1] = 0.f; }
    }
    __syncthreads();


    // ---- C1: [dWl | dbl] += dxl^T [h, 1] over the window, [dWr | dbr] += dxr^T [h, 1] over the tile
    if (!(SP_SKIP & 4)) {
      outer_accumulate<DXP>(accL, dxlw, hw, wa, wb, wave, lane);
      outer_accumulate(accR, dxr - ta * CP, hw, ta, tb, wave, lane);
    }

    // ---- C2: embedding part of dh = dxl.WlE (+ tile rows: dxr.WrE + dout.Sel) -> node-table accumulators and
    //          the temporal tables, all three products chained into one MFMA accumulator per 32-row block
    if (!(SP_SKIP & 8)) {
      const int j = lane & 31, kq = lane >> 5;
      float colsum = 0.f;
      for (int r0 = wa + 32 * wave; r0 < wb; r0 += 128) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        mfma_rows<DXP>(acc, dxlw, r0, wb, mats.WlE, lane);
        const bool overlaps = r0 + 32 > ta && r0 < tb;                       // wave-uniform
        if (overlaps) {
          // tile operands are indexed relative to the tile; rows of this block outside the tile contribute 0
          const int i = lane & 31;
          const int row = r0 + i;
          const bool in = row >= ta && row < tb;
          const float* ar = dxr + (in ? row - ta : 0) * CP + kq;
          const float* gg = gt + (in ? row - ta : 0) * CP + kq;
          const float* b1 = mats.WrE + kq * 32 + i;
          const float* b2 = mats.Sel + kq * 32 + i;
#pragma unroll
          for (int s = 0; s < CP / 2; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(in ? ar[2 * s] : 0.f, b1[2 * s * 32], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(in ? gg[2 * s] : 0.f, b2[2 * s * 32], acc, 0, 0, 0);
          }
        }
        if (j < Demb) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * kq;
            if (row < wb) {
              const float v = acc[e];
              dnode[row * Demb + j] += v;                                     // (row, j) has exactly one owner
              if (tf_uniform) {
                colsum += v;
              } else {
                const TimeIdx ti = load_time_idx(d, b, t, lo + row);
                atomicAdd(&gr.d_tod_tab[ti.tod * Demb + j], v);
                atomicAdd(&gr.d_doy_tab[ti.doy * Demb + j], v);
                atomicAdd(&gr.d_year_tab[ti.year * Demb + j], v);
                atomicAdd(&gr.d_season_tab[ti.season * Demb + j], v);
              }
            }
          }
        }
      }
      if (tf_uniform && j < Demb) atomicAdd(&tsum[j], colsum);
    }
    __syncthreads();
    if (tf_uniform && tid < Demb) {
      const float v = tsum[tid];
      atomicAdd(&gr.d_tod_tab[tiu.tod * Demb + tid], v);
      atomicAdd(&gr.d_doy_tab[tiu.doy * Demb + tid], v);
      atomicAdd(&gr.d_year_tab[tiu.year * Demb + tid], v);
      atomicAdd(&gr.d_season_tab[tiu.season * Demb + tid], v);
    }
  }

  // ---- block results.  partial row layout: dWl (C*C) | dbl (C) | dWr (C*C) | dbr (C) | datt (C) | (C unused)
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const float a = wave_sum(datt_acc[c]);
    if (lane == 0) atomicAdd(&vec[(tid >> 7) * CH + c], a);
  }
  __syncthreads();
  float* part = gr.partials + (int64_t)blockIdx.x * gr.partial_ld;
  if (tid < C) part[2 * C * C + 2 * C + tid] = vec[tid];
  for (int q = tid; q < W * Demb; q += 256) atomicAdd(&gr.d_node_tab[(int64_t)lo * Demb + q], dnode[q]);
  __syncthreads();                                     // every LDS array is dead from here on
  // reduce the four waves' MFMA accumulators through LDS: [2][4][32][32] floats (host guarantees the size)
  float* red = smem;
  {
    const int col = lane & 31, hq = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * hq;
      red[((0 * 4 + wave) * 32 + row) * 32 + col] = accL[e];
      red[((1 * 4 + wave) * 32 + row) * 32 + col] = accR[e];
    }
  }
  __syncthreads();
  for (int o = tid; o < 2 * C * (C + 1); o += 256) {
    const int mtx = o / (C * (C + 1));
    const int rem = o - mtx * C * (C + 1);
    const int a = rem / (C + 1), k = rem - a * (C + 1);
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += red[((mtx * 4 + w) * 32 + a) * 32 + k];
    const int base = mtx * (C * C + C);
    if (k < C)
      part[base + a * C + k] = s;
    else
      part[base + C * C + a] = s;
  }
}

}  // namespace

extern "C" int tecm_spatial_bwd(const TecmSpatial* dp, const TecmSpatialGrads* gp, void* stream) {
  TECM_REQUIRE(dp != nullptr && gp != nullptr, TECM_E_ARG, "tecm_spatial_bwd: null descriptor");
  const TecmSpatial& d = *dp;
  const TecmSpatialGrads& g = *gp;
  const int rc = check_common("tecm_spatial_bwd", d);
  if (rc) return rc;
  TECM_REQUIRE(g.dout && g.d_node_tab && g.d_tod_tab && g.d_doy_tab && g.d_year_tab && g.d_season_tab && g.partials,
               TECM_E_ARG, "tecm_spatial_bwd: null pointer");
  TECM_REQUIRE(tecm_aligned(g.dout, 16), TECM_E_ALIGN, "tecm_spatial_bwd: dout must be 16-byte aligned");
  TECM_REQUIRE(g.t_chunk > 0, TECM_E_ARG, "tecm_spatial_bwd: t_chunk must be positive");
  TECM_REQUIRE(g.src_ptr && g.src_col && g.src_ptr_off, TECM_E_ARG, "tecm_spatial_bwd: by-source edge lists missing");
  TECM_REQUIRE(d.tile_nodes <= 128, TECM_E_ARG, "tecm_spatial_bwd: tile_nodes must be <= 128 (two threads per node)");
  const int nchunks = (d.L + g.t_chunk - 1) / g.t_chunk;
  const int nblocks = d.num_tiles * d.B * nchunks;
  TECM_REQUIRE(g.num_blocks == nblocks, TECM_E_ARG, "tecm_spatial_bwd: num_blocks must be %d (got %d)", nblocks,
               g.num_blocks);
  TECM_REQUIRE(g.partial_ld >= 2 * C * C + 4 * C, TECM_E_ARG, "tecm_spatial_bwd: partial_ld must be >= %d",
               2 * C * C + 4 * C);
  const int wm4 = (d.win_max + 3) & ~3;
  size_t floats = (size_t)2 * wm4 * CP + (size_t)wm4 * DXP + (size_t)2 * d.tile_nodes * CP + (size_t)wm4 * d.Demb + 64 + CP + 5 * KM_FLOATS +
                  d.tile_nodes + 1 + 2 * (size_t)d.tile_edges_max + wm4 + 1 + (size_t)d.tile_nodes * H * 3 + CP;
  if (floats < 8192 + 64) floats = 8192 + 64;          // the final 4-wave MFMA reduction needs 2*4*32*32 floats
  const size_t lds = sizeof(float) * floats;
  TECM_REQUIRE(lds <= (size_t)kLdsBudget, TECM_E_LDS,
               "tecm_spatial_bwd: neighbour window of %d rows needs %zu B of LDS (> 160 KiB); renumber the graph "
               "(e.g. RCM) or shrink tile_nodes", d.win_max, lds);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        kLdsBudget);
    attr_set = true;
  }
  hipLaunchKernelGGL(spatial_bwd_kernel, dim3(nblocks), dim3(256), lds, (hipStream_t)stream, d, g, nchunks);
  TECM_CHECK_LAUNCH("tecm_spatial_bwd");
  return TECM_OK;
}
