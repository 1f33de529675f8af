// Fused stage a-1..a-3: SpatioTemporalEmbedding (modules.py:230-266) + GATv2Conv (modules.py:329-336,
// :356; torch_geometric semantics restated in oracle/ref_cpu.py:gatv2_conv) + residual (tec_mollm.py:94).
//
// HBM-bound (about 15 flop/byte): the only full-size traffic is one coalesced read of the
// (B,L,N,Cin) input slab and one coalesced write of the (B,L,N,Cp) output slab -- the reference's
// 4 gathers + 4 adds + cat + 2 permute copies + PyG's per-edge tensors never exist.
//
// A block owns one graph (b,t) and a tile of `tile_nodes` target nodes.  The host computes, from the
// CSR-by-target of the graph, the window [lo,hi) of node ids that covers the tile and all of its
// sources; x_l = lin_l(h) for the whole window is staged in LDS (neighbour features), the per-node
// attention logits / online softmax live in registers, lin_l / lin_r weights arrive through the scalar
// cache (wave-uniform addresses -> s_load), and the output tile goes back through LDS so the global
// store is contiguous float4.  Graphs with g = t*B + b >= graphs_with_edges see only their self loop
// (the reference's literal behaviour for everything but graph 0).
//
// Backward recomputes the forward from x (no activations are saved) and reduces straight to parameter
// gradients; x needs no gradient.  By linearity every edge's contribution to d x_l[j] is folded into
// LDS accumulators of the block that owns the *target*, so no cross-block scatter of dx_l is needed:
// only the (N, Demb) node table sees float atomics (256-byte contiguous per wave instruction).
#include "common.h"

namespace {

constexpr float NEG_SLOPE = 0.2f;

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v >= hi ? hi - 1 : v); }

struct TimeIdx {
  int tod, doy, year, season;
};
__device__ __forceinline__ TimeIdx load_time_idx(const TecmSpatial& d, int b, int t, int node) {
  const float* p = d.tf + (int64_t)b * d.tf_sb + (int64_t)t * d.tf_sl + (int64_t)node * d.tf_sn;
  TimeIdx ti;
  ti.tod = clampi((int)p[0], 12);                       // .long() truncation, modules.py:250-253
  ti.doy = clampi((int)p[d.tf_sf], 366);
  ti.year = clampi((int)p[2 * d.tf_sf], d.year_rows);
  ti.season = clampi((int)p[3 * d.tf_sf], 4);
  return ti;
}
// ((tod + doy) + year) + season -- the exact association of modules.py:260
__device__ __forceinline__ float temporal_emb(const TecmSpatial& d, const TimeIdx& ti, int k) {
  const int D = d.Demb;
  return ((d.tod_tab[ti.tod * D + k] + d.doy_tab[ti.doy * D + k]) + d.year_tab[ti.year * D + k]) +
         d.season_tab[ti.season * D + k];
}

// h = cat([x, node_emb + temporal_emb])  (modules.py:261-264)
template <int C>
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

// out = W h + b with W (C,C) row-major: wave-uniform addresses, so the compiler feeds the FMAs from SGPRs
template <int C>
__device__ __forceinline__ void dense(const float* __restrict__ W, const float* __restrict__ bias,
                                      const float (&h)[C], float (&out)[C]) {
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float a = bias[c];
#pragma unroll
    for (int k = 0; k < C; ++k) a = fmaf(W[c * C + k], h[k], a);
    out[c] = a;
  }
}

__device__ __forceinline__ float lrelu(float s) { return s > 0.f ? s : NEG_SLOPE * s; }

template <int C, int H>
__device__ __forceinline__ void logits(const float* xlj, const float (&xr)[C], const float* __restrict__ att,
                                       float (&e)[H]) {
  constexpr int CH = C / H;
#pragma unroll
  for (int hh = 0; hh < H; ++hh) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) a = fmaf(att[hh * CH + c], lrelu(xlj[hh * CH + c] + xr[hh * CH + c]), a);
    e[hh] = a;
  }
}

template <int C, int H>
__global__ __launch_bounds__(256) void spatial_fwd_kernel(const TecmSpatial d, int Cp) {
  constexpr int CH = C / H;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int b = blockIdx.y / d.L, t = blockIdx.y - b * d.L;
  const bool use_edges = (t * d.B + b) < d.graphs_with_edges;
  const int n0 = tile * d.tile_nodes;
  const int n1 = min(d.N, n0 + d.tile_nodes);
  const int lo = use_edges ? d.tile_lo[tile] : n0;
  const int hi = use_edges ? d.tile_hi[tile] : n1;
  const int W = hi - lo;
  const int wm4 = (d.win_max + 3) & ~3;
  float* xin = smem;
  float* xlw = xin + wm4 * d.Cin;
  float* outt = xlw + wm4 * C;
  float* temb = outt + d.tile_nodes * Cp;
  const bool tf_uniform = d.tf_sn == 0;

  const int64_t grow = ((int64_t)b * d.L + t) * d.N;          // first row of this graph
  {
    const float* src = d.x + (grow + lo) * d.Cin;
    for (int i = tid; i < W * d.Cin; i += 256) xin[i] = src[i];
  }
  if (tf_uniform && tid < d.Demb) {
    const TimeIdx ti = load_time_idx(d, b, t, 0);
    temb[tid] = temporal_emb(d, ti, tid);
  }
  __syncthreads();
  const float* temb_p = tf_uniform ? temb : nullptr;

  if (use_edges) {
    for (int w = tid; w < W; w += 256) {
      float h[C], xl[C];
      build_h<C>(d, xin + w * d.Cin, lo + w, temb_p, b, t, h);
      dense<C>(d.Wl, d.bl, h, xl);
#pragma unroll
      for (int c = 0; c < C; ++c) xlw[w * C + c] = xl[c];
    }
    __syncthreads();
  }

  const int i = n0 + tid;
  if (i < n1) {
    float h[C], xr[C], xls[C];
    build_h<C>(d, xin + (i - lo) * d.Cin, i, temb_p, b, t, h);
    dense<C>(d.Wr, d.br, h, xr);
    if (use_edges) {
#pragma unroll
      for (int c = 0; c < C; ++c) xls[c] = xlw[(i - lo) * C + c];
    } else {
      dense<C>(d.Wl, d.bl, h, xls);
    }
    float m[H], z[H], acc[C];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) { m[hh] = -INFINITY; z[hh] = 0.f; }
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    const uint32_t dth = d.alpha_drop.p > 0.f ? tecm_drop_thresh(d.alpha_drop.p) : 0u;
    const float dinv = d.alpha_drop.p > 0.f ? 1.0f / (1.0f - d.alpha_drop.p) : 1.0f;
    const int64_t rowi = (int64_t)(t * d.B + b) * d.N + i;     // row in the reference's (L*B*N) flattening
    const int e0 = use_edges ? d.rowptr[i] : 0;
    const int deg = use_edges ? d.rowptr[i + 1] - e0 : 0;
    for (int s = 0; s <= deg; ++s) {
      const bool self = s == deg;
      float xlj[C];
      if (self) {
#pragma unroll
        for (int c = 0; c < C; ++c) xlj[c] = xls[c];
      } else {
        const int j = d.colidx[e0 + s] - lo;
#pragma unroll
        for (int c = 0; c < C; ++c) xlj[c] = xlw[j * C + c];
      }
      float e[H];
      logits<C, H>(xlj, xr, d.att, e);
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        const float mn = fmaxf(m[hh], e[hh]);
        const float corr = expf(m[hh] - mn);     // exp(-inf) = 0 on the first edge
        const float p = expf(e[hh] - mn);
        float pm = p;
        if (dth) pm *= tecm_drop_mult(d.alpha_drop.seed, (uint64_t)((rowi * H + hh) * d.alpha_drop.ld + s), dth, dinv);
        z[hh] = z[hh] * corr + p;
        m[hh] = mn;
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[hh * CH + c] = acc[hh * CH + c] * corr + pm * xlj[hh * CH + c];
      }
    }
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const float inv = 1.0f / (z[hh] + 1e-16f);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k = hh * CH + c;
        outt[tid * Cp + k] = h[k] + (acc[k] * inv + d.bias[k]);
      }
    }
    for (int k = C; k < Cp; ++k) outt[tid * Cp + k] = 0.f;
  }
  __syncthreads();
  {
    const int nf4 = (n1 - n0) * Cp / 4;
    float4* dst = reinterpret_cast<float4*>(d.out + (grow + n0) * Cp);
    const float4* src = reinterpret_cast<const float4*>(outt);
    for (int q = tid; q < nf4; q += 256) dst[q] = src[q];
  }
}

template <int C, int H>
__global__ __launch_bounds__(256) void spatial_bwd_kernel(const TecmSpatial d, const TecmSpatialGrads gr, int Cp,
                                                          int nchunks) {
  constexpr int CH = C / H;
  constexpr int NOUT = C * (C + 1);             // (a, k) with k == C the bias column
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int tile = blockIdx.x % d.num_tiles;
  const int rest = blockIdx.x / d.num_tiles;
  const int tch = rest % nchunks, b = rest / nchunks;
  const int n0 = tile * d.tile_nodes;
  const int n1 = min(d.N, n0 + d.tile_nodes);
  const int lo = d.tile_lo[tile], hi = d.tile_hi[tile];
  const int W = hi - lo;
  const int wm4 = (d.win_max + 3) & ~3;
  const int Demb = d.Demb, Cin = d.Cin;
  float* hw = smem;                        // [wm4][C]
  float* xlw = hw + wm4 * C;               // [wm4][C]
  float* dxlw = xlw + wm4 * C;             // [wm4][C]
  float* dxr = dxlw + wm4 * C;             // [tile_nodes][C]
  float* dnode = dxr + d.tile_nodes * C;   // [wm4][Demb]
  float* temb = dnode + wm4 * Demb;        // [32]
  float* tsum = temb + 32;                 // [32]
  float* vec = tsum + 32;                  // [2*C]  datt | dbias block reduction
  const bool tf_uniform = d.tf_sn == 0;

  float accL[2] = {0.f, 0.f}, accR[2] = {0.f, 0.f};
  float datt_acc[C], dbias_acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { datt_acc[c] = 0.f; dbias_acc[c] = 0.f; }
  for (int q = tid; q < W * Demb; q += 256) dnode[q] = 0.f;
  if (tid < 2 * C) vec[tid] = 0.f;

  const uint32_t dth = d.alpha_drop.p > 0.f ? tecm_drop_thresh(d.alpha_drop.p) : 0u;
  const float dinv = d.alpha_drop.p > 0.f ? 1.0f / (1.0f - d.alpha_drop.p) : 1.0f;
  const int tbeg = tch * gr.t_chunk;
  const int tend = min(d.L, tbeg + gr.t_chunk);

  for (int t = tbeg; t < tend; ++t) {
    const bool use_edges = (t * d.B + b) < d.graphs_with_edges;
    const int wa = use_edges ? 0 : n0 - lo;
    const int wb = use_edges ? W : n1 - lo;
    const int64_t grow = ((int64_t)b * d.L + t) * d.N;
    __syncthreads();                                  // previous timestep fully consumed
    TimeIdx tiu;
    if (tf_uniform) tiu = load_time_idx(d, b, t, 0);
    if (tf_uniform && tid < Demb) temb[tid] = temporal_emb(d, tiu, tid);
    if (tid < 32) tsum[tid] = 0.f;
    __syncthreads();
    const float* temb_p = tf_uniform ? temb : nullptr;

    // ---- A: recompute h and x_l for the window, clear d x_l
    for (int w = wa + tid; w < wb; w += 256) {
      float h[C], xl[C];
      build_h<C>(d, d.x + (grow + lo + w) * Cin, lo + w, temb_p, b, t, h);
      dense<C>(d.Wl, d.bl, h, xl);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        hw[w * C + c] = h[c];
        xlw[w * C + c] = xl[c];
        dxlw[w * C + c] = 0.f;
      }
    }
    __syncthreads();

    // ---- B: per target node, edge phase
    const int i = n0 + tid;
    if (i < n1) {
      const int wi = i - lo;
      float h[C], xr[C], g[C], dxr_acc[C];
#pragma unroll
      for (int c = 0; c < C; ++c) { h[c] = hw[wi * C + c]; dxr_acc[c] = 0.f; }
      dense<C>(d.Wr, d.br, h, xr);
      const float* grow_p = gr.dout + (grow + i) * Cp;
#pragma unroll
      for (int c = 0; c < C; ++c) { g[c] = grow_p[c]; dbias_acc[c] += g[c]; }
      const int64_t rowi = (int64_t)(t * d.B + b) * d.N + i;
      const int e0 = use_edges ? d.rowptr[i] : 0;
      const int deg = use_edges ? d.rowptr[i + 1] - e0 : 0;
      float m[H], z[H], dot[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) { m[hh] = -INFINITY; z[hh] = 0.f; dot[hh] = 0.f; }
      // pass 1: softmax statistics
      for (int s = 0; s <= deg; ++s) {
        const int j = s == deg ? wi : d.colidx[e0 + s] - lo;
        float e[H];
        logits<C, H>(xlw + j * C, xr, d.att, e);
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
          const float mn = fmaxf(m[hh], e[hh]);
          z[hh] = z[hh] * expf(m[hh] - mn) + expf(e[hh] - mn);
          m[hh] = mn;
        }
      }
      float zinv[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) zinv[hh] = 1.0f / (z[hh] + 1e-16f);
      // pass 2: dot_h = sum_j alpha_ij * dalpha_ij
      for (int s = 0; s <= deg; ++s) {
        const int j = s == deg ? wi : d.colidx[e0 + s] - lo;
        float e[H];
        logits<C, H>(xlw + j * C, xr, d.att, e);
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
          const float alpha = expf(e[hh] - m[hh]) * zinv[hh];
          float mult = 1.0f;
          if (dth) mult = tecm_drop_mult(d.alpha_drop.seed, (uint64_t)((rowi * H + hh) * d.alpha_drop.ld + s), dth, dinv);
          float da = 0.f;
#pragma unroll
          for (int c = 0; c < CH; ++c) da = fmaf(g[hh * CH + c], xlw[j * C + hh * CH + c], da);
          dot[hh] += alpha * (da * mult);
        }
      }
      // pass 3: gradients
      for (int s = 0; s <= deg; ++s) {
        const int j = s == deg ? wi : d.colidx[e0 + s] - lo;
        float e[H];
        logits<C, H>(xlw + j * C, xr, d.att, e);
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
          const float alpha = expf(e[hh] - m[hh]) * zinv[hh];
          float mult = 1.0f;
          if (dth) mult = tecm_drop_mult(d.alpha_drop.seed, (uint64_t)((rowi * H + hh) * d.alpha_drop.ld + s), dth, dinv);
          float da = 0.f;
#pragma unroll
          for (int c = 0; c < CH; ++c) da = fmaf(g[hh * CH + c], xlw[j * C + hh * CH + c], da);
          const float de = alpha * (da * mult - dot[hh]);
          const float am = alpha * mult;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int k = hh * CH + c;
            const float sv = xlw[j * C + k] + xr[k];
            const float ds = de * d.att[k] * (sv > 0.f ? 1.0f : NEG_SLOPE);
            datt_acc[k] += de * lrelu(sv);
            dxr_acc[k] += ds;
            atomicAdd(&dxlw[j * C + k], am * g[k] + ds);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) dxr[tid * C + c] = dxr_acc[c];
    }
    __syncthreads();

    // ---- C1: dWl[a][k] += sum_w dxl[w][a] h[w][k], dWr[a][k] += sum_i dxr[i][a] h[i][k]; k == C -> bias
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int o = tid + sl * 256;
      if (o < NOUT) {
        const int a = o / (C + 1), k = o - a * (C + 1);
        float sL = 0.f, sR = 0.f;
        if (k < C) {
          for (int w = wa; w < wb; ++w) sL = fmaf(dxlw[w * C + a], hw[w * C + k], sL);
          for (int q = 0; q < n1 - n0; ++q) sR = fmaf(dxr[q * C + a], hw[(n0 - lo + q) * C + k], sR);
        } else {
          for (int w = wa; w < wb; ++w) sL += dxlw[w * C + a];
          for (int q = 0; q < n1 - n0; ++q) sR += dxr[q * C + a];
        }
        accL[sl] += sL;
        accR[sl] += sR;
      }
    }
    // ---- C2: embedding part of dh -> node table accumulators and temporal tables
    for (int w0 = wa; w0 < wb; w0 += 256) {
      const int w = w0 + tid;
      const bool act = w < wb;
      const int node = lo + w;
      const bool intile = act && node >= n0 && node < n1;
      TimeIdx ti;
      if (act && !tf_uniform) ti = load_time_idx(d, b, t, node);
      for (int e = 0; e < Demb; ++e) {
        float v = 0.f;
        if (act) {
          const int col = Cin + e;
#pragma unroll
          for (int a = 0; a < C; ++a) v = fmaf(d.Wl[a * C + col], dxlw[w * C + a], v);
          if (intile) {
            const int q = node - n0;
#pragma unroll
            for (int a = 0; a < C; ++a) v = fmaf(d.Wr[a * C + col], dxr[q * C + a], v);
            v += gr.dout[(grow + node) * Cp + col];
          }
          dnode[w * Demb + e] += v;
          if (!tf_uniform) {
            atomicAdd(&gr.d_tod_tab[ti.tod * Demb + e], v);
            atomicAdd(&gr.d_doy_tab[ti.doy * Demb + e], v);
            atomicAdd(&gr.d_year_tab[ti.year * Demb + e], v);
            atomicAdd(&gr.d_season_tab[ti.season * Demb + e], v);
          }
        }
        if (tf_uniform) {
          const float sv = wave_sum(v);
          if (lane == 0) atomicAdd(&tsum[e], sv);
        }
      }
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

  // ---- block results
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float a = wave_sum(datt_acc[c]);
    const float bsum = wave_sum(dbias_acc[c]);
    if (lane == 0) {
      atomicAdd(&vec[c], a);
      atomicAdd(&vec[C + c], bsum);
    }
  }
  __syncthreads();
  float* part = gr.partials + (int64_t)blockIdx.x * gr.partial_ld;
  // layout: dWl (C*C) | dbl (C) | dWr (C*C) | dbr (C) | datt (C) | dbias (C)
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int o = tid + sl * 256;
    if (o < NOUT) {
      const int a = o / (C + 1), k = o - a * (C + 1);
      if (k < C) {
        part[a * C + k] = accL[sl];
        part[C * C + C + a * C + k] = accR[sl];
      } else {
        part[C * C + a] = accL[sl];
        part[2 * C * C + C + a] = accR[sl];
      }
    }
  }
  if (tid < 2 * C) part[2 * C * C + 2 * C + tid] = vec[tid];
  for (int q = tid; q < W * Demb; q += 256) atomicAdd(&gr.d_node_tab[(int64_t)lo * Demb + q], dnode[q]);
}

int check_common(const char* who, const TecmSpatial& d) {
  TECM_REQUIRE(d.B > 0 && d.L > 0 && d.N > 0 && d.Cin > 0 && d.Demb > 0 && d.H > 0, TECM_E_ARG, "%s: bad shape", who);
  TECM_REQUIRE(d.Cin + d.Demb == 22 && d.H == 2, TECM_E_ARG,
               "%s: built for C = Cin + Demb = 22 channels and 2 heads (got C=%d H=%d)", who, d.Cin + d.Demb, d.H);
  TECM_REQUIRE(d.Demb <= 32, TECM_E_ARG, "%s: Demb must be <= 32", who);
  TECM_REQUIRE(d.x && d.tf && d.node_tab && d.tod_tab && d.doy_tab && d.year_tab && d.season_tab && d.Wl && d.bl &&
                   d.Wr && d.br && d.att && d.bias && d.rowptr && d.colidx && d.tile_lo && d.tile_hi,
               TECM_E_ARG, "%s: null pointer", who);
  TECM_REQUIRE(d.num_tiles > 0 && d.tile_nodes > 0 && d.tile_nodes <= 256 &&
                   (int64_t)d.num_tiles * d.tile_nodes >= d.N && d.win_max >= 1,
               TECM_E_ARG, "%s: bad node tiling", who);
  TECM_REQUIRE(d.year_rows > 0, TECM_E_ARG, "%s: year_rows must be positive", who);
  return TECM_OK;
}

constexpr int kLdsBudget = 160 * 1024;

}  // namespace

extern "C" int tecm_spatial_fwd(const TecmSpatial* dp, void* stream) {
  TECM_REQUIRE(dp != nullptr, TECM_E_ARG, "tecm_spatial_fwd: null descriptor");
  const TecmSpatial& d = *dp;
  const int rc = check_common("tecm_spatial_fwd", d);
  if (rc) return rc;
  TECM_REQUIRE(d.out != nullptr && tecm_aligned(d.out, 16), TECM_E_ALIGN, "tecm_spatial_fwd: out must be 16-byte aligned");
  constexpr int C = 22;
  const int Cp = (C + 3) & ~3;
  const int wm4 = (d.win_max + 3) & ~3;
  const size_t lds = sizeof(float) * ((size_t)wm4 * d.Cin + (size_t)wm4 * C + (size_t)d.tile_nodes * Cp + 32);
  TECM_REQUIRE(lds <= (size_t)kLdsBudget, TECM_E_LDS,
               "tecm_spatial_fwd: neighbour window of %d rows needs %zu B of LDS (> 160 KiB); renumber the graph "
               "(e.g. RCM) or shrink tile_nodes", d.win_max, lds);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_fwd_kernel<22, 2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
    attr_set = true;
  }
  hipLaunchKernelGGL((spatial_fwd_kernel<22, 2>), dim3(d.num_tiles, d.B * d.L), dim3(256), lds, (hipStream_t)stream, d,
                     Cp);
  TECM_CHECK_LAUNCH("tecm_spatial_fwd");
  return TECM_OK;
}

extern "C" int tecm_spatial_bwd(const TecmSpatial* dp, const TecmSpatialGrads* gp, void* stream) {
  TECM_REQUIRE(dp != nullptr && gp != nullptr, TECM_E_ARG, "tecm_spatial_bwd: null descriptor");
  const TecmSpatial& d = *dp;
  const TecmSpatialGrads& g = *gp;
  const int rc = check_common("tecm_spatial_bwd", d);
  if (rc) return rc;
  constexpr int C = 22;
  const int Cp = (C + 3) & ~3;
  TECM_REQUIRE(g.dout && g.d_node_tab && g.d_tod_tab && g.d_doy_tab && g.d_year_tab && g.d_season_tab && g.partials,
               TECM_E_ARG, "tecm_spatial_bwd: null pointer");
  TECM_REQUIRE(g.t_chunk > 0, TECM_E_ARG, "tecm_spatial_bwd: t_chunk must be positive");
  const int nchunks = (d.L + g.t_chunk - 1) / g.t_chunk;
  const int nblocks = d.num_tiles * d.B * nchunks;
  TECM_REQUIRE(g.num_blocks == nblocks, TECM_E_ARG, "tecm_spatial_bwd: num_blocks must be %d (got %d)", nblocks,
               g.num_blocks);
  TECM_REQUIRE(g.partial_ld >= 2 * C * C + 4 * C, TECM_E_ARG, "tecm_spatial_bwd: partial_ld must be >= %d",
               2 * C * C + 4 * C);
  const int wm4 = (d.win_max + 3) & ~3;
  const size_t lds =
      sizeof(float) * ((size_t)3 * wm4 * C + (size_t)d.tile_nodes * C + (size_t)wm4 * d.Demb + 64 + 2 * C);
  TECM_REQUIRE(lds <= (size_t)kLdsBudget, TECM_E_LDS,
               "tecm_spatial_bwd: neighbour window of %d rows needs %zu B of LDS (> 160 KiB); renumber the graph "
               "(e.g. RCM) or shrink tile_nodes", d.win_max, lds);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_bwd_kernel<22, 2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
    attr_set = true;
  }
  hipLaunchKernelGGL((spatial_bwd_kernel<22, 2>), dim3(nblocks), dim3(256), lds, (hipStream_t)stream, d, g, Cp, nchunks);
  TECM_CHECK_LAUNCH("tecm_spatial_bwd");
  return TECM_OK;
}
