// Backward of the fused stage a-1..a-3 (SpatioTemporalEmbedding modules.py:230-266 + GATv2Conv modules.py:329-336,
// :356 + residual tec_mollm.py:94).  Nothing was saved by the forward: the kernel recomputes x_l / x_r from x (cheaper
// than storing (B,L,N,22) x 3) and reduces straight to PARAMETER gradients -- x needs no gradient.
//
// Same skeleton as the forward (spatial_fwd.hip): work item = (tile of <= 128 target nodes, graph (b,t)), items
// tile-major, one persistent 512-thread block per CU owning a contiguous item range, so the CSR slices (by target and
// by source), the node-embedding rows of the window and the weights are staged once per block / tile.  Per item:
//   A   x_l (window) and x_r (tile) on the f32 matrix cores from the k-major window image hT = [x | 1 | emb]^T;
//   B1  by TARGET: a lane pair per (node, head) shares the node's in-edges (even / odd slots); one sweep computes the
//       logits e and dalpha = <dout_i, x_l[j]> with online-softmax statistics (pair-combined through DPP), a second
//       one turns them into (alpha~, de) per edge -- parked in an LDS edge array -- and accumulates d x_r and d att;
//   B2  by SOURCE: a thread per (window row, head) walks the tile's edges that leave the row (host-built by-source
//       lists), reads (alpha~, de) back and sums  d x_l[j] = sum_i alpha~_ij dout_i + de_ij att (.) lrelu'(s_ij)
//       in registers -- no LDS float atomics, no recomputation of logits or exponentials; d x_l overwrites x_l in place;
//   C   every weight gradient is an outer product  sum_rows [x | 1 | emb]^T (x) [d x_l | d x_r | dout]  on the matrix
//       cores (v_mfma_f32_16x16x4_f32, accumulators live in registers across all items of the block); the ones row
//       yields the per-item column sums that the temporal tables need (d temb_g = sum_n d h[n, Cin:]).
// The node table receives  sum_g d h[g, n, Cin:]  once per tile from register accumulators of d x_l, d x_r, dout.
#include "spatial_common.h"

using namespace tecm_spatial;

namespace {

constexpr int BT = 512;        // 8 waves; waves 0-3 work on head 0, waves 4-7 on head 1
constexpr int MAXI = 40;       // items one block may own (host sizes the grid accordingly)
constexpr int NPX = 4;         // float2 registers per thread: x rows of the next item
constexpr int NPG = 2;         // float4 registers per thread: dout rows of the next item
constexpr int HR = C + 1;      // rows of hT: [x (Cin) | ones | emb (Demb)]
constexpr int SRC_R_MAX = 2;   // by-source rounds a thread may own ((window rows) <= SRC_R * 256); 1 when the host
                               // tiles the graph so that every window has <= 256 rows (the regular grid does)

struct BwdArgs {               // the kernel's single by-value argument: kernarg offset 0
  TecmSpatial d;
  TecmSpatialGrads g;
  int total;
};

// prologue scratch (floats, relative to map.scr)
constexpr int SCR_WL = 0, SCR_WR = C * C, SCR_ATT = 2 * C * C, SCR_BL = SCR_ATT + 32, SCR_BR = SCR_BL + 32,
              SCR_UW = SCR_BR + 32, SCR_BIASL = SCR_UW + 4 * 32, SCR_BIASR = SCR_BIASL + 32, SCR_CS = SCR_BIASR + 32,
              SCR_BSUM = SCR_CS + 2 * 8 * 96, SCR_DATT = SCR_BSUM + 2 * 32, SCR_FLOATS = SCR_DATT + 32;

struct Map {                   // LDS map, float offsets into smem (ints share the same 4-byte cells)
  int P, wm4, hT_sz;
  int hT, nodeT, xl, xr, gt, ea, rsum, gsum, tb, ti, scr, eptr, ecol, sptr, scol, total;
};
__host__ __device__ inline Map make_map(const TecmSpatial& d) {
  Map m;
  m.P = d.win_max | 1;
  m.wm4 = (d.win_max + 3) & ~3;
  m.hT_sz = (HR * m.P + 3) & ~3;
  const int T = d.tile_nodes, E = d.tile_edges_max;
  m.hT = 0;                                    // [HR][P]   k-major window image  [x | 1 | node_emb + temb]
  m.xl = m.hT + m.hT_sz;                       // [wm4][CP] x_l of the window (head-sliced), later d x_l
  m.xr = m.xl + m.wm4 * CP;                    // [T][CP]   x_r of the tile, later d x_r
  m.gt = m.xr + T * CP;                        // [T][CP]   dout rows of the tile (head-sliced)
  m.ea = m.gt + T * CP;                        // [E + T][2 heads][2]   per edge: (e, dalpha) then (alpha~, de)
  m.nodeT = m.ea + (E + T) * 4;                // [Demb][P] static node-embedding rows of the window
  m.rsum = m.nodeT + ((d.Demb * m.P + 3) & ~3);   // [T][CP] sum over the tile's graphs of d x_r (node-table gradient)
  m.gsum = m.rsum + T * CP;                    // [T][CP] ... of dout
  m.tb = m.gsum + T * CP;                      // [MAXI][32] temporal embedding per item
  m.ti = m.tb + MAXI * 32;                     // [MAXI][4]  time indices per item (ints)
  m.scr = m.ti + MAXI * 4;                     // prologue scratch + small per-item vectors
  m.eptr = m.scr + SCR_FLOATS;                 // [T + 1]     CSR slice of the tile (by target)
  m.ecol = m.eptr + T + 1;                     // [E]         window-relative sources
  m.sptr = m.ecol + E;                         // [wm4 + 1]   the tile's edges grouped by source row
  m.scol = m.sptr + m.wm4 + 1;                 // [E]         (tile target << 16) | slot
  m.total = m.scol + E;
  return m;
}

// The final reduction of the waves' MFMA accumulators needs RED_FLOATS of LDS that must not overlap the scratch block:
// it reuses the (dead) data arrays when they are large enough, otherwise it sits behind everything (tiny graphs).
constexpr int RED_FLOATS = (BT / 64) * 8 * 4 * 64;
__host__ __device__ inline int red_offset(const Map& m) { return m.scr >= RED_FLOATS ? 0 : m.total; }

#ifdef SPB_STAMPS
__device__ unsigned long long g_spb_stamps[16];
#define SPB_T(slot)                                                        \
  do {                                                                     \
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {                      \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
      atomicAdd(&g_spb_stamps[slot], now_ - stamp_);                       \
      stamp_ = now_;                                                       \
    }                                                                      \
  } while (0)
#else
#define SPB_T(slot) do {} while (0)
#endif

__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 a;
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] = v;
  return a;
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// value of the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]: no LDS, one instruction
__device__ __forceinline__ float swap1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

struct TileState {
  int n0, n1, lo, hi;
};

// ---------------------------------------------------------------------------------------- out-of-line helpers
// Block prologue: weights through LDS scratch (coalesced global reads only), u weights, slot-ordered bias vectors,
// per-item temporal embeddings and time indices.
__device__ __attribute__((noinline)) void bwd_prologue(const BwdArgs* ap, int it0, int nit, float* smem, bool tf_uniform) {
  const TecmSpatial& d = ap->d;
  const Map m = make_map(d);            // rebuilt here: a struct argument of an out-of-line call would go through scratch
  const int tid = threadIdx.x, Demb = d.Demb;
  const int scr = m.scr;
  for (int i = tid; i < C * C; i += BT) {
    smem[scr + SCR_WL + i] = d.Wl[i];
    smem[scr + SCR_WR + i] = d.Wr[i];
  }
  if (tid < C) {
    smem[scr + SCR_ATT + tid] = d.att[tid];
    smem[scr + SCR_BL + tid] = d.bl[tid];
    smem[scr + SCR_BR + tid] = d.br[tid];
  }
  for (int i = tid; i < d.tile_nodes * CP; i += BT) smem[m.gt + i] = 0.f;   // pad slots of the dout tile stay zero
  if (tf_uniform) {
    for (int i = tid; i < nit * Demb; i += BT) {
      const int q = i / Demb, e = i - q * Demb;
      const Item it = decode_item(d, it0 + q);
      const TimeIdx ti = load_time_idx(d, it.b, it.t, 0);
      smem[m.tb + q * 32 + e] = temporal_emb(d, ti, e);
      if (e == 0) {
        int* tix = reinterpret_cast<int*>(smem + m.ti + q * 4);
        tix[0] = ti.tod; tix[1] = ti.doy; tix[2] = ti.year; tix[3] = ti.season;
      }
    }
  }
  __syncthreads();
  if (tid < 4 * C) {                                         // u weights: (matrix mm, head h, input k)
    const int mm = tid / (2 * C), h = (tid / C) & 1, k = tid % C;
    const float* W = smem + scr + (mm ? SCR_WR : SCR_WL);
    float v = 0.f;
    for (int a = 0; a < CH; ++a) v = fmaf(smem[scr + SCR_ATT + h * CH + a], W[(h * CH + a) * C + k], v);
    smem[scr + SCR_UW + (2 * mm + h) * 32 + k] = v;
  }
  if (tid < 64) {                                            // accumulator-init vectors of x_l / x_r in slot order
    const int mm = tid >> 5, sl = tid & 31;
    const float* bv = smem + scr + (mm ? SCR_BR : SCR_BL);
    const int ch = chan_of(sl);
    float v = 0.f;
    if (ch >= 0) {
      v = bv[ch];
    } else if (sl == CH || sl == 2 * CH + 1) {
      const int h = sl == CH ? 0 : 1;
      for (int a = 0; a < CH; ++a) v = fmaf(smem[scr + SCR_ATT + h * CH + a], bv[h * CH + a], v);
    }
    smem[scr + (mm ? SCR_BIASR : SCR_BIASL) + sl] = v;
  }
  if (tid < 2 * 32 + 32) smem[scr + SCR_BSUM + tid] = 0.f;   // bias-gradient and d att block sums (SCR_BSUM, SCR_DATT)
  // (per-item column sums SCR_CS: [item parity][wave][3][32], fully rewritten by phase C of every item)
}

__device__ __attribute__((noinline)) TileState bwd_tile_switch(const BwdArgs* ap, int tile, float* smem, bool tf_uniform) {
  const TecmSpatial& d = ap->d;
  const Map m = make_map(d);
  const TecmSpatialGrads& g = ap->g;
  const int tid = threadIdx.x;
  int* eptr = reinterpret_cast<int*>(smem + m.eptr);
  int* ecol = reinterpret_cast<int*>(smem + m.ecol);
  int* sptr = reinterpret_cast<int*>(smem + m.sptr);
  int* scol = reinterpret_cast<int*>(smem + m.scol);
  TileState ts;
  ts.n0 = tile * d.tile_nodes;
  ts.n1 = min(d.N, ts.n0 + d.tile_nodes);
  ts.lo = d.tile_lo[tile];
  ts.hi = d.tile_hi[tile];
  const int W = ts.hi - ts.lo;
  const int ebase = d.rowptr[ts.n0];
  for (int r = tid; r <= ts.n1 - ts.n0; r += BT) eptr[r] = d.rowptr[ts.n0 + r] - ebase;
  const int ne = d.rowptr[ts.n1] - ebase;
  for (int r = tid; r < ne; r += BT) {
    ecol[r] = d.colidx[ebase + r] - ts.lo;
    scol[r] = g.src_col[ebase + r];
  }
  const int pbase = g.src_ptr_off[tile];
  for (int r = tid; r <= W; r += BT) sptr[r] = g.src_ptr[pbase + r];
  for (int r = tid; r < W; r += BT) smem[m.hT + d.Cin * m.P + r] = 1.0f;          // the ones row
  for (int r = tid; r < 2 * d.tile_nodes * CP; r += BT) smem[m.rsum + r] = 0.f;    // d x_r and dout sums of this tile
  if (tf_uniform) {                                          // static node-embedding rows of the window
    const int Demb = d.Demb;
    for (int i = tid; i < W * Demb; i += BT) {
      const int w = i / Demb, e = i - w * Demb;
      smem[m.nodeT + e * m.P + w] = d.node_tab[(int64_t)(ts.lo + w) * Demb + e];
    }
  }
  return ts;
}

// per-node time features (general path): embedding rows of the window for this graph
__device__ __attribute__((noinline)) void bwd_stage_emb_rows(const BwdArgs* ap, int b, int t, int lo, int wa, int wb,
                                                             float* smem) {
  const TecmSpatial& d = ap->d;
  const Map m = make_map(d);
  const int Demb = d.Demb;
  for (int i = threadIdx.x; i < (wb - wa) * Demb; i += BT) {
    const int w = wa + i / Demb, e = i % Demb;
    const TimeIdx ti = load_time_idx(d, b, t, lo + w);
    smem[m.hT + (d.Cin + 1 + e) * m.P + w] = d.node_tab[(int64_t)(lo + w) * Demb + e] + temporal_emb(d, ti, e);
  }
}

// embedding part of d h for one row:  sum_a d x_l[w, a] Wl[a, Cin+e]  (+ tile rows: d x_r . Wr + dout)
__device__ __forceinline__ float demb_of_row(const float* smem, const Map& m, int Cin, int w, int ta, int tb, int e,
                                             bool residual, int xr_base, int gt_base) {
  const float* Wl = smem + m.scr + SCR_WL + Cin + e;
  const float* Wr = smem + m.scr + SCR_WR + Cin + e;
  const float* xlr = smem + m.xl + w * CP;
  float v = 0.f;
#pragma unroll
  for (int a = 0; a < C; ++a) v = fmaf(xlr[slot_of(a)], Wl[a * C], v);
  if (w >= ta && w < tb) {
    const float* xrr = smem + xr_base + (w - ta) * CP;
#pragma unroll
    for (int a = 0; a < C; ++a) v = fmaf(xrr[slot_of(a)], Wr[a * C], v);
    if (residual) v += smem[gt_base + (w - ta) * CP + slot_of(Cin + e)];
  }
  return v;
}

// per-node time features: the temporal tables get d h[w, Cin:] row by row (general path, once per item)
__device__ __attribute__((noinline)) void bwd_temporal_per_node(const BwdArgs* ap, int b, int t, int lo, int wa, int wb,
                                                                int ta, int tb, const float* smem, bool residual) {
  const TecmSpatial& d = ap->d;
  const Map m = make_map(d);
  const TecmSpatialGrads& g = ap->g;
  const int Demb = d.Demb;
  for (int i = threadIdx.x; i < (wb - wa) * Demb; i += BT) {
    const int w = wa + i / Demb, e = i % Demb;
    const float v = demb_of_row(smem, m, d.Cin, w, ta, tb, e, residual, m.xr, m.gt);      // this item's d x_r, dout
    const TimeIdx ti = load_time_idx(d, b, t, lo + w);
    atomicAdd(&g.d_tod_tab[ti.tod * Demb + e], v);
    atomicAdd(&g.d_doy_tab[ti.doy * Demb + e], v);
    atomicAdd(&g.d_year_tab[ti.year * Demb + e], v);
    atomicAdd(&g.d_season_tab[ti.season * Demb + e], v);
  }
}

// node table: the rows of x_l / x_r / gt hold  sum over the tile's graphs  of d x_l / d x_r / dout at this point
__device__ __attribute__((noinline)) void bwd_node_table(const BwdArgs* ap, int lo, int W, int ta, int tb,
                                                         const float* smem, bool residual) {
  const TecmSpatial& d = ap->d;
  const Map m = make_map(d);
  const int Demb = d.Demb;
  for (int i = threadIdx.x; i < W * Demb; i += BT) {
    const int w = i / Demb, e = i - w * Demb;
    atomicAdd(&ap->g.d_node_tab[(int64_t)(lo + w) * Demb + e],
              demb_of_row(smem, m, d.Cin, w, ta, tb, e, residual, m.rsum, m.gsum));           // the tile's sums
  }
}

__device__ __forceinline__ void load12(const float* p, float (&v)[CH + 1]) {
  const float4* q = reinterpret_cast<const float4*>(p);
  const float4 a = q[0], b = q[1], c = q[2];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ __forceinline__ void store12(float* p, const float (&v)[CH], float last) {
  float4* q = reinterpret_cast<float4*>(p);
  q[0] = make_float4(v[0], v[1], v[2], v[3]);
  q[1] = make_float4(v[4], v[5], v[6], v[7]);
  q[2] = make_float4(v[8], v[9], v[10], last);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int SRC_R>
__global__ __launch_bounds__(BT, 2) void spatial_bwd_kernel(const BwdArgs args) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TecmSpatial& d = args.d;
  const TecmSpatialGrads& gr = args.g;
  const BwdArgs* ap = reinterpret_cast<const BwdArgs*>(kernarg_base());
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = wave >> 2;                                  // this thread's head in B1 and B2 (wave-uniform)
  const int Cin = d.Cin, Demb = d.Demb, L = d.L, N = d.N, T = d.tile_nodes, E = d.tile_edges_max;
  const Map m = make_map(d);
  const int P = m.P;
  int* eptr = reinterpret_cast<int*>(smem + m.eptr);
  int* ecol = reinterpret_cast<int*>(smem + m.ecol);
  int* sptr = reinterpret_cast<int*>(smem + m.sptr);
  int* scol = reinterpret_cast<int*>(smem + m.scol);

  const int nblk = gridDim.x, blk = blockIdx.x;
  const int it0 = (int)((int64_t)blk * args.total / nblk);
  const int nit = (int)((int64_t)(blk + 1) * args.total / nblk) - it0;   // <= MAXI (host)
  float* part = gr.partials + (int64_t)blk * gr.partial_ld;
  const bool tf_uniform = Demb > 0 && d.tf_sn == 0;
  const bool residual = !(d.flags & TECM_SPATIAL_NO_RESIDUAL);

  bwd_prologue(ap, it0, nit, smem, tf_uniform);
  __syncthreads();
  // MFMA B operands of the recomputation (32x32x2): lane (c31, kq) reads column c31 of the slot-ordered W^T,
  // k = 2s + kq, from the LDS scratch for every task (registers are the scarce resource of this kernel)
  const int c31 = lane & 31, kq = lane >> 5;
  int wl_off, wr_off;
  bool w_on;
  {
    const int ch = chan_of(c31);
    const bool is_u = c31 == CH || c31 == 2 * CH + 1;
    wl_off = m.scr + (ch >= 0 ? SCR_WL + ch * C : SCR_UW + (c31 == CH ? 0 : 1) * 32) + kq;
    wr_off = m.scr + (ch >= 0 ? SCR_WR + ch * C : SCR_UW + (c31 == CH ? 2 : 3) * 32) + kq;
    w_on = ch >= 0 || is_u;
  }
  float att[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) att[c] = smem[m.scr + SCR_ATT + hh * CH + c];
  const uint32_t dth = d.alpha_drop.p > 0.f ? tecm_drop_thresh(d.alpha_drop.p) : 0u;
  const float dinv = d.alpha_drop.p > 0.f ? 1.0f / (1.0f - d.alpha_drop.p) : 1.0f;
  const uint64_t dseed = tecm_seed_now(d.alpha_drop.seed, d.alpha_drop.seed_dev);
  const bool even_cin = (Cin & 1) == 0;
  const unsigned cin_magic = (unsigned)((0x100000000ull + Cin - 1) / Cin);

  // ---- accumulators that live across the block's items
  f32x4 accL[2][2], accR[2][2], accG[2];                     // [row tile of hT][column tile of slots]
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    accG[a] = zero4();
#pragma unroll
    for (int b = 0; b < 2; ++b) { accL[a][b] = zero4(); accR[a][b] = zero4(); }
  }
  float dbl_acc[2] = {0.f, 0.f}, dbr_acc[2] = {0.f, 0.f};    // bias gradients (the ones row, moved out per item)
  float datt[CH];                                            // d att of head hh
  float DXL[SRC_R][CH];                                      // sum over the tile's graphs of d x_l, rows tid&255 + 256 r
                                                             // (the sums of d x_r and dout live in LDS: m.rsum, m.gsum)
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    datt[c] = 0.f;
#pragma unroll
    for (int r = 0; r < SRC_R; ++r) DXL[r][c] = 0.f;
  }

  const int pr = tid >> 1, sub = tid & 1;                    // B1: lane pair `pr` works on (tile node pr & 127, head hh)
  const int tn = pr & 127;
  const int sw = tid & 255;                                  // B2: window row wa + sw + 256 r, head hh

  int cur_tile = -1, n0 = 0, n1 = 0, lo = 0, hi = 0;
  bool x_ready = false, g_ready = false;
  float2 pfx[NPX];
  float4 pfg[NPG];
#ifdef SPB_STAMPS
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif

  // flush of everything that belongs to the current tile: node-table gradient from the register sums
  auto flush_tile = [&]() {
    if (cur_tile < 0 || Demb == 0) return;
    const int W = hi - lo, ta = n0 - lo, tb = n1 - lo;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SRC_R; ++r) {
      const int w = sw + 256 * r;
      if (w < W) store12(smem + m.xl + w * CP + hh * 12, DXL[r], 0.f);
#pragma unroll
      for (int c = 0; c < CH; ++c) DXL[r][c] = 0.f;
    }
    __syncthreads();
    bwd_node_table(ap, lo, W, ta, tb, smem, residual);
    __syncthreads();
  };

  for (int q = 0; q < nit; ++q) {
    const Item it = decode_item(d, it0 + q);
    if (it.tile != cur_tile) {
      flush_tile();
      __syncthreads();
      cur_tile = it.tile;
      const TileState ts = bwd_tile_switch(ap, it.tile, smem, tf_uniform);
      n0 = __builtin_amdgcn_readfirstlane(ts.n0);
      n1 = __builtin_amdgcn_readfirstlane(ts.n1);
      lo = __builtin_amdgcn_readfirstlane(ts.lo);
      hi = __builtin_amdgcn_readfirstlane(ts.hi);
      x_ready = false;
      g_ready = false;
      __syncthreads();                                       // nodeT is read right below
    }
    const int ta = n0 - lo, tb = n1 - lo, nt = n1 - n0;      // tile rows inside the window
    const int wa = it.use_edges ? 0 : ta, wb = it.use_edges ? hi - lo : tb;
    const int64_t grow = ((int64_t)it.b * L + it.t) * N;

    // ---- staging: x rows, embedding rows, dout tile (head-sliced)
    if (!x_ready) {
      const float* xb = d.x + (grow + lo + wa) * Cin;
      const int cnt = (wb - wa) * Cin;
      if (even_cin) {
        const float2* xb2 = reinterpret_cast<const float2*>(xb);
        for (int f = tid; f < (cnt >> 1); f += BT) {
          const float2 v = xb2[f];
          const int idx = 2 * f, row = (int)__umulhi((unsigned)idx, cin_magic), k = idx - row * Cin;
          smem[m.hT + k * P + wa + row] = v.x;
          smem[m.hT + (k + 1) * P + wa + row] = v.y;
        }
      } else {
        for (int f = tid; f < cnt; f += BT) {
          const int row = f / Cin, k = f - row * Cin;
          smem[m.hT + k * P + wa + row] = xb[f];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < NPX; ++r) {
        const int f = tid + r * BT;
        if (f < (((wb - wa) * Cin) >> 1)) {
          const int idx = 2 * f, row = (int)__umulhi((unsigned)idx, cin_magic), k = idx - row * Cin;
          smem[m.hT + k * P + wa + row] = pfx[r].x;
          smem[m.hT + (k + 1) * P + wa + row] = pfx[r].y;
        }
      }
    }
    if (Demb > 0) {
      if (tf_uniform) {
        for (int e = wave; e < Demb; e += BT / 64) {        // a wave per embedding row: no division, scalar temb
          const float te = smem[m.tb + q * 32 + e];
          for (int w = wa + lane; w < wb; w += 64) smem[m.hT + (Cin + 1 + e) * P + w] = smem[m.nodeT + e * P + w] + te;
        }
      } else {
        bwd_stage_emb_rows(ap, it.b, it.t, lo, wa, wb, smem);
      }
    }
    {
      const float4* gsrc = reinterpret_cast<const float4*>(gr.dout + (grow + n0) * CP);
#pragma unroll
      for (int r = 0; r < NPG; ++r) {
        const int f = tid + r * BT;
        if (f < nt * (CP / 4)) {
          const float4 v = g_ready ? pfg[r] : gsrc[f];
          const int row = f / (CP / 4), c = 4 * (f - row * (CP / 4));
          float* dst = smem + m.gt + row * CP;
          dst[slot_of(c)] = v.x;
          dst[slot_of(c + 1)] = v.y;
          if (c + 2 < C) dst[slot_of(c + 2)] = v.z;
          if (c + 3 < C) dst[slot_of(c + 3)] = v.w;
        }
      }
      for (int f = tid + NPG * BT; f < nt * (CP / 4); f += BT) {             // tiles wider than the prefetch window
        const float4 v = gsrc[f];
        const int row = f / (CP / 4), c = 4 * (f - row * (CP / 4));
        float* dst = smem + m.gt + row * CP;
        dst[slot_of(c)] = v.x;
        dst[slot_of(c + 1)] = v.y;
        if (c + 2 < C) dst[slot_of(c + 2)] = v.z;
        if (c + 3 < C) dst[slot_of(c + 3)] = v.w;
      }
    }
    __syncthreads();
    SPB_T(1);

    // ---- phase A: x_l (window) and x_r (tile), 32-row blocks dealt to the eight waves
    {
      const int nl = (wb - wa + 31) >> 5, nr = (tb - ta + 31) >> 5;
      for (int task = wave; task < nl + nr; task += BT / 64) {
        const bool isr = task >= nl;
        const int r0 = isr ? ta + 32 * (task - nl) : wa + 32 * task;
        const int rend = isr ? tb : wb;
        f32x16 acc = splat16(smem[m.scr + (isr ? SCR_BIASR : SCR_BIASL) + c31]);
        const int row = min(r0 + c31, rend - 1);
        float av[C / 2], bv[C / 2];
        const int wo = isr ? wr_off : wl_off;
#pragma unroll
        for (int s = 0; s < C / 2; ++s) {
          const int k = 2 * s + kq;                          // hT row of input k: the ones row sits at index Cin
          av[s] = smem[m.hT + (k + (k >= Cin ? 1 : 0)) * P + row];
          bv[s] = w_on ? smem[wo + 2 * s] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < C / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
        const int oo = (isr ? m.xr - ta * CP : m.xl) + (r0 + 4 * kq) * CP + c31;
        if (c31 < CP) {
          if (r0 + 32 <= rend) {
#pragma unroll
            for (int e = 0; e < 16; ++e) smem[oo + ((e & 3) + 8 * (e >> 2)) * CP] = acc[e];
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (r0 + 4 * kq + (e & 3) + 8 * (e >> 2) < rend) smem[oo + ((e & 3) + 8 * (e >> 2)) * CP] = acc[e];
          }
        }
      }
    }
    __syncthreads();
    SPB_T(2);

    // ---- prefetch the next item's x rows and dout rows (registers; committed at the top of the next iteration)
    x_ready = false;
    g_ready = false;
    if (q + 1 < nit) {
      const Item nx = decode_item(d, it0 + q + 1);
      if (nx.tile == cur_tile) {
        const int64_t ngrow = ((int64_t)nx.b * L + nx.t) * N;
        const int nwa = nx.use_edges ? 0 : ta, nwb = nx.use_edges ? hi - lo : tb;
        const int cnt2 = ((nwb - nwa) * Cin) >> 1;
        if (even_cin && cnt2 <= NPX * BT) {
          x_ready = true;
          const float2* xb2 = reinterpret_cast<const float2*>(d.x + (ngrow + lo + nwa) * Cin);
#pragma unroll
          for (int r = 0; r < NPX; ++r) {
            const int f = tid + r * BT;
            if (f < cnt2) pfx[r] = xb2[f];
          }
        }
        g_ready = true;
        const float4* gsrc = reinterpret_cast<const float4*>(gr.dout + (ngrow + n0) * CP);
#pragma unroll
        for (int r = 0; r < NPG; ++r) {
          const int f = tid + r * BT;
          if (f < nt * (CP / 4)) pfg[r] = gsrc[f];
        }
      }
    }


    // ---- phase B1: by target.  Lane pair (sub = 0 / 1) of (tile node tn, head hh) takes the even / odd slots.
    float dxr[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) dxr[c] = 0.f;
    const int i = n0 + tn;
    const bool tgt = tn < T && i < n1;
    if (tgt) {
      const int wi = i - lo;
      float xr[CH + 1], gv[CH + 1];
      load12(smem + m.xr + tn * CP + hh * 12, xr);
      load12(smem + m.gt + tn * CP + hh * 12, gv);
      const int e0 = it.use_edges ? eptr[tn] : 0;
      const int deg = it.use_edges ? eptr[tn + 1] - e0 : 0;
      const int64_t rowi = (int64_t)(it.t * d.B + it.b) * N + i;            // row in the reference's (L*B*N) flattening
      const uint64_t dbase = (uint64_t)((rowi * H + hh) * d.alpha_drop.ld);
      const float base = (0.6f * LOG2E) * xr[CH];
      // sweep 1: logits, dalpha, online softmax statistics over this lane's slots (slot deg = the implicit self loop)
      float mx = -INFINITY, z = 0.f, num = 0.f;
      for (int s = sub; s <= deg; s += 2) {
        const int j = s < deg ? ecol[e0 + s] : wi;
        const int pos = s < deg ? e0 + s : E + tn;
        float a[CH + 1];
        load12(smem + m.xl + j * CP + hh * 12, a);
        // e = log2(e) * (0.6 (u_l + u_r) + 0.4 sum_c att_c |x_l[j,c] + x_r[i,c]|); even / odd channels in separate chains
        float t0 = 0.f, u0 = 0.f, da = 0.f, db = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (c & 1) {
            u0 = fmaf(att[c], fabsf(a[c] + xr[c]), u0);
            db = fmaf(gv[c], a[c], db);
          } else {
            t0 = fmaf(att[c], fabsf(a[c] + xr[c]), t0);
            da = fmaf(gv[c], a[c], da);
          }
        }
        const float e = fmaf(0.4f * LOG2E, t0 + u0, fmaf(0.6f * LOG2E, a[CH], base));
        da += db;
        if (dth) da *= tecm_drop_mult(dseed, dbase + s, dth, dinv);
        *reinterpret_cast<float2*>(smem + m.ea + (pos * 2 + hh) * 2) = make_float2(e, da);
        const float mn = fmaxf(mx, e);
        const float corr = __builtin_amdgcn_exp2f(mx - mn), pw = __builtin_amdgcn_exp2f(e - mn);
        z = z * corr + pw;
        num = num * corr + pw * da;
        mx = mn;
      }
      // pair combine: the even lane always owns slot 0, so the common maximum is finite
      const float mo = swap1(mx), zo = swap1(z), no = swap1(num);
      const float mm = fmaxf(mx, mo);
      const float k0 = __builtin_amdgcn_exp2f(mx - mm), k1 = __builtin_amdgcn_exp2f(mo - mm);
      z = z * k0 + zo * k1;
      num = num * k0 + no * k1;
      const float zinv = 1.0f / (z + 1e-16f);
      const float dot = num * zinv;
      // sweep 2: (alpha~, de) per edge into the edge array; d x_r and d att
      for (int s = sub; s <= deg; s += 2) {
        const int j = s < deg ? ecol[e0 + s] : wi;
        const int pos = s < deg ? e0 + s : E + tn;
        float2* slot = reinterpret_cast<float2*>(smem + m.ea + (pos * 2 + hh) * 2);
        const float2 ed = *slot;                             // (e, dalpha * mult)
        float a[CH + 1];
        load12(smem + m.xl + j * CP + hh * 12, a);
        const float alpha = __builtin_amdgcn_exp2f(ed.x - mm) * zinv;
        float mult = 1.0f;
        if (dth) mult = tecm_drop_mult(dseed, dbase + s, dth, dinv);
        const float de = alpha * (ed.y - dot);
        *slot = make_float2(alpha * mult, de);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float sv = a[c] + xr[c];
          const bool pos_ = sv > 0.f;
          dxr[c] = fmaf(de, pos_ ? att[c] : NEG_SLOPE * att[c], dxr[c]);
          datt[c] = fmaf(de, pos_ ? sv : NEG_SLOPE * sv, datt[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) dxr[c] += swap1(dxr[c]);  // both lanes now hold d x_r[i, head]
      if (sub == 0) {                                        // the tile's running sums (node-table gradient), in LDS
        float4* rs = reinterpret_cast<float4*>(smem + m.rsum + tn * CP + hh * 12);
        float4 r0 = rs[0], r1 = rs[1], r2 = rs[2];
        r0.x += dxr[0]; r0.y += dxr[1]; r0.z += dxr[2]; r0.w += dxr[3]; r1.x += dxr[4]; r1.y += dxr[5];
        r1.z += dxr[6]; r1.w += dxr[7]; r2.x += dxr[8]; r2.y += dxr[9]; r2.z += dxr[10];
        rs[0] = r0; rs[1] = r1; rs[2] = r2;
        if (residual) {
          float4* gs = reinterpret_cast<float4*>(smem + m.gsum + tn * CP + hh * 12);
          float4 g0 = gs[0], g1 = gs[1], g2 = gs[2];
          g0.x += gv[0]; g0.y += gv[1]; g0.z += gv[2]; g0.w += gv[3]; g1.x += gv[4]; g1.y += gv[5];
          g1.z += gv[6]; g1.w += gv[7]; g2.x += gv[8]; g2.y += gv[9]; g2.z += gv[10];
          gs[0] = g0; gs[1] = g1; gs[2] = g2;
        }
      }
    }
    __syncthreads();
    SPB_T(3);

    // ---- phase B2: by source.  Thread (window row w, head hh) gathers d x_l[w] from the tile's edges leaving w.
#pragma unroll
    for (int r = 0; r < SRC_R; ++r) {
      const int w = sw + 256 * r;                            // window row (the register sums DXL[r] belong to it)
      if (w >= wa && w < wb) {
        float xl[CH + 1], acc[CH];
        load12(smem + m.xl + w * CP + hh * 12, xl);
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0.f;
        const bool self = w >= ta && w < tb;
        const int q0 = it.use_edges ? sptr[w] : 0;
        const int q1 = it.use_edges ? sptr[w + 1] : 0;
        const int selfc = ((w - ta) << 16) | (E + w - ta);   // the implicit self loop of a tile row: last "edge"
        for (int qq = q0; qq < q1 + (self ? 1 : 0); ++qq) {
          const int code = qq < q1 ? scol[qq] : selfc;       // (tile target << 16) | position in the edge array
          const int tt = code >> 16;
          const float2 ad = *reinterpret_cast<const float2*>(smem + m.ea + ((code & 0xffff) * 2 + hh) * 2);   // (alpha~, de)
          float xr[CH + 1], gv[CH + 1];
          load12(smem + m.xr + tt * CP + hh * 12, xr);
          load12(smem + m.gt + tt * CP + hh * 12, gv);
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const float sv = xl[c] + xr[c];
            acc[c] = fmaf(ad.x, gv[c], fmaf(ad.y, sv > 0.f ? att[c] : NEG_SLOPE * att[c], acc[c]));
          }
        }
        store12(smem + m.xl + w * CP + hh * 12, acc, 0.f);   // d x_l[w, head] replaces x_l (only this thread read it)
#pragma unroll
        for (int c = 0; c < CH; ++c) DXL[r][c] += acc[c];
      }
    }
    __syncthreads();
    // d x_r from the B1 registers into the (now dead) x_r rows
    if (tgt && sub == 0) store12(smem + m.xr + tn * CP + hh * 12, dxr, 0.f);
    __syncthreads();
    SPB_T(4);

    // ---- phase C: outer products on the matrix cores, 4 window rows per MFMA
    {
      const int nkl = (wb - wa + 3) >> 2, nkr = (tb - ta + 3) >> 2;
      const int i0 = lane & 15, k4 = lane >> 4;
      const bool a1_on = 16 + i0 < HR, b1_on = i0 < CP - 16;
      // window rows: [x | 1 | emb]^T (x) d x_l.  Two loops without branches around the MFMAs (the accumulators stay
      // in place), software pipelined: the operands of the next k-step are in flight while this one's MFMAs issue.
      {
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
        auto ldw = [&](int ks, float& x0, float& x1, float& y0, float& y1) {
          const int row = wa + 4 * ks + k4;
          const bool ok = row < wb;
          x0 = ok ? smem[m.hT + i0 * P + row] : 0.f;
          x1 = (ok && a1_on) ? smem[m.hT + (16 + i0) * P + row] : 0.f;
          y0 = ok ? smem[m.xl + row * CP + i0] : 0.f;
          y1 = (ok && b1_on) ? smem[m.xl + row * CP + 16 + i0] : 0.f;
        };
        if (wave < nkl) ldw(wave, a0, a1, b0, b1);
#pragma unroll 1
        for (int ks = wave; ks < nkl; ks += BT / 64) {
          float n0_ = 0.f, n1_ = 0.f, m0_ = 0.f, m1_ = 0.f;
          if (ks + BT / 64 < nkl) ldw(ks + BT / 64, n0_, n1_, m0_, m1_);
          accL[0][0] = MFMA16(a0, b0, accL[0][0]);
          accL[0][1] = MFMA16(a0, b1, accL[0][1]);
          accL[1][0] = MFMA16(a1, b0, accL[1][0]);
          accL[1][1] = MFMA16(a1, b1, accL[1][1]);
          a0 = n0_; a1 = n1_; b0 = m0_; b1 = m1_;
        }
      }
      // tile rows: [x | 1 | emb]^T (x) d x_r, and the ones row against dout (its column sums)
      {
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, g0 = 0.f, g1 = 0.f;
        auto ldt = [&](int ks, float& x0, float& x1, float& y0, float& y1, float& z0, float& z1) {
          const int row = ta + 4 * ks + k4;
          const bool ok = row < tb;
          const int tr = row - ta;
          x0 = ok ? smem[m.hT + i0 * P + row] : 0.f;
          x1 = (ok && a1_on) ? smem[m.hT + (16 + i0) * P + row] : 0.f;
          y0 = ok ? smem[m.xr + tr * CP + i0] : 0.f;
          y1 = (ok && b1_on) ? smem[m.xr + tr * CP + 16 + i0] : 0.f;
          z0 = ok ? smem[m.gt + tr * CP + i0] : 0.f;
          z1 = (ok && b1_on) ? smem[m.gt + tr * CP + 16 + i0] : 0.f;
        };
        if (wave < nkr) ldt(wave, a0, a1, b0, b1, g0, g1);
#pragma unroll 1
        for (int ks = wave; ks < nkr; ks += BT / 64) {
          float n0_ = 0.f, n1_ = 0.f, m0_ = 0.f, m1_ = 0.f, h0_ = 0.f, h1_ = 0.f;
          if (ks + BT / 64 < nkr) ldt(ks + BT / 64, n0_, n1_, m0_, m1_, h0_, h1_);
          const float as = Cin < 16 ? a0 : a1;               // the row tile that holds the ones row
          accR[0][0] = MFMA16(a0, b0, accR[0][0]);
          accR[0][1] = MFMA16(a0, b1, accR[0][1]);
          accR[1][0] = MFMA16(a1, b0, accR[1][0]);
          accR[1][1] = MFMA16(a1, b1, accR[1][1]);
          accG[0] = MFMA16(as, g0, accG[0]);
          accG[1] = MFMA16(as, g1, accG[1]);
          a0 = n0_; a1 = n1_; b0 = m0_; b1 = m1_; g0 = h0_; g1 = h1_;
        }
      }
      // the ones row (hT row Cin): per-item column sums -> LDS, bias gradients -> registers, then cleared
      const int rt1 = Cin >> 4, lr = Cin & 15;
      if (k4 == (lr >> 2)) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            if (rg == (lr & 3)) {
              float vl, vr;
              if (rt1 == 0) { vl = accL[0][ct][rg]; accL[0][ct][rg] = 0.f; vr = accR[0][ct][rg]; accR[0][ct][rg] = 0.f; }
              else          { vl = accL[1][ct][rg]; accL[1][ct][rg] = 0.f; vr = accR[1][ct][rg]; accR[1][ct][rg] = 0.f; }
              const float vg = accG[ct][rg];
              accG[ct][rg] = 0.f;
              const int cso = m.scr + SCR_CS + ((q & 1) * 8 + wave) * 96 + ct * 16 + i0;   // this wave's partial sums
              smem[cso + 0 * 32] = vl;                       // (LDS float atomics run at about a lane per 3.5 clk)
              smem[cso + 1 * 32] = vr;
              smem[cso + 2 * 32] = vg;
              dbl_acc[ct] += vl;
              dbr_acc[ct] += vr;
            }
          }
        }
      }
    }
    __syncthreads();
    SPB_T(5);

    // ---- temporal tables: d temb_g = sum_n d h[n, Cin:]  from the column sums of d x_l, d x_r, dout.  A dozen
    //      threads; the others go on to stage the next item (the column sums are double buffered by item parity and
    //      cleared here for the item after next, two barriers ahead of their next use)
    if (Demb > 0) {
      if (tf_uniform) {
        float* csl = smem + m.scr + SCR_CS + (q & 1) * 8 * 96;
        if (wave == 0) {
          // the eight waves' partial column sums -> row 0 (same wave reads it next: program order suffices)
          for (int k = lane; k < 96; k += 64) {
            float v = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < BT / 64; ++w8) v += csl[w8 * 96 + k];
            csl[k] = v;
          }
          // four lanes per embedding column e: lane part p sums 11 of the 44 terms (matrix p >> 1, half p & 1)
          const int* tix = reinterpret_cast<const int*>(smem + m.ti + q * 4);
          for (int e0 = 0; e0 < Demb; e0 += 16) {
            const int e = e0 + (lane >> 2), p4 = lane & 3;
            float v = 0.f;
            if (e < Demb) {
              const float* cs = csl + (p4 >> 1) * 32;
              const float* W = smem + m.scr + ((p4 >> 1) ? SCR_WR : SCR_WL) + Cin + e;
              const int abeg = (p4 & 1) * CH;
#pragma unroll
              for (int a = 0; a < CH; ++a) v = fmaf(cs[slot_of(abeg + a)], W[(abeg + a) * C], v);
              if (p4 == 0 && residual) v += csl[2 * 32 + slot_of(Cin + e)];
            }
            v += swap1(v);
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
            if (e < Demb && p4 == 0) {
              atomicAdd(&gr.d_tod_tab[tix[0] * Demb + e], v);
              atomicAdd(&gr.d_doy_tab[tix[1] * Demb + e], v);
              atomicAdd(&gr.d_year_tab[tix[2] * Demb + e], v);
              atomicAdd(&gr.d_season_tab[tix[3] * Demb + e], v);
            }
          }
        }
      } else {
        bwd_temporal_per_node(ap, it.b, it.t, lo, wa, wb, ta, tb, smem, residual);
        __syncthreads();                                     // the rows it read are restaged right away
      }
    }
    SPB_T(6);
  }

  // ---- block results
  flush_tile();
  __syncthreads();
  // the eight waves' MFMA accumulators through LDS: red[wave][8 accumulators][4 registers][64 lanes]
  {
    float* red = smem + red_offset(m);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          red[((wave * 8 + (0 * 4 + rt * 2 + ct)) * 4 + rg) * 64 + lane] = accL[rt][ct][rg];
          red[((wave * 8 + (1 * 4 + rt * 2 + ct)) * 4 + rg) * 64 + lane] = accR[rt][ct][rg];
        }
    const int lr = Cin & 15;
    if ((lane >> 4) == (lr >> 2)) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        atomicAdd(&smem[m.scr + SCR_BSUM + 0 * 32 + ct * 16 + (lane & 15)], dbl_acc[ct]);
        atomicAdd(&smem[m.scr + SCR_BSUM + 1 * 32 + ct * 16 + (lane & 15)], dbr_acc[ct]);
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float a = wave_sum(datt[c]);
      if (lane == 0) atomicAdd(&smem[m.scr + SCR_DATT + hh * CH + c], a);
    }
    __syncthreads();
    // partial row layout: dWl (C*C) | dbl (C) | dWr (C*C) | dbr (C) | datt (C) | (C unused)
    for (int o = tid; o < 2 * HR * CP; o += BT) {
      const int mm = o / (HR * CP), rem = o - mm * HR * CP;
      const int i = rem / CP, j = rem - i * CP;              // hT row, slot
      const int a = chan_of(j);
      if (a < 0 || i == Cin) continue;                       // u slots / padding; the ones row lives in dbl / dbr
      const int rt = i >> 4, ct = j >> 4, il = i & 15;
      const int rl = (il >> 2) * 16 + (j & 15), rg = il & 3;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < BT / 64; ++w) sum += red[((w * 8 + (mm * 4 + rt * 2 + ct)) * 4 + rg) * 64 + rl];
      const int k = i < Cin ? i : i - 1;                     // input channel of hT row i
      part[mm * (C * C + C) + a * C + k] = sum;
    }
    if (tid < 2 * CP) {
      const int mm = tid / CP, j = tid - mm * CP, a = chan_of(j);
      if (a >= 0) part[mm * (C * C + C) + C * C + a] = smem[m.scr + SCR_BSUM + mm * 32 + j];
    }
    if (tid < C) part[2 * (C * C + C) + tid] = smem[m.scr + SCR_DATT + tid];
  }
}

// Backward of the stand-alone SpatioTemporalEmbedding.forward (modules.py:230-266): out = cat([x, emb]), so
// d table[idx] += dout[..., Cin:] summed over the rows that looked idx up.  One block = 256 nodes of one graph (b, t):
// the node table takes one atomic per (row, column), the four temporal tables one per (block, column) after a
// reduction in LDS when the time features are constant over the nodes, one per row otherwise.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const TecmSpatial d, const TecmSpatialGrads g) {
  __shared__ float tsum[32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = blockIdx.x * 256 + tid;
  const int gm = blockIdx.y, b = gm / d.L, t = gm - b * d.L;
  const int Demb = d.Demb;
  const bool uni = d.tf_sn == 0;
  if (tid < 32) tsum[tid] = 0.f;
  __syncthreads();
  const bool on = n < d.N;
  const float* row = g.dout + ((int64_t)gm * d.N + (on ? n : 0)) * d.out_ld + d.Cin;
  TimeIdx ti = load_time_idx(d, b, t, uni ? 0 : (on ? n : 0));
  for (int e = 0; e < Demb; ++e) {
    const float v = on ? row[e] : 0.f;
    if (on) atomicAdd(&g.d_node_tab[(int64_t)n * Demb + e], v);
    if (uni) {
      const float sw = wave_sum(v);
      if (lane == 0) atomicAdd(&tsum[e], sw);
    } else if (on) {
      atomicAdd(&g.d_tod_tab[ti.tod * Demb + e], v);
      atomicAdd(&g.d_doy_tab[ti.doy * Demb + e], v);
      atomicAdd(&g.d_year_tab[ti.year * Demb + e], v);
      atomicAdd(&g.d_season_tab[ti.season * Demb + e], v);
    }
  }
  __syncthreads();
  if (uni && tid < Demb) {
    const float v = tsum[tid];
    atomicAdd(&g.d_tod_tab[ti.tod * Demb + tid], v);
    atomicAdd(&g.d_doy_tab[ti.doy * Demb + tid], v);
    atomicAdd(&g.d_year_tab[ti.year * Demb + tid], v);
    atomicAdd(&g.d_season_tab[ti.season * Demb + tid], v);
  }
}

}  // namespace

#ifdef SPB_STAMPS
extern "C" int tecm_debug_spb_stamps(unsigned long long* out16, int reset) {
  if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_spb_stamps), sizeof(g_spb_stamps));
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_spb_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

static int64_t bwd_blocks(const TecmSpatial& d) {
  const int64_t total = (int64_t)d.B * d.L * d.num_tiles;
  int64_t nblk = total < 256 ? total : 256;                  // one 512-thread block per CU
  if ((total + nblk - 1) / nblk > MAXI) nblk = (total + MAXI - 1) / MAXI;
  return nblk;
}

extern "C" int tecm_spatial_bwd_blocks(const TecmSpatial* dp) {
  if (dp == nullptr || dp->B <= 0 || dp->L <= 0 || dp->num_tiles <= 0) return TECM_E_ARG;
  return (int)bwd_blocks(*dp);
}

extern "C" int tecm_spatial_bwd(const TecmSpatial* dp, const TecmSpatialGrads* gp, void* stream) {
  TECM_REQUIRE(dp != nullptr && gp != nullptr, TECM_E_ARG, "tecm_spatial_bwd: null descriptor");
  const TecmSpatial& d = *dp;
  const TecmSpatialGrads& g = *gp;
  const int rc = check_common("tecm_spatial_bwd", d);
  if (rc) return rc;
  if (d.flags & TECM_SPATIAL_EMBED_ONLY) {                   // stand-alone SpatioTemporalEmbedding: table gradients only
    TECM_REQUIRE(d.Demb > 0 && d.Demb <= 32 && g.dout && d.out_ld >= C, TECM_E_ARG, "tecm_spatial_bwd(embed only): bad arguments");
    TECM_REQUIRE(g.d_node_tab && g.d_tod_tab && g.d_doy_tab && g.d_year_tab && g.d_season_tab, TECM_E_ARG,
                 "tecm_spatial_bwd(embed only): null table gradient");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((d.N + 255) / 256, d.B * d.L), dim3(256), 0, (hipStream_t)stream, d, g);
    TECM_CHECK_LAUNCH("tecm_spatial_bwd(embed only)");
    return TECM_OK;
  }
  TECM_REQUIRE(g.dout && g.partials, TECM_E_ARG, "tecm_spatial_bwd: null pointer");
  TECM_REQUIRE(d.Demb == 0 || (g.d_node_tab && g.d_tod_tab && g.d_doy_tab && g.d_year_tab && g.d_season_tab), TECM_E_ARG,
               "tecm_spatial_bwd: null table gradient");
  TECM_REQUIRE(tecm_aligned(g.dout, 16) && tecm_aligned(d.x, 8), TECM_E_ALIGN,
               "tecm_spatial_bwd: dout must be 16-byte aligned (24-float rows), x 8-byte aligned");
  TECM_REQUIRE(g.src_ptr && g.src_col && g.src_ptr_off, TECM_E_ARG, "tecm_spatial_bwd: by-source edge lists missing");
  const int64_t nblk = bwd_blocks(d);
  TECM_REQUIRE(g.num_blocks == nblk, TECM_E_ARG, "tecm_spatial_bwd: num_blocks must be %d = tecm_spatial_bwd_blocks() (got %d)",
               (int)nblk, g.num_blocks);
  TECM_REQUIRE(g.partial_ld >= 2 * C * C + 4 * C, TECM_E_ARG, "tecm_spatial_bwd: partial_ld must be >= %d", 2 * C * C + 4 * C);
  TECM_REQUIRE(d.win_max <= SRC_R_MAX * 256, TECM_E_LDS,
               "tecm_spatial_bwd: neighbour window of %d rows exceeds %d; renumber the graph (e.g. RCM) or shrink tile_nodes",
               d.win_max, SRC_R_MAX * 256);
  const Map m = make_map(d);
  size_t floats = (size_t)m.total;
  if (floats < (size_t)red_offset(m) + RED_FLOATS) floats = (size_t)red_offset(m) + RED_FLOATS;
  const size_t lds = sizeof(float) * floats;
  TECM_REQUIRE(lds <= (size_t)kLdsBudget, TECM_E_LDS,
               "tecm_spatial_bwd: neighbour window of %d rows needs %zu B of LDS (> 160 KiB); renumber the graph "
               "(e.g. RCM) or shrink tile_nodes", d.win_max, lds);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        kLdsBudget);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        kLdsBudget);
    attr_set = true;
  }
  BwdArgs args;
  args.d = d;
  args.g = g;
  args.total = (int)((int64_t)d.B * d.L * d.num_tiles);
  if (d.win_max <= 256)
    hipLaunchKernelGGL(spatial_bwd_kernel<1>, dim3((unsigned)nblk), dim3(BT), lds, (hipStream_t)stream, args);
  else
    hipLaunchKernelGGL(spatial_bwd_kernel<2>, dim3((unsigned)nblk), dim3(BT), lds, (hipStream_t)stream, args);
  TECM_CHECK_LAUNCH("tecm_spatial_bwd");
  return TECM_OK;
}
