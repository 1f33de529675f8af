// Forward of the fused stage a-1..a-3: SpatioTemporalEmbedding (modules.py:230-266) + GATv2Conv (modules.py:329-336,
// :356; torch_geometric semantics restated in oracle/ref_cpu.py:gatv2_conv) + residual (tec_mollm.py:94).
//
// HBM-bound by byte count (about 15 flop/byte): the only full-size traffic is one coalesced read of the (B,L,N,Cin)
// input and one coalesced write of the (B,L,N,24) output -- the reference's 4 gathers + 4 adds + cat + 2 permute
// copies + PyG's per-edge tensors never exist.
//
// Work item = (tile of <= 128 target nodes, graph (b,t)); items are numbered tile-major and every block owns a
// CONTIGUOUS range of them (<= MAXI), so that everything that depends on the tile alone is staged once per block:
// the CSR slice of the tile, the node-embedding rows of its neighbour window, the weights (MFMA B operands live in
// registers for the whole block), and -- computed in the block prologue for all of its items -- the temporal
// embedding of each graph folded into per-item bias vectors:
//      x_l = Wl [x | node_emb + temb_g] + bl = Wl [x | node_emb] + (Wl[:, Cin:] temb_g + bl)
// so the per-item staging is just the x rows of the window (prefetched into registers one phase ahead).
//
// Per item, three phases separated by block barriers:
//   1. dense: x_l (window) and x_r (tile) on the f32 matrix cores; the A operand is read from a K-MAJOR image
//      hT[k][w] of the window (consecutive lanes = consecutive rows: conflict-free ds_read_b32), the bias enters as
//      the accumulator's initial value;
//   2. edges: one thread per (target node, head) -- all 256 threads busy, the head is wave-uniform -- walks the
//      node's in-edges two at a time (two independent online-softmax chains, merged at the end), neighbour rows of
//      x_l come from LDS; the result overwrites the thread's own x_r slice;
//   3. the output tile leaves through contiguous float4 stores.
// Graphs with g = t*B + b >= graphs_with_edges see only their self loop (the reference's literal behaviour for
// everything but graph 0); their window is the tile itself.
#include "spatial_common.h"

using namespace tecm_spatial;

namespace {

constexpr int THREADS = 512;   // 8 waves: two SLOTS of 4 waves, each slot works on its own (tile, graph) item
constexpr int SLOT_T = 256;    // threads of a slot = (128 tile nodes) x (2 heads)
constexpr int MAXI = 40;       // items one block may own (host sizes the grid accordingly)
constexpr int NPF = 8;         // float2 registers per thread for the x prefetch of the slot's next item

#ifdef SPF_STAMPS
// diagnostic builds only (tools/build_variant.py): cycles per phase, summed over the items of a few blocks
__device__ unsigned long long g_spf_stamps[16];
#define SPF_T(slot)                                                        \
  do {                                                                     \
    if (tid == 0 && (blockIdx.x & 63) == 0) {                              \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
      atomicAdd(&g_spf_stamps[slot], now_ - stamp_);                       \
      stamp_ = now_;                                                       \
    }                                                                      \
  } while (0)
#define SPF_E(slot)                                                        \
  do {                                                                     \
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {                      \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
      atomicAdd(&g_spf_stamps[slot], now_ - estamp_);                      \
      estamp_ = now_;                                                      \
    }                                                                      \
  } while (0)
#else
#define SPF_T(slot) do {} while (0)
#define SPF_E(slot) do {} while (0)
#endif
#ifndef SPF_SKIP
#define SPF_SKIP 0     // diagnostic builds: bit0 no dense phase, bit1 no edge loop, bit2 no x prefetch
#endif

__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 a;
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] = v;
  return a;
}

__device__ __forceinline__ const TecmSpatial* kernarg_desc() {
  return reinterpret_cast<const TecmSpatial*>(kernarg_base());
}

struct TileState {
  int n0, n1, lo, hi;
};

__device__ __attribute__((noinline)) TileState tile_switch(const TecmSpatial* dp, int tile, float* smem, int slot_floats,
                                                           int P, int* eptr, int* ecol, bool tf_uniform) {
  const TecmSpatial& d = *dp;
  const int tid = threadIdx.x;
  TileState ts;
  ts.n0 = tile * d.tile_nodes;
  ts.n1 = min(d.N, ts.n0 + d.tile_nodes);
  ts.lo = d.tile_lo[tile];
  ts.hi = d.tile_hi[tile];
  const int ebase = d.rowptr[ts.n0];
  for (int r = tid; r <= ts.n1 - ts.n0; r += THREADS) eptr[r] = d.rowptr[ts.n0 + r] - ebase;
  const int ne = d.rowptr[ts.n1] - ebase;
  for (int r = tid; r < ne; r += THREADS) ecol[r] = d.colidx[ebase + r] - ts.lo;
  if (tf_uniform) {                                          // static node-embedding rows of the window
    const int Demb = d.Demb;
    for (int i = tid; i < (ts.hi - ts.lo) * Demb; i += THREADS) {
      const int w = i / Demb, e = i - w * Demb;
      const float v = d.node_tab[(int64_t)(ts.lo + w) * Demb + e];
      smem[(d.Cin + e) * P + w] = v;                         // slot 0 and slot 1 images
      smem[slot_floats + (d.Cin + e) * P + w] = v;
    }
  }
  return ts;
}

// per-node time features (general path): the embedding rows of the window depend on the graph
__device__ __attribute__((noinline)) void stage_emb_rows(const TecmSpatial* dp, int b, int t, int lo, int wa, int wb,
                                                         float* smem, int hT_off, int P) {
  const TecmSpatial& d = *dp;
  const int Demb = d.Demb;
  for (int i = threadIdx.x & (SLOT_T - 1); i < (wb - wa) * Demb; i += SLOT_T) {     // the slot's own 256 threads
    const int w = wa + i / Demb, e = i % Demb;
    const TimeIdx ti = load_time_idx(d, b, t, lo + w);
    smem[hT_off + (d.Cin + e) * P + w] = d.node_tab[(int64_t)(lo + w) * Demb + e] + temporal_emb(d, ti, e);
  }
}

// Block prologue, out of line: everything the block's items share is staged ONCE, through LDS, with coalesced
// global reads only (a persistent block pays this once, but a serial chain of dependent global loads per lane here
// would still cost tens of microseconds):
//   scratch (the not yet used x_l area of slot 0): Wl, Wr (22x22 each), att, bl, br, and the four u-weight vectors
//   UW[m][h][k] = sum_a att[h][a] W_m[h*11 + a][k];
//   itemb[q][e]  = temporal embedding of item q's graph;
//   ivec[q][m*32 + slot] = accumulator init of x_l (m = 0) / x_r (m = 1): b_m[ch] + W_m[ch, Cin:] . temb_q for a channel
//   slot, the att-weighted sum over the head for a u slot.
constexpr int SCR_WL = 0, SCR_WR = C * C, SCR_ATT = 2 * C * C, SCR_BL = SCR_ATT + 32, SCR_BR = SCR_BL + 32,
              SCR_UW = SCR_BR + 32, SCR_FLOATS = SCR_UW + 4 * 32;

__device__ __attribute__((noinline)) void block_prologue(const TecmSpatial* dp, int it0, int nit, float* smem, int scr,
                                                         int iv_off, int tb_off, bool tf_uniform) {
  const TecmSpatial& d = *dp;
  const int tid = threadIdx.x, Demb = d.Demb, Cin = d.Cin;
  for (int i = tid; i < C * C; i += THREADS) {
    smem[scr + SCR_WL + i] = d.Wl[i];
    smem[scr + SCR_WR + i] = d.Wr[i];
  }
  if (tid < C) {
    smem[scr + SCR_ATT + tid] = d.att[tid];
    smem[scr + SCR_BL + tid] = d.bl[tid];
    smem[scr + SCR_BR + tid] = d.br[tid];
  }
  if (tf_uniform) {
    for (int i = tid; i < nit * Demb; i += THREADS) {
      const int q = i / Demb, e = i - q * Demb;
      const Item it = decode_item(d, it0 + q);
      const TimeIdx ti = load_time_idx(d, it.b, it.t, 0);
      smem[tb_off + q * 32 + e] = temporal_emb(d, ti, e);
    }
  }
  __syncthreads();
  if (tid < 4 * C) {                                         // u weights: (matrix m, head h, input k)
    const int m = tid / (2 * C), h = (tid / C) & 1, k = tid % C;
    const float* W = smem + scr + (m ? SCR_WR : SCR_WL);
    float v = 0.f;
    for (int a = 0; a < CH; ++a) v = fmaf(smem[scr + SCR_ATT + h * CH + a], W[(h * CH + a) * C + k], v);
    smem[scr + SCR_UW + (2 * m + h) * 32 + k] = v;
  }
  for (int i = tid; i < nit * 2 * C; i += THREADS) {         // channel slots of the accumulator-init vectors
    const int q = i / (2 * C), r = i - q * 2 * C, m = r / C, a = r - m * C;
    const float* W = smem + scr + (m ? SCR_WR : SCR_WL) + a * C + Cin;
    float v = smem[scr + (m ? SCR_BR : SCR_BL) + a];
    if (tf_uniform)
      for (int e = 0; e < Demb; ++e) v = fmaf(W[e], smem[tb_off + q * 32 + e], v);
    smem[iv_off + q * 64 + m * 32 + slot_of(a)] = v;
  }
  __syncthreads();
  for (int i = tid; i < nit * 64; i += THREADS) {            // u slots and the padding of the accumulator-init vectors
    const int q = i >> 6, c = i & 63, m = c >> 5, sl = c & 31;
    if (sl == CH || sl == 2 * CH + 1) {
      const int h = sl == CH ? 0 : 1;
      float v = 0.f;
      for (int a = 0; a < CH; ++a) v = fmaf(smem[scr + SCR_ATT + h * CH + a], smem[iv_off + q * 64 + m * 32 + h * 12 + a], v);
      smem[i + iv_off] = v;
    } else if (sl >= CP) {
      smem[i + iv_off] = 0.f;
    }
  }
}

// Phase 2 for one head (compile time): online softmax over the in-edges + the implicit self loop, four slots per
// step with ONE rescale of the accumulators per step; logits pre-scaled by log2(e) so the exponentials are bare
// v_exp_f32.  The result (residual + aggregate + bias) overwrites the thread's own x_r slice.
template <int HH>
__device__ __forceinline__ void edge_phase(const float* __restrict__ xg, int Cin, float* smem, int xl_off, int xr_off,
                                           int hT_off, int P, const int* eptr, const int* ecol, const float* temb,
                                           int tn, int wi, bool use_edges, uint64_t dseed, uint64_t dbase,
                                           const float (&att4)[CH], const float (&bias)[CH], bool residual,
                                           bool tf_uniform, uint32_t dth, float dinv) {
#ifdef SPF_STAMPS
  unsigned long long estamp_ = __builtin_amdgcn_s_memtime();
#endif
  float xr[CH + 1];                                          // xr[11] = u_r
  {
    const float4* p = reinterpret_cast<const float4*>(smem + xr_off + tn * CP + HH * 12);
    const float4 a = p[0], b = p[1], c = p[2];
    xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w; xr[4] = b.x; xr[5] = b.y; xr[6] = b.z; xr[7] = b.w;
    xr[8] = c.x; xr[9] = c.y; xr[10] = c.z; xr[11] = c.w;
  }
  float hres[CH];                                            // residual input h[i]: issued now, consumed at the very end
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = HH * CH + c;                              // compile time; `ch < Cin` is a scalar branch
    float h = 0.f;
    if (residual) h = ch < Cin ? xg[ch] : smem[hT_off + ch * P + wi] + (tf_uniform ? temb[ch - Cin] : 0.f);
    hres[c] = h;
  }
  const int e0 = use_edges ? eptr[tn] : 0;
  const int deg = use_edges ? eptr[tn + 1] - e0 : 0;
  const float base = (0.6f * LOG2E) * xr[11];
  SPF_E(8);                                                  // x_r slice + degree
  float m = -INFINITY, z = 0.f;
  float acc[CH + 1];
#pragma unroll
  for (int c = 0; c <= CH; ++c) acc[c] = 0.f;
  for (int s = 0; s <= deg; s += 4) {                        // slot deg is the implicit self loop
    float a[4][CH + 1], ev[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sl = s + u;
      const int j = sl < deg ? ecol[e0 + sl] : wi;           // slots past the self loop re-read it and get weight 0
      const float4* p = reinterpret_cast<const float4*>(smem + xl_off + j * CP + HH * 12);
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
    const float mn = fmaxf(fmaxf(fmaxf(m, ev[0]), fmaxf(ev[1], ev[2])), ev[3]);   // slot s is always valid: mn is finite
    const float corr = __builtin_amdgcn_exp2f(m - mn);       // exp2(-inf) = 0 on the first step
    float pm[4], zs = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
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
      for (int u = 0; u < 4; ++u) t = f32x2{a[u][c], a[u][c + 1]} * pm[u] + t;
      acc[c] = t.x;
      acc[c + 1] = t.y;
    }
  }
  SPF_E(9);                                                  // edge loop
  const float inv = 1.0f / (z + 1e-16f);
  float* o = smem + xr_off + tn * CP + HH * 12;              // this thread's x_r slice is dead: it becomes the output
#pragma unroll
  for (int c = 0; c < CH; ++c) o[c] = hres[c] + (acc[c] * inv + bias[c]);
  SPF_E(10);                                                 // residual + store of the slice
}

__global__ __launch_bounds__(THREADS, 2) void spatial_fwd_kernel(const TecmSpatial d, const int total) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave-uniform by construction: keep it scalar
  const int sg = wave >> 2;                                  // this wave's slot
  const int st = tid & (SLOT_T - 1);                         // thread inside the slot
  const int Cin = d.Cin, Demb = d.Demb, L = d.L, N = d.N;
  const int P = d.win_max | 1;                               // odd pitch of the k-major window image
  const int wm4 = (d.win_max + 3) & ~3;
  // LDS map (float offsets into smem; integer offsets keep every access in the LDS address space).  Per slot:
  //   hT [C][P]  [x | node_emb (+ temb)]^T of the window;  x_l [wm4][CP] (head-sliced rows);  x_r [tile_nodes][CP]
  //   (the tile's x_r, later its output).  Shared: per-item accumulator-init vectors, temporal embeddings, CSR slice.
  const int hT_sz = (C * P + 3) & ~3;
  const int slot_floats = hT_sz + wm4 * CP + d.tile_nodes * CP;
  const int hT_off = sg * slot_floats;
  const int xl_off = hT_off + hT_sz;
  const int xr_off = xl_off + wm4 * CP;
  const int iv_off = 2 * slot_floats;                        // [MAXI][64]  per item: accumulator init of x_l (+0) / x_r (+32)
  const int tb_off = iv_off + MAXI * 64;                     // [MAXI][32]  per item: temporal embedding
  int* eptr = reinterpret_cast<int*>(smem + tb_off + MAXI * 32);        // [tile_nodes + 1]   CSR slice of the tile
  int* ecol = eptr + d.tile_nodes + 1;                       // [tile_edges_max]   window-relative sources

  const int nblk = gridDim.x, blk = blockIdx.x;
  const int it0 = (int)((int64_t)blk * total / nblk);
  const int nit = (int)((int64_t)(blk + 1) * total / nblk) - it0;      // <= MAXI (host)
  if (nit <= 0) return;
  const bool tf_uniform = Demb > 0 && d.tf_sn == 0;          // time features constant over the nodes (train.py:65)
  const bool residual = !(d.flags & TECM_SPATIAL_NO_RESIDUAL);

  // ---- block prologue (block_prologue above), then the MFMA B operands into registers: lane (c31, kq) holds
  //      column `c31` of the slot-ordered W^T for k = 2s + kq; the two u slots carry att . W, so that u = att . x_l
  //      falls out of the same MFMA.
  const TecmSpatial* dp = kernarg_desc();                    // `d` again, as a pointer for the out-of-line helpers
  const int scr = tb_off + MAXI * 32 + d.tile_nodes + 1 + d.tile_edges_max;   // prologue scratch, behind the CSR slice
  block_prologue(dp, it0, nit, smem, scr, iv_off, tb_off, tf_uniform);
  __syncthreads();
  const int c31 = lane & 31, kq = lane >> 5;
  float bwl[C / 2], bwr[C / 2];
  {
    const int ch = chan_of(c31);
    const bool is_u = c31 == CH || c31 == 2 * CH + 1;
    const int ol = ch >= 0 ? SCR_WL + ch * C : SCR_UW + (c31 == CH ? 0 : 1) * 32;
    const int orr = ch >= 0 ? SCR_WR + ch * C : SCR_UW + (c31 == CH ? 2 : 3) * 32;
#pragma unroll
    for (int s = 0; s < C / 2; ++s) {
      const int k = 2 * s + kq;
      bwl[s] = (ch >= 0 || is_u) ? smem[scr + ol + k] : 0.f;
      bwr[s] = (ch >= 0 || is_u) ? smem[scr + orr + k] : 0.f;
    }
  }
  const int tn = st & 127;                                   // edge phase: (tile node, head); the head is wave-uniform
  const int hh = (wave >> 1) & 1;
  float att4[CH], bias[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    att4[c] = (0.4f * LOG2E) * smem[scr + SCR_ATT + hh * CH + c];
    bias[c] = d.bias[hh * CH + c];
  }

  const uint32_t dth = d.alpha_drop.p > 0.f ? tecm_drop_thresh(d.alpha_drop.p) : 0u;
  const float dinv = d.alpha_drop.p > 0.f ? 1.0f / (1.0f - d.alpha_drop.p) : 1.0f;
  const uint64_t dseed_now = tecm_seed_now(d.alpha_drop.seed, d.alpha_drop.seed_dev);
  const bool even_cin = (Cin & 1) == 0;
  const unsigned cin_magic = (unsigned)((0x100000000ull + Cin - 1) / Cin);   // idx / Cin == umulhi(idx, magic) for idx * Cin < 2^32

  int cur_tile = -1, n0 = 0, n1 = 0, lo = 0, hi = 0;
  bool x_ready = false;                                      // per slot (uniform over the slot's waves)
  float2 pf[NPF];
#ifdef SPF_STAMPS
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
  SPF_T(0);                                                  // prologue

  int q = 0;
  while (q < nit) {
    const Item itA = decode_item(d, it0 + q);
    if (itA.tile != cur_tile) {
      // ---- tile switch (once or twice per block): CSR slice, static node-embedding rows of the window
      __syncthreads();
      cur_tile = itA.tile;
      const TileState ts = tile_switch(dp, itA.tile, smem, slot_floats, P, eptr, ecol, tf_uniform);
      // the helper is out of line, so its results come back in vector registers: make them provably uniform again
      n0 = __builtin_amdgcn_readfirstlane(ts.n0);
      n1 = __builtin_amdgcn_readfirstlane(ts.n1);
      lo = __builtin_amdgcn_readfirstlane(ts.lo);
      hi = __builtin_amdgcn_readfirstlane(ts.hi);
      x_ready = false;
    }
    // slot 0 takes item q; slot 1 takes item q + 1 when that one shares the tile
    const Item itB = decode_item(d, it0 + min(q + 1, nit - 1));
    const bool two = q + 1 < nit && itB.tile == cur_tile;
    const int np = two ? 2 : 1;
    const bool act = sg == 0 || two;                         // slot 1 idles through a lone item (barriers only)
    const Item it = sg == 0 ? itA : itB;
    const int qi = q + (two ? sg : 0);                       // this slot's item inside the block's range
    const int ta = n0 - lo, tb = n1 - lo;                    // tile rows inside the window
    const int waA = itA.use_edges ? 0 : ta, wbA = itA.use_edges ? hi - lo : tb;
    const int waB = itB.use_edges ? 0 : ta, wbB = itB.use_edges ? hi - lo : tb;
    const int wa = sg == 0 ? waA : waB, wb = sg == 0 ? wbA : wbB;
    const int64_t grow = ((int64_t)it.b * L + it.t) * N;     // first row of this graph in the (B, L, N, *) tensors

    if (act && !x_ready) {
      // x rows of the window -> hT[k][w], k < Cin (first item of the block, after a tile switch, or no prefetch)
      const float* xb = d.x + (grow + lo + wa) * Cin;
      const int cnt = (wb - wa) * Cin;
      if (even_cin) {
        const float2* xb2 = reinterpret_cast<const float2*>(xb);
        for (int f = st; f < (cnt >> 1); f += SLOT_T) {
          const float2 v = xb2[f];
          const int idx = 2 * f, row = (int)__umulhi((unsigned)idx, cin_magic), k = idx - row * Cin;
          smem[hT_off + k * P + wa + row] = v.x;
          smem[hT_off + (k + 1) * P + wa + row] = v.y;
        }
      } else {
        for (int f = st; f < cnt; f += SLOT_T) {
          const int row = f / Cin, k = f - row * Cin;
          smem[hT_off + k * P + wa + row] = xb[f];
        }
      }
    }
    if (act && Demb > 0 && !tf_uniform) stage_emb_rows(dp, it.b, it.t, lo, wa, wb, smem, hT_off, P);
    __syncthreads();                                         // hT complete; the previous out tiles have been stored
    SPF_T(1);                                                // staging (+ tile switch)

    // ---- prefetch the x rows of this slot's next item (registers), consumed after the dense phase
    bool pf_ok = false;
    int pf_cnt2 = 0, pf_wa = 0;
    {
      const int nq = q + np + sg;                            // slot 1 only ever pairs with an item of the same tile
      if (nq < nit && even_cin && !(SPF_SKIP & 4)) {
        const Item nx = decode_item(d, it0 + nq);
        if (nx.tile == cur_tile) {
          pf_wa = nx.use_edges ? 0 : ta;
          const int nwb = nx.use_edges ? hi - lo : tb;
          pf_cnt2 = ((nwb - pf_wa) * Cin) >> 1;
          if (pf_cnt2 <= NPF * SLOT_T) {
            pf_ok = true;
            const float2* xb2 =
                reinterpret_cast<const float2*>(d.x + (((int64_t)nx.b * L + nx.t) * N + lo + pf_wa) * Cin);
#pragma unroll
            for (int r = 0; r < NPF; ++r) {
              const int f = st + r * SLOT_T;
              if (f < pf_cnt2) pf[r] = xb2[f];
            }
          }
        }
      }
    }

    // ---- phase 1: x_l for the window rows, x_r for the tile rows of BOTH slots' items; 32-row blocks dealt to the
    //      eight waves.  Everything about a task is wave-uniform (scalar registers, scalar branches).
    if (!(SPF_SKIP & 1)) {
      const int nlA = (wbA - waA + 31) >> 5, nr = (tb - ta + 31) >> 5;
      const int nlB = two ? (wbB - waB + 31) >> 5 : 0;
      const int TA = nlA + nr, TT = TA + (two ? nlB + nr : 0);
      for (int task0 = wave; task0 < TT; task0 += THREADS / 64) {
        const int tsl = task0 >= TA ? 1 : 0;                 // the task's slot
        const int task = task0 - tsl * TA;
        const int nl = tsl ? nlB : nlA, twa = tsl ? waB : waA, twb = tsl ? wbB : wbA;
        const bool isr = task >= nl;
        const int r0 = isr ? ta + 32 * (task - nl) : twa + 32 * task;
        const int rend = isr ? tb : twb;
        const int sb = tsl * slot_floats;                    // the slot's LDS base
        f32x16 acc = splat16(smem[iv_off + (q + tsl) * 64 + (isr ? 32 : 0) + c31]);
        const int row = min(r0 + c31, rend - 1);             // clamped rows feed accumulator rows that are never stored
        const int ao = sb + kq * P + row;
        float av[C / 2];                                     // all A operands in flight before the first MFMA needs one
#pragma unroll
        for (int s = 0; s < C / 2; ++s) av[s] = smem[ao + 2 * s * P];
        if (isr) {
#pragma unroll
          for (int s = 0; s < ((SPF_SKIP & 8) ? 5 : C / 2); ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bwr[s], acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int s = 0; s < ((SPF_SKIP & 8) ? 5 : C / 2); ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bwl[s], acc, 0, 0, 0);
        }
        // accumulator register e holds row r0 + (e & 3) + 8 (e >> 2) + 4 kq of column c31
        const int oo = sb + hT_sz + (isr ? wm4 * CP - ta * CP : 0) + (r0 + 4 * kq) * CP + c31;
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
    SPF_T(5);                                                // own dense tasks (thread 0 = wave 0: the most tasks)
    __syncthreads();                                         // x_l, x_r complete; the x rows of hT are free
    SPF_T(2);                                                // dense phase: wait for the other waves

    if (pf_ok) {
#pragma unroll
      for (int r = 0; r < NPF; ++r) {
        const int f = st + r * SLOT_T;
        if (f < pf_cnt2) {
          const int idx = 2 * f, row = (int)__umulhi((unsigned)idx, cin_magic), k = idx - row * Cin;
          smem[hT_off + k * P + pf_wa + row] = pf[r].x;
          smem[hT_off + (k + 1) * P + pf_wa + row] = pf[r].y;
        }
      }
    }
    x_ready = pf_ok;
    SPF_T(6);                                                // prefetch commit

    // ---- phase 2: one thread per (target node, head) of the slot's item
    {
      const int i = n0 + tn;
      if (act && tn < d.tile_nodes && i < n1 && !(SPF_SKIP & 2)) {
        const int64_t rowi = (int64_t)(it.t * d.B + it.b) * N + i;          // row in the reference's (L*B*N) flattening
        const float* xg = d.x + (grow + i) * Cin;
        const float* temb = smem + tb_off + qi * 32;
        if (hh == 0)
          edge_phase<0>(xg, Cin, smem, xl_off, xr_off, hT_off, P, eptr, ecol, temb, tn, i - lo, it.use_edges,
                        dseed_now, (uint64_t)((rowi * H + 0) * d.alpha_drop.ld), att4, bias, residual,
                        tf_uniform, dth, dinv);
        else
          edge_phase<1>(xg, Cin, smem, xl_off, xr_off, hT_off, P, eptr, ecol, temb, tn, i - lo, it.use_edges,
                        dseed_now, (uint64_t)((rowi * H + 1) * d.alpha_drop.ld), att4, bias, residual,
                        tf_uniform, dth, dinv);
      }
    }
    SPF_T(7);                                                // own edge work
    __syncthreads();                                         // output tiles complete (head-sliced rows)
    SPF_T(3);                                                // edge phase: wait for the other waves

    // ---- phase 3: the tile leaves in contiguous 16-byte stores; LDS slot = channel + (channel >= 11), columns 22
    //      and 23 of every output row are the zero padding
    if (act) {
      if (d.out_ld == CP) {
        const int nf4 = (n1 - n0) * (CP / 4);
        float4* dst = reinterpret_cast<float4*>(d.out + (grow + n0) * CP);
        for (int f = st; f < nf4; f += SLOT_T) {
          const int r = f / (CP / 4), c = 4 * (f - r * (CP / 4));
          const float* src = smem + xr_off + r * CP;
          float4 v;
          v.x = src[slot_of(c)];
          v.y = src[slot_of(c + 1)];
          v.z = c + 2 < C ? src[slot_of(c + 2)] : 0.f;
          v.w = c + 3 < C ? src[slot_of(c + 3)] : 0.f;
          dst[f] = v;
        }
      } else {
        float* dst = d.out + (grow + n0) * (int64_t)d.out_ld;
        for (int f = st; f < (n1 - n0) * C; f += SLOT_T) {
          const int r = f / C, c = f - r * C;
          dst[(int64_t)r * d.out_ld + c] = smem[xr_off + r * CP + slot_of(c)];
        }
      }
    }
    SPF_T(4);                                                // store
    q += np;
  }
}

// Stand-alone SpatioTemporalEmbedding.forward (modules.py:230-266): out (B, L, N, Cin + Demb) = cat([x, emb]).
// A bandwidth kernel of its own: nothing to stage, one thread per (row, channel).
__global__ __launch_bounds__(256) void embed_only_kernel(const TecmSpatial d) {
  const int64_t total = (int64_t)d.B * d.L * d.N * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / C;
    const int ch = (int)(i - m * C);
    float v;
    if (ch < d.Cin) {
      v = d.x[m * d.Cin + ch];
    } else {
      const int n = (int)(m % d.N);
      const int64_t g = m / d.N;
      const int b = (int)(g / d.L), t = (int)(g - (int64_t)b * d.L);
      const TimeIdx ti = load_time_idx(d, b, t, n);
      v = d.node_tab[(int64_t)n * d.Demb + ch - d.Cin] + temporal_emb(d, ti, ch - d.Cin);
    }
    d.out[m * d.out_ld + ch] = v;
  }
}

}  // namespace

#ifdef SPF_STAMPS
extern "C" int tecm_debug_spf_stamps(unsigned long long* out16, int reset) {
  if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_spf_stamps), sizeof(g_spf_stamps));
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_spf_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

size_t tecm_spatial_fwd_lds(const TecmSpatial& d) {
  const int P = d.win_max | 1, wm4 = (d.win_max + 3) & ~3;
  return sizeof(float) * (2 * ((size_t)((C * P + 3) & ~3) + (size_t)wm4 * CP + (size_t)d.tile_nodes * CP) + MAXI * 96 +
                          d.tile_nodes + 1 + d.tile_edges_max + SCR_FLOATS);
}

extern "C" int tecm_spatial_fwd(const TecmSpatial* dp, void* stream) {
  TECM_REQUIRE(dp != nullptr, TECM_E_ARG, "tecm_spatial_fwd: null descriptor");
  const TecmSpatial& d = *dp;
  const int rc = check_common("tecm_spatial_fwd", d);
  if (rc) return rc;
  TECM_REQUIRE(d.out != nullptr && d.out_ld >= C, TECM_E_ARG, "tecm_spatial_fwd: out / out_ld");
  TECM_REQUIRE(tecm_aligned(d.x, 8), TECM_E_ALIGN, "tecm_spatial_fwd: x must be 8-byte aligned");
  if (d.flags & TECM_SPATIAL_EMBED_ONLY) {
    TECM_REQUIRE(d.Demb > 0, TECM_E_ARG, "tecm_spatial_fwd: embed-only mode needs embedding tables");
    const int64_t total = (int64_t)d.B * d.L * d.N * C;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(embed_only_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d);
    TECM_CHECK_LAUNCH("tecm_spatial_fwd(embed only)");
    return TECM_OK;
  }
  TECM_REQUIRE(d.out_ld != CP || tecm_aligned(d.out, 16), TECM_E_ALIGN, "tecm_spatial_fwd: out must be 16-byte aligned");
  const size_t lds = tecm_spatial_fwd_lds(d);
  TECM_REQUIRE(lds <= (size_t)kLdsBudget, TECM_E_LDS,
               "tecm_spatial_fwd: neighbour window of %d rows needs %zu B of LDS (> 160 KiB); renumber the graph "
               "(e.g. RCM) or shrink tile_nodes", d.win_max, lds);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        kLdsBudget);
    attr_set = true;
  }
  // contiguous item ranges: one 512-thread block per CU (it works on two items at a time), never more than MAXI items
  // a block; small problems still spread over the chip two items a block
  const int64_t total = (int64_t)d.B * d.L * d.num_tiles;
  int64_t nblk = (total + 1) / 2 < 256 ? (total + 1) / 2 : 256;
  if ((total + nblk - 1) / nblk > MAXI) nblk = (total + MAXI - 1) / MAXI;
  hipLaunchKernelGGL(spatial_fwd_kernel, dim3((unsigned)nblk), dim3(THREADS), lds, (hipStream_t)stream, d, (int)total);
  TECM_CHECK_LAUNCH("tecm_spatial_fwd");
  return TECM_OK;
}
