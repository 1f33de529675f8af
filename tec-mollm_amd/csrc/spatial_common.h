// Shared pieces of the fused spatial stage a-1..a-3 (SpatioTemporalEmbedding modules.py:230-266 + GATv2Conv
// modules.py:329-336,:356 + residual tec_mollm.py:94): constants, time-index handling, small LDS row helpers.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace tecm_spatial {

constexpr float NEG_SLOPE = 0.2f;
constexpr int C = 22;      // feature channels (Cin + Demb)
constexpr int H = 2;       // heads
constexpr int CH = C / H;  // 11 channels per head
constexpr int CP = 24;     // padded row: 16-byte aligned
constexpr int kLdsBudget = 160 * 1024;

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v >= hi ? hi - 1 : v); }

struct TimeIdx {
  int tod, doy, year, season;
  int bad;      // TECM_BAD_* bits: an index outside its table (the reference's nn.Embedding raises, modules.py:255-258)
};
__device__ __forceinline__ TimeIdx load_time_idx(const TecmSpatial& d, int b, int t, int node) {
  const float* p = d.tf + (int64_t)b * d.tf_sb + (int64_t)t * d.tf_sl + (int64_t)node * d.tf_sn;
  TimeIdx ti;
  const int tod = (int)p[0], doy = (int)p[d.tf_sf];     // .long() truncation, modules.py:250-253
  const int year = (int)p[2 * d.tf_sf], season = (int)p[3 * d.tf_sf];
  ti.bad = ((unsigned)tod >= 12u ? TECM_BAD_TOD : 0) | ((unsigned)doy >= 366u ? TECM_BAD_DOY : 0) |
           ((unsigned)year >= (unsigned)d.year_rows ? TECM_BAD_YEAR : 0) | ((unsigned)season >= 4u ? TECM_BAD_SEASON : 0);
  // clamped only so that the table reads below stay inside their allocations; a bad index never yields a value
  ti.tod = clampi(tod, 12);
  ti.doy = clampi(doy, 366);
  ti.year = clampi(year, d.year_rows);
  ti.season = clampi(season, 4);
  return ti;
}
// ((tod + doy) + year) + season -- the exact association of modules.py:260.  An out-of-range index is reported through
// the device error word and turns the embedding into NaN: it is rejected, not repaired.
__device__ __forceinline__ float temporal_emb(const TecmSpatial& d, const TimeIdx& ti, int k) {
  const int D = d.Demb;
  if (ti.bad) {
    atomicOr(d.err_flag, ti.bad);
    return __builtin_nanf("");
  }
  return ((d.tod_tab[ti.tod * D + k] + d.doy_tab[ti.doy * D + k]) + d.year_tab[ti.year * D + k]) +
         d.season_tab[ti.season * D + k];
}

__device__ __forceinline__ float lrelu(float s) { return s > 0.f ? s : NEG_SLOPE * s; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float LOG2E = 1.4426950408889634f;

// LDS rows of x_l / x_r / dout (24 floats) are laid out per HEAD so that a (node, head) thread reads exactly three
// aligned float4:   [ head 0: channels 0..10 | u0 | head 1: channels 11..21 | u1 ]
// where, for x_l / x_r, u_h = sum_c att[h][c] * row[h][c] comes out of the same MFMA as two extra weight columns.
// With lrelu(s) = 0.6 s + 0.4 |s| the GATv2 logit of edge j -> i is
//      e = 0.6 (u_l[j] + u_r[i]) + 0.4 sum_c att_c |x_l[j,c] + x_r[i,c]|
// -- two VALU operations per channel (add, fma with |.| modifier) instead of four.
__device__ __forceinline__ int slot_of(int ch) { return ch + (ch >= CH ? 1 : 0); }
// channel of an LDS slot: -1 for the two u slots and the padding
__device__ __forceinline__ int chan_of(int sl) { return sl < CH ? sl : (sl == CH ? -1 : (sl < 2 * CH + 1 ? sl - 1 : -1)); }

// The descriptor is the first kernel argument: rarely executed paths (tile switch, per-node time features, block
// prologue) read it through the kernarg segment inside NOINLINE helpers, so that their two dozen pointers are not kept
// in scalar registers (or spilled to VGPR lanes) across the per-item phases.  Call from the KERNEL body only and hand
// the pointer down.
__device__ __forceinline__ const char* kernarg_base() {
  return (const char*)__builtin_amdgcn_kernarg_segment_ptr();             // C cast: drops the constant address space
}

// the CH = 11 channels of head hh out of a 24-float row (16-byte aligned): head 0 = floats 0..10, head 1 = 11..21
__device__ __forceinline__ void load_head(const float* row, int hh, float (&v)[CH]) {
  static_assert(CH == 11, "head slicing is written for 11 channels per head");
  const float4* p = reinterpret_cast<const float4*>(row);
  if (hh == 0) {
    const float4 a = p[0], b = p[1], c = p[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z;
  } else {
    const float4 a = p[2], b = p[3], c = p[4], e = p[5];
    v[0] = a.w; v[1] = b.x; v[2] = b.y; v[3] = b.z; v[4] = b.w; v[5] = c.x; v[6] = c.y; v[7] = c.z;
    v[8] = c.w; v[9] = e.x; v[10] = e.y;
  }
}
// store the 11 channels of head hh into a 24-float row; only this head's floats are written (scalar stores: the two
// heads of a row are written by different threads and share the float4 that holds floats 8..11)
__device__ __forceinline__ void store_head(float* row, int hh, const float (&v)[CH]) {
  float* p = row + hh * CH;
#pragma unroll
  for (int c = 0; c < CH; ++c) p[c] = v[c];
}

// One work item = (tile of target nodes, graph (b, t)).  Items are numbered tile-major so that a block's contiguous
// range keeps its tile (CSR slice, node-embedding rows) across most of its items.
struct Item {
  int tile, b, t;
  bool use_edges;
};
__device__ __forceinline__ Item decode_item(const TecmSpatial& d, int item) {
  const int G = d.B * d.L;
  Item it;
  it.tile = item / G;
  const int gm = item - it.tile * G;           // memory order of the (B, L, N, *) tensors: gm = b*L + t
  it.b = gm / d.L;
  it.t = gm - it.b * d.L;
  it.use_edges = (it.t * d.B + it.b) < d.graphs_with_edges;   // the reference's flattening is (L*B): g = t*B + b
  return it;
}

inline int check_common(const char* who, const TecmSpatial& d) {
  TECM_REQUIRE(d.B > 0 && d.L > 0 && d.N > 0 && d.Cin > 0 && d.Demb >= 0 && d.H > 0, TECM_E_ARG, "%s: bad shape", who);
  TECM_REQUIRE(d.Cin + d.Demb == C && d.H == H, TECM_E_ARG,
               "%s: built for C = Cin + Demb = 22 channels and 2 heads (got C=%d H=%d)", who, d.Cin + d.Demb, d.H);
  TECM_REQUIRE(d.x && d.Wl && d.bl && d.Wr && d.br && d.att && d.bias && d.rowptr && d.colidx && d.tile_lo &&
                   d.tile_hi && d.err_flag,
               TECM_E_ARG, "%s: null pointer", who);
  TECM_REQUIRE(d.Demb == 0 || (d.tf && d.node_tab && d.tod_tab && d.doy_tab && d.year_tab && d.season_tab), TECM_E_ARG,
               "%s: null embedding table / time features", who);
  TECM_REQUIRE(d.num_tiles > 0 && d.tile_nodes > 0 && d.tile_nodes <= 128 &&
                   (int64_t)d.num_tiles * d.tile_nodes >= d.N && d.win_max >= 1,
               TECM_E_ARG, "%s: bad node tiling (tile_nodes must be <= 128: two threads per target node)", who);
  TECM_REQUIRE(d.Demb == 0 || d.year_rows > 0, TECM_E_ARG, "%s: year_rows must be positive", who);
  TECM_REQUIRE(d.tile_edges_max >= 0, TECM_E_ARG, "%s: tile_edges_max must be the max edge count of a tile", who);
  TECM_REQUIRE((int64_t)d.B * d.L * d.num_tiles < (int64_t)1 << 30, TECM_E_ARG, "%s: too many (tile, graph) items", who);
  return TECM_OK;
}

}  // namespace tecm_spatial
