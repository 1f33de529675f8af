// Pieces shared by the second formulation of the fused spatial stage, forward (spatial_fwd2.hip) and backward
// (spatial_bwd2.hip): the workspace layout, the set-up kernel that fills it and the per-row transform.
//     x_m[g, n] = A_m x[g, n] + P_m[n] + c_m(g),   A_m = W_m[:, :Cin],  P_m[n] = W_m[:, Cin:] node_emb[n],
//                                                  c_m(g) = W_m[:, Cin:] temb_g + b_m
#pragma once
#include "spatial_common.h"

namespace tecm_spatial2 {
using namespace tecm_spatial;

constexpr int T2 = 256;          // threads of a block = rows of the largest neighbour window
constexpr int TN2 = 128;         // largest tile
constexpr int GV = 64;           // floats per graph in the workspace: c_l (24) | c_r (24) | temb (16)
#ifndef SP2_EU
#define SP2_EU 3
#endif
constexpr int EU = SP2_EU;       // edge slots per online-softmax step.  The <= 150 km grid has 8 neighbours in its interior,
                                 // 5 on an edge, 10 at high latitudes: with the self loop 9 / 6 / 11 slots -- steps of three
                                 // waste 0 / 0 / 1 slot where steps of four (the first kernel) waste 3 / 2 / 1

// workspace layout (floats): A[2][24][16] | P[2][N][24] | per-graph vectors [G][64]
__host__ __device__ inline int64_t ws_A() { return 0; }
__host__ __device__ inline int64_t ws_P(int) { return 2 * 24 * 16; }
__host__ __device__ inline int64_t ws_G(int N) { return 2 * 24 * 16 + (int64_t)2 * N * CP; }

// row `s` (LDS slot order: head 0 channels | u0 | head 1 channels | u1) of transform m applied to basis vector k of h
__device__ __forceinline__ float ext_weight(const TecmSpatial&, const float* W, int s, int k) {
  const int ch = chan_of(s);
  return ch >= 0 ? W[ch * C + k] : 0.f;
}

// ---- set-up: one thread per (row, m, slot) for the N node rows of P_m and the G per-graph vectors c_m (a channel slot is
//      one Demb-long dot product; the u slots stay zero), 16 more threads per graph for its temporal embedding, and
//      2 x 24 x 16 threads for the Cin-wide maps A_m in slot order
static __global__ __launch_bounds__(256) void spatial_prep_kernel(const TecmSpatial d, float* __restrict__ ws) {
  const int N = d.N, Cin = d.Cin, Demb = d.Demb, G = d.B * d.L;
  const int nA = 2 * 24 * 16, nR = (N + G) * 48, nT = G * 16;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nA) {
    const int m = i / (24 * 16), r = i % (24 * 16), sl = r / 16, k = r % 16;
    ws[ws_A() + i] = k < Cin ? ext_weight(d, m ? d.Wr : d.Wl, sl, k) : 0.f;
    return;
  }
  int j = i - nA;
  if (j >= nR + nT) return;
  if (j >= nR) {                                             // temporal embedding of graph gm, element e
    j -= nR;
    const int gm = j >> 4, e = j & 15, b = gm / d.L, t = gm - b * d.L;
    ws[ws_G(N) + (int64_t)gm * GV + 48 + e] = e < Demb ? temporal_emb(d, load_time_idx(d, b, t, 0), e) : 0.f;
    return;
  }
  const int row = j / 48, ms = j - row * 48, m = ms / 24, sl = ms - m * 24;
  const float* W = m ? d.Wr : d.Wl;
  const float* bv = m ? d.br : d.bl;
  const bool node_row = row < N;
  const int gm = row - N;                                    // gm = b*L + t (memory order of the (B, L, N, *) tensors)
  float emb[16];
  if (node_row) {
#pragma unroll
    for (int e = 0; e < 16; ++e) emb[e] = e < Demb ? d.node_tab[(int64_t)row * Demb + e] : 0.f;
  } else {
    const int b = gm / d.L, t = gm - b * d.L;
    const TimeIdx ti = load_time_idx(d, b, t, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) emb[e] = e < Demb ? temporal_emb(d, ti, e) : 0.f;   // NaN + error word on a bad index
  }
  auto chan = [&](int ch) {
    float v = node_row ? 0.f : bv[ch];
    for (int e = 0; e < Demb; ++e) v = fmaf(W[ch * C + Cin + e], emb[e], v);
    return v;
  };
  const int ch = chan_of(sl);
  const float v = ch >= 0 ? chan(ch) : 0.f;                  // the u slots are formed in the main kernel, from the finished row
  float* dst = node_row ? ws + ws_P(N) + ((int64_t)m * N + row) * CP : ws + ws_G(N) + (int64_t)gm * GV + m * 24;
  dst[sl] = v;
}

// One row of x_m in LDS slot order [head 0: 11 channels | u0 | head 1: 11 channels | u1]: the 22 channel slots are
// A[slot][16] . x + P[n][slot] + c[slot] (A, c and att are uniform: scalar loads feeding v_pk_fma_f32 as scalar operands;
// P is this thread's row), the two u slots are att_h . (the head's channels).
template <int CIN>
__device__ __forceinline__ void transform_row(const float* __restrict__ A, const float* __restrict__ Prow,
                                              const float* __restrict__ cvec, const float* __restrict__ att,
                                              const float (&x)[CIN], float* dst) {
  float o[CP];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const float4 p = reinterpret_cast<const float4*>(Prow)[q];
    o[4 * q] = p.x + cvec[4 * q]; o[4 * q + 1] = p.y + cvec[4 * q + 1];
    o[4 * q + 2] = p.z + cvec[4 * q + 2]; o[4 * q + 3] = p.w + cvec[4 * q + 3];
  }
#pragma unroll
  for (int sl = 0; sl < CP; ++sl) {
    if (sl == CH || sl == 2 * CH + 1) continue;              // u slots: below
#pragma unroll
    for (int k = 0; k < CIN; ++k) o[sl] = fmaf(A[sl * 16 + k], x[k], o[sl]);
  }
  float u0 = 0.f, u1 = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    u0 = fmaf(att[c], o[c], u0);
    u1 = fmaf(att[CH + c], o[CH + 1 + c], u1);
  }
  o[CH] = u0;
  o[2 * CH + 1] = u1;
#pragma unroll
  for (int q = 0; q < 6; ++q) reinterpret_cast<float4*>(dst)[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}


inline bool v2_eligible(const TecmSpatial& d) {
  return d.flags == 0 && d.Demb > 0 && d.Demb <= 16 && d.tf != nullptr && d.tf_sn == 0 && d.out_ld == CP &&
         (d.Cin == 10 || d.Cin == 6) && d.win_max <= T2 && d.tile_nodes <= TN2;
}
inline int64_t ws_floats(const TecmSpatial& d) { return ws_G(d.N) + (int64_t)d.B * d.L * GV; }
inline int prep_threads(const TecmSpatial& d) { return 2 * 24 * 16 + (d.N + d.B * d.L) * 48 + d.B * d.L * 16; }

}  // namespace tecm_spatial2
