/*
 * tecmollm.h -- C ABI of libtecmollm_hip.so, the MI355X (gfx950) implementation of the
 * TEC-MoLLM forward/backward hot path.
 *
 * The reference (PANXIONG-CN/TEC-MoLLM) is 100 % Python and has no FFI of its own: its
 * hot path is `TEC_MoLLM.forward` (src/model/tec_mollm.py:59-125) calling torch /
 * torch_geometric / transformers / peft ops.  Each entry point below therefore cites the
 * reference *operation* it replaces; the Python-side mirror of the reference's module API
 * (tec-mollm_amd/src/model/{tec_mollm,modules}.py) binds these through ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless said otherwise;
 *   - the library never allocates or frees caller-visible memory (workspaces are passed in);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), no internal sync;
 *   - return value: 0 = launched, negative = TECM_E_* (nothing launched);
 *   - activation tensors after the spatial stage are "time-major": rows m = (b*T + t)*N + n,
 *     i.e. logical shape (B, T, N, C) with C contiguous -- see DESIGN.md "Data layout";
 *   - dropout masks are a pure function keep(seed, idx) (tecm_keep in csrc/common.h), so the
 *     backward pass recomputes them; idx = logical_row * ld + col of the tensor being dropped.
 */
#ifndef TECMOLLM_H
#define TECMOLLM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TECM_OK 0
#define TECM_E_ARG (-1)        /* bad shape / null pointer / unsupported size            */
#define TECM_E_ALIGN (-2)      /* pointer or leading dimension not aligned as required   */
#define TECM_E_LAUNCH (-3)     /* hipGetLastError() != hipSuccess after the launch       */
#define TECM_E_LDS (-4)        /* problem does not fit the 160 KiB LDS budget            */

#define TECM_ABI_VERSION 17
int tecm_abi_version(void);
/* Human-readable text for the last error on this thread (host pointer, never NULL). */
const char* tecm_last_error(void);

/* ------------------------------------------------------------------ temporal-window row view
 * A row view maps a logical row m = (bq*Lout + t_out)*N + n and an inner index kk = tap*Cw + c
 * onto a (B, Lin, N, ld) time-major tensor:
 *      t_in = t_out*stride_t + tap - pad ;  valid iff 0 <= t_in < Lin  (else the element is 0)
 *      addr = base + ((bq*Lin + t_in)*N + n)*ld + c
 * enabled == 0 means the plain view addr = base + m*ld + kk.
 * It expresses, without materialising anything: Conv1d im2col over time (modules.py:27),
 * the stride-2 1x1 conv (modules.py:36-41), einops 'b (p l) d -> b p (l d)' (modules.py:114)
 * and PredictionHead's view(batch, -1) (modules.py:307). */
typedef struct TecmWin {
  int32_t enabled;
  int32_t N;
  int32_t Lin, Lout;
  int32_t stride_t;
  int32_t taps;
  int32_t Cw;
  int32_t pad;
} TecmWin;

typedef struct TecmDrop {      /* keep-mask spec; p == 0 disables */
  float p;
  int32_t _pad;
  uint64_t seed;
  int64_t ld;                  /* idx = row*ld + col */
  /* NULL, or a device word ADDED to `seed` by the kernel when it starts: a training step whose launches were recorded
   * once (hipGraph) draws fresh masks at every replay by advancing that one word (tecm_seed_advance) -- the seeds in the
   * recorded kernel arguments never change.  The forward and the backward of a step read the same value. */
  const uint64_t* seed_dev;
} TecmDrop;

enum { TECM_A_MK = 0, TECM_A_KM = 1 };   /* A stored [m][k] (k contiguous) or [k][m] (m contiguous) */
enum { TECM_B_NK = 0, TECM_B_KN = 1 };   /* B stored [n][k] (k contiguous) or [k][n] (n contiguous) */
enum { TECM_ACT_NONE = 0, TECM_ACT_GELU_ERF = 1, TECM_ACT_GELU_TANH = 2 };

/* fp32 GEMM on exact-f32 MFMA (v_mfma_f32_32x32x2_f32):  C = epilogue(alpha * A_view . B_view).
 * Replaces every dense contraction on the path: nn.Conv1d (modules.py:27,36), nn.Linear
 * (modules.py:98,287,290), transformers Conv1D c_attn/c_proj/c_fc (pytorch_utils.py:95-121 via
 * modules.py:208), peft LoRA A/B (modules.py:177-186), GATv2 is NOT here (tecm_spatial_*), and
 * their autograd backward (dX and dW forms).
 * Epilogue order: v = alpha*acc; v += bias[n]; v += rowbias[((m / rb_div) % rb_mod)*rb_ld + n];
 *   if preact: preact[m*ldp+n] = v;  v = act(v);  if dact_src: v *= act'(dact_src[m*ldd+n]);
 *   v = dropout(v, out_drop);  if residual: v += residual[m*ldr+n];  if accumulate: v += C[m,n];
 *   store C (through c_win when enabled: column n is the inner index kk of the view). */
/* io_bf16 (tecm_gemm_bf16 only, 0 everywhere else): which tensors of the call already are / shall be bf16 in HBM.
 * A bf16 operand is [row][k] with k contiguous and its leading dimension counted in bf16 elements (multiple of 8,
 * 16-byte aligned, K % 8 == 0); the call must be the plain MK x NK contraction.  TECM_IO_C_BF16 writes C as bf16
 * (ldc in bf16 elements; needs the float4-friendly epilogue, no residual / accumulate / c_win / split_k); preact,
 * dact_src, bias stay fp32.  Rounding an activation once where it is produced instead of at every consumer's
 * loader gives bit-identical results at half the bytes.
 * TECM_IO_PRE_BF16: `preact` and `dact_src` are bf16 tensors (ldp / ldd in bf16 elements, multiples of 4, 8-byte
 * aligned; float4-friendly epilogue, tanh-GELU only, no split_k).  The value is rounded to bf16 BEFORE the
 * activation -- the store is `preact = bf16(v); v = act(float(preact))` -- so the forward's act() and the
 * backward's act'() are evaluated at the same point: what torch.autocast does to the output of a Linear that
 * feeds an activation (train.py:68), and what the bf16-emulating oracle (oracle/ref_cpu.BF16) restates. */
#define TECM_IO_A_BF16 1
#define TECM_IO_B_BF16 2
#define TECM_IO_C_BF16 4
#define TECM_IO_PRE_BF16 8
typedef struct TecmGemm {
  int64_t M, N, K;
  const float* A; int64_t lda; int32_t a_layout; int32_t io_bf16; TecmWin a_win; TecmDrop a_drop;
  const float* B; int64_t ldb; int32_t b_layout; int32_t _p1; TecmWin b_win; TecmDrop b_drop;
  float* C; int64_t ldc; TecmWin c_win;
  float alpha; int32_t act;
  const float* bias;
  const float* rowbias; int64_t rb_ld; int32_t rb_div; int32_t rb_mod;
  float* preact; int64_t ldp;
  const float* dact_src; int64_t ldd;
  TecmDrop out_drop;
  const float* residual; int64_t ldr;
  int32_t accumulate;
  int32_t split_k;             /* >1: partial sums go to workspace[split][M][N], then reduced */
  float* workspace;            /* >= split_k*M*N floats when split_k > 1 */
} TecmGemm;
int tecm_gemm_f32(const TecmGemm* desc, void* stream);
/* Same contract on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): operands are rounded to bf16 (RNE) while
 * staged into LDS, accumulation / epilogue / outputs stay fp32 -- what torch.autocast(bf16) does to the inputs
 * of Linear / Conv1d / Conv1D (train.py:68).  Operands must be 16-byte friendly (else TECM_E_ALIGN). */
int tecm_gemm_bf16(const TecmGemm* desc, void* stream);
/* "bf16x3": fp32 tensors, every factor split into a bf16 head and a bf16 remainder while it is staged, each product
 * evaluated as lo.hi + hi.lo + hi.hi on the bf16 matrix cores with fp32 accumulation (~16 mantissa bits per factor,
 * dot-product relative error ~1e-5; the bf16 MFMA runs at 16x the exact-f32 MFMA rate on gfx950).  Opt-in
 * (model_config["precision"] = "bf16x3"); serves the plain MK x NK contraction only, TECM_E_ARG otherwise. */
int tecm_gemm_bf16x3(const TecmGemm* g, void* stream);
/* "bf16x6": three-way split (hi + mid + lo carries all 24 mantissa bits), six products hi.lo + lo.hi + mid.mid +
 * hi.mid + mid.hi + hi.hi: only terms below 2^-24 of a product are dropped -- fp32-grade accuracy at 6/16 of the exact
 * matrix time.  Same scope and return convention as tecm_gemm_bf16x3. */
int tecm_gemm_bf16x6(const TecmGemm* g, void* stream);

/* Weight gradients in bf16 mode (a_layout KM x b_layout KN, both operands bf16 tensors, split_k >= 2: the backward of
 * every trainable nn.Linear / Conv1d(k=1) / LoRA matrix of the path, reference train.py:85) are served inside
 * tecm_gemm_bf16 by a natural-orientation LDS-DMA kernel (csrc/gemm_bf16_tn.hip) for the output shapes it is built for.
 * This returns the split count that kernel wants for an M x N output contracted over K rows (one or two rounds of
 * blocks on 256 CUs, slabs kept under a tenth of the operand bytes), or 0 when the shape is not served -- the caller
 * sizes split_k / the workspace with it; any split_k >= 2 is accepted.  Pure host function. */
int32_t tecm_gemm_tn_splits(int64_t M, int64_t N, int64_t K);
/* Rows per wave row (half the tile height: 128, 112 or 96) the eight-phase bf16 GEMM inside tecm_gemm_bf16 uses for an
 * M x N result on this device: the height whose tile count wastes least of the last round of CUs (csrc/gemm_bf16_p8.hip).
 * Exposed so that a caller's per-kernel accounting can name the instantiation a call runs on. */
int tecm_p8_rows(int64_t M, int64_t N);

/* ------------------------------------------------------------------ stage a-1..a-3 (fused)
 * SpatioTemporalEmbedding.forward (modules.py:230-266) + GATv2Conv (modules.py:329-336,:356)
 * + residual (tec_mollm.py:94).  The two permute copies (tec_mollm.py:84, :100-106) disappear:
 * input and output both stay (B, L, N, *) time-major. */
typedef struct TecmSpatial {
  int32_t B, L, N, Cin, Demb, H;          /* C = Cin + Demb = H*Ch */
  int32_t graphs_with_edges;              /* R: graphs g = t*B + b < R aggregate neighbours;
                                             1 = the reference's literal behaviour, B*L = every timestep */
  int32_t num_tiles, tile_nodes, win_max; /* node tiling computed by the host from the CSR */
  const float* x;                         /* (B, L, N, Cin) contiguous */
  const float* tf; int64_t tf_sb, tf_sl, tf_sn, tf_sf;   /* (B,L,N,4) with element strides; sn may be 0 */
  const float* node_tab; const float* tod_tab; const float* doy_tab; const float* year_tab;
  const float* season_tab;                /* (rows, Demb) each */
  int32_t year_rows;
  int32_t tile_edges_max;                 /* max over tiles of rowptr[n1] - rowptr[n0] (CSR slice staged in LDS) */
  const float* Wl; const float* bl; const float* Wr; const float* br;   /* (C,C),(C) */
  const float* att;                       /* (H, Ch) */
  const float* bias;                      /* (C) */
  const int32_t* rowptr;                  /* (N+1) CSR by target, self loops removed */
  const int32_t* colidx;                  /* (E) source node ids */
  const int32_t* tile_lo; const int32_t* tile_hi;   /* (num_tiles) window [lo,hi) of sources u targets */
  TecmDrop alpha_drop;                    /* GATv2 dropout on attention coefficients (modules.py:333) */
  float* out;                             /* (B, L, N, C) */
  /* Device error word (one int32, never NULL, owned by the caller, sticky: the kernels only OR bits in).  The
   * reference's nn.Embedding lookups raise on an out-of-range index (modules.py:255-258); here an index outside its
   * table sets TECM_BAD_TOD/DOY/YEAR/SEASON, and every output row built from it is NaN -- never a clamped, valid
   * looking embedding.  The caller reads the word back when it next synchronises (tecmollm/devcheck.py). */
  int32_t* err_flag;
  int32_t flags;                          /* TECM_SPATIAL_*; 0 = the fused stage as TEC_MoLLM.forward runs it */
  int32_t out_ld;                         /* row pitch of out: 24 (padded, 16-byte rows) in the fused path, >= 22 otherwise */
} TecmSpatial;
/* Stand-alone forms of the two modules the kernel fuses (their own `forward` in the reference):
 *   TECM_SPATIAL_EMBED_ONLY   SpatioTemporalEmbedding.forward (modules.py:230-266): out = cat([x, emb]), no graph work;
 *   TECM_SPATIAL_NO_RESIDUAL  SpatialEncoder.forward (modules.py:340-359): out = GATv2Conv(h) without `h +`; with
 *                             Demb = 0 the input rows ARE h (Cin = 22) and no table is read. */
#define TECM_SPATIAL_EMBED_ONLY 1
#define TECM_SPATIAL_NO_RESIDUAL 2
#define TECM_BAD_TOD 1
#define TECM_BAD_DOY 2
#define TECM_BAD_YEAR 4
#define TECM_BAD_SEASON 8
int tecm_spatial_fwd(const TecmSpatial* d, void* stream);
/* The same stage in a second formulation for the configuration TEC_MoLLM.forward runs (tec_mollm.py:84-94 with the
 * block-uniform time features of train.py:65, flags == 0, out_ld == 24, Cin 6 or 10): one 256-thread block per (tile,
 * graph) item, 37 KiB of LDS, four blocks per CU; the input transforms split into a graph-independent node part, a
 * per-graph vector (both computed by a set-up launch into `ws`) and a Cin-wide per-row part (csrc/spatial_fwd2.hip).
 * tecm_spatial_fwd2_ws_floats: floats of workspace the call needs, or 0 when this formulation does not serve `d` (then
 * use tecm_spatial_fwd).  Results agree with tecm_spatial_fwd to fp32 summation order. */
int64_t tecm_spatial_fwd2_ws_floats(const TecmSpatial* d);
int tecm_spatial_fwd2(const TecmSpatial* d, float* ws, void* stream);

typedef struct TecmSpatialGrads {
  const float* dout;                      /* (B, L, N, C) */
  float* d_node_tab;                      /* (N, Demb)  accumulated with float atomics: zero before call */
  float* d_tod_tab; float* d_doy_tab; float* d_year_tab; float* d_season_tab;   /* atomics: zero before call */
  float* partials; int64_t partial_ld;    /* (num_blocks, partial_ld) per-block sums of
                                             [dWl(C*C) dbl(C) dWr(C*C) dbr(C) datt(C) dbias(C)] */
  int32_t t_chunk; int32_t num_blocks;    /* num_blocks = tecm_spatial_bwd_blocks(d): the grid the library launches
                                             (persistent blocks over contiguous (tile, graph) ranges); t_chunk unused */
  /* the tile's edges grouped by SOURCE (d x_l is gathered per source row, no atomics): for tile k with window
   * [lo, hi) and edge segment [rowptr[n0], rowptr[n1]) of the by-target CSR,
   *   src_ptr[src_ptr_off[k] + w] .. src_ptr[src_ptr_off[k] + w + 1]   (w = j - lo, 0 <= w < hi - lo)
   * index, relative to rowptr[n0], the entries of src_col that leave source j; an entry is
   * ((target - n0) << 16) | pos, pos = position of the edge inside the tile's by-target segment (e - rowptr[n0]). */
  const int32_t* src_ptr; const int32_t* src_col; const int32_t* src_ptr_off;
} TecmSpatialGrads;
int tecm_spatial_bwd(const TecmSpatial* d, const TecmSpatialGrads* g, void* stream);
/* The same backward in the second formulation (csrc/spatial_bwd2.hip) for the configuration tecm_spatial_fwd2 serves: blocks
 * of 256 threads over (tile, chunk of graphs), ~45 KiB of LDS, three per CU.  Same TecmSpatialGrads contract (tables by
 * atomics into zeroed buffers, one row of `partials` per block in the same column layout) with
 * num_blocks = tecm_spatial_bwd2_blocks(d) (0 = not served: use tecm_spatial_bwd) and a workspace of
 * tecm_spatial_fwd2_ws_floats(d) floats that the call fills itself. */
int tecm_spatial_bwd2_blocks(const TecmSpatial* d);
int tecm_spatial_bwd2(const TecmSpatial* d, const TecmSpatialGrads* g, float* ws, void* stream);
/* Number of blocks (= rows of TecmSpatialGrads.partials) tecm_spatial_bwd will launch for `d`; negative = TECM_E_*. */
int tecm_spatial_bwd_blocks(const TecmSpatial* d);

/* ------------------------------------------------------------------ stage a-4 normalisation
 * nn.GroupNorm(1, C) + nn.GELU() (modules.py:28-29) for the three parallel branches at once.
 * y/act: (B, L, N, CT) time-major with CT = 3*Cout (branch j owns channels [j*Cout,(j+1)*Cout)).
 * stats: (B*N, 3, 2) = mean, rstd per sequence and branch. */
/* io_bf16 (BASELINE configs[2] only, 0 otherwise): TECM_GN_OUT_BF16 -- act (forward) / dy (backward) is written as bf16:
 * its only readers are bf16 matrix-core GEMMs that would round it in their loaders (train.py:68 autocast semantics).
 * y, the statistics and all arithmetic stay fp32. */
#define TECM_GN_OUT_BF16 2
/* backward only, with TECM_GN_OUT_BF16: dact is a bf16 tensor (written by the bf16 GEMM of the strided 1x1 conv's d-input) */
#define TECM_GN_DACT_BF16 4
/* y itself is the bf16 tensor a bf16 Conv1d hands to the fp32 GroupNorm under autocast (train.py:68; tecm_conv_fwd_bf16 with
 * y_bf16): forward with TECM_GN_OUT_BF16, backward with TECM_GN_OUT_BF16 | TECM_GN_DACT_BF16 -- every tensor bf16, statistics
 * and arithmetic fp32.  Served for L * 3*Cout/8 <= 2304 (tecm_gn_y16_supported). */
#define TECM_GN_Y_BF16 1
/* forward only, with TECM_GN_Y_BF16 | TECM_GN_OUT_BF16: `stats` is an INPUT -- the (mean, rstd) tecm_conv_fwd_bf16 computed from
 * the rounded y it produced (TecmConvFwd::stats); the norm + GELU then is elementwise over the time steps act keeps. */
#define TECM_GN_STATS_GIVEN 8
int tecm_gn_y16_supported(int32_t L, int32_t N, int32_t Cout);
/* act_stride s >= 1: only the time steps t % s == 0 are written, into a COMPACT (B, ceil(L / s), N, CT) tensor -- the
 * stride-s 1x1 conv behind the block (modules.py:36-41) reads nothing else; the statistics cover every step.  s > 1 is
 * served by the register-resident kernels only (TECM_E_ARG otherwise: callers ask tecm_gn_reg_ok first). */
int tecm_groupnorm_gelu_fwd(const void* y, const float* gamma, const float* beta, void* act,
                            float* stats, int32_t B, int32_t L, int32_t N, int32_t Cout, float eps,
                            int32_t io_bf16, int32_t act_stride, void* stream);
/* dact is (B, L/dstride, N, CT): the gradient exists only at t % dstride == 0 (stride-s 1x1 conv).
 * dgb_partials: (num_blocks, 3*CT) per-block [dgamma | dbeta | column sums of dy] -- the last third is the
 * gradient of the Conv1d biases in front of the norm (modules.py:27), free here since dy is being written;
 * returns num_blocks via *num_blocks when dy == NULL (query mode, nothing launched). */
int tecm_groupnorm_gelu_bwd(const void* dact, int32_t dstride, const void* y, const float* gamma,
                            const float* beta, const float* stats, void* dy, float* dgb_partials,
                            int32_t* num_blocks, int32_t B, int32_t L, int32_t N, int32_t Cout,
                            int32_t io_bf16,
                            float* seq_sums /* optional workspace of B*N*6 floats (all-bf16 form, Cout 64 / 128): the backward
                                               then runs as two streaming kernels -- per-sequence sums, then an elementwise
                                               pass -- instead of holding each sequence in registers.  NULL: one kernel */,
                            void* stream);

/* ------------------------------------------------------------------ stage a-6 pieces
 * nn.LayerNorm(768, eps=1e-5) of GPT2Block / ln_f (modeling_gpt2.py:262-310, :620). */
/* y (fp32) and / or y16 (bf16, RNE): the bf16 copy feeds a bf16 matrix-core GEMM (BASELINE configs[2]) that would
 * round the fp32 value in its loader anyway -- same bits, half the bytes; either pointer may be NULL, not both. */
int tecm_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                       int64_t ldy, void* y16, int64_t ldy16,
                       void* y16d /* optional third output: bf16(dropout(y, drop)) -- the LoRA branch's input, peft
                                     lora_dropout in front of lora_A (modules.py:181), cast as autocast casts it */,
                       int64_t ldy16d, const TecmDrop* drop /* of y16d; element index row*drop->ld + c; p = 0: no mask */,
                       int32_t y16d_seq_T, int32_t y16d_seq_N /* both > 0: y16d is written SEQUENCE-major -- the time-major
                                     row (b, t, n) of M = B*T*N goes to row (b, n, t): PredictionHead's view(batch, -1)
                                     (modules.py:307) of ln_f's output as a plain [B*N][T*D] matrix.  0, 0: same rows as y */,
                       float* stats /* (M,2) mean,rstd */, int64_t M, int32_t D, float eps, void* stream);
/* dx = dres (optional) + LN'(dy).  Optional second output dx_masked = dropout(dx, mask_drop): the
 * residual-stream gradient is consumed twice in GPT2Block's backward, once as is (residual path) and
 * once through the resid dropout in front of a GEMM; emitting the masked copy here costs one extra
 * store instead of one hash per element per GEMM column tile.  dgb_partials (num_blocks, 2*D);
 * dx == NULL only queries *num_blocks. */
/* masked_bf16 != 0: dx_masked is a bf16 (M, D) matrix (its only reader is a bf16 GEMM). */
/* Optional second gradient stream of the same tensor, added before the LayerNorm backward:
 *   dy[row][c] += keep(row*drop.ld + c) / (1 - p) * dy2[row][c]        (drop.p == 0: no mask)
 * -- the gradient peft's LoRA branch returns for its input (lora_A's d-input GEMM, modules.py:177-186) reaching the
 * LayerNorm output through lora_dropout's backward (modules.py:181).  dy2: (M, D) fp32 or bf16 with leading dimension ld. */
typedef struct TecmLnAdd {
  const void* dy2;
  int64_t ld;
  int32_t bf16, _pad;
  TecmDrop drop;
} TecmLnAdd;
/* dy in the sequence-major form tecm_layernorm_fwd can write y16d in (rows (b, n, t), bf16) and still in front of the
 * forward's dropout:  dy[(b,t,n)][c] = keep((b,t,n), c) / (1 - p) * dy16[(b,n,t)][c]  -- the gradient the head's first Linear
 * returns for F.dropout(ln_f(h)) (tec_mollm.py:115), taken as it is.  T == 0 / NULL: plain rows. */
typedef struct TecmLnDyMap {
  int32_t T, N;
  TecmDrop drop;                 /* p == 0: no mask */
} TecmLnDyMap;
int tecm_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                       const float* stats, const float* dres, float* dx, void* dx_masked, int32_t masked_bf16,
                       const TecmDrop* mask_drop, float* dgb_partials, int32_t* num_blocks, int64_t M, int32_t D,
                       const TecmLnAdd* add /* NULL: none */,
                       int32_t dy_bf16 /* != 0: dy is bf16 -- the gradient a bf16 Linear returns for its input under
                                          autocast (train.py:68) */,
                       const TecmLnDyMap* dymap /* NULL: none; needs dy_bf16 and no second stream */, void* stream);

/* Causal multi-head self-attention over T tokens per sequence (GPT2Attention, modeling_gpt2.py:54-73,
 * :144-226; all-ones attention_mask tec_mollm.py:111 => pure causal).  qkv: (B,T,N,3*D) time-major
 * rows, ctx: (B,T,N,D).  head_dim = D/heads must be 64.  Dropout on the probabilities. */
/* io_bf16: TECM_ATT_OUT_BF16 -- ctx is a bf16 (B,T,N,D) tensor (its only reader, attn.c_proj, is a bf16 GEMM);
 * TECM_ATT_QKV_BF16 -- qkv is a bf16 (B,T,N,3*D) tensor (bf16 mode: the c_attn GEMM stores its output as bf16, which is
 * what a Linear's output is under torch.autocast, train.py:68; scores, softmax and the weighted sum stay fp32). */
#define TECM_ATT_OUT_BF16 1
#define TECM_ATT_QKV_BF16 2
#define TECM_ATT_DCTX_BF16 4     /* backward only, with TECM_ATT_QKV_BF16: dctx is the bf16 tensor attn.c_proj's d-input GEMM wrote */
int tecm_attention_fwd(const float* qkv, void* ctx, int32_t io_bf16, int32_t B, int32_t T, int32_t N, int32_t heads,
                       int32_t D, const TecmDrop* prob_drop, void* stream);
/* io_bf16: TECM_ATT_OUT_BF16 -- dqkv is a bf16 (B,T,N,3*D) tensor (its readers, the c_attn dX and the LoRA-B dW GEMMs, are
 * bf16 GEMMs); TECM_ATT_QKV_BF16 -- qkv is the bf16 tensor the forward read; TECM_ATT_DCTX_BF16 -- dctx is bf16 (else fp32). */
int tecm_attention_bwd(const float* qkv, const float* dctx, void* dqkv, int32_t io_bf16, int32_t B, int32_t T, int32_t N,
                       int32_t heads, int32_t D, const TecmDrop* prob_drop, void* stream);

/* ------------------------------------------------------------------ reductions / small ops
 * out[s][c] (+)= sum over rows r = o*outer_stride + j (o < outer, j < inner) + seg s offset
 * of in[r*ld + c]; row(s,o,j) = (o*nseg + s)*inner + j when nseg > 1 (wpe / bias gradients). */
int tecm_colsum(const float* in, int64_t ld, int64_t outer, int64_t inner, int32_t nseg, int32_t C,
                float* out, int64_t ldo, int32_t accumulate, float scale, const TecmDrop* in_drop,
                float* workspace /* >= 1024*nseg*C floats */, void* stream);
/* ... and, from the same pass, the (masked) input as a bf16 matrix twin[row][ld_twin] (C % 4 == 0, C <= 1024, 16-byte friendly
 * rows): a gradient whose column sums are a bias gradient (fp32 values) and which two bf16 contractions read next (the
 * rounded values) -- backward of nn.Linear / Conv1d(k=1), train.py:85. */
int tecm_colsum_twin(const float* in, int64_t ld, int64_t outer, int64_t inner, int32_t nseg, int32_t C,
                     float* out, int64_t ldo, int32_t accumulate, float scale, const TecmDrop* in_drop,
                     float* workspace, void* twin_bf16, int64_t ld_twin, void* stream);

/* nn.HuberLoss(delta) mean (train.py:372) fused with its gradient: pred/target/dpred are (n) floats
 * addressed through pred_index: element e of pred = pred[e], target likewise (both contiguous in the
 * SAME logical order).  loss_out[0] = mean huber; dpred = dloss/dpred * grad_scale. */
int tecm_huber_fwd_bwd(const float* pred, const float* target, float* dpred, float* loss_out,
                       int64_t n, float delta, float grad_scale, float* workspace /* >= 1024 */,
                       void* stream);

/* Conv1d weight (Cout, Cin, k) -> GEMM operands.  fwd_pack: [Cout][k*Cin] with kk = tap*Cin + ci;
 * bwd_pack: [k*Cout][Cin] with row = tap'*Cout + co holding W[co][ci][k-1-tap'] (dX correlation). */
int tecm_conv_weight_pack(const float* w, float* fwd_pack, float* bwd_pack, int32_t Cout, int32_t Cin,
                          int32_t k, void* stream);
/* inverse of fwd_pack for the weight gradient: dW[co][ci][tap] = dpack[co][tap*Cin+ci]. */
int tecm_conv_weight_unpack(const float* dpack, float* dw, int32_t Cout, int32_t Cin, int32_t k,
                            void* stream);

/* Multi_Scale_Conv_Block (reference src/model/modules.py:43-60), bf16 mode (train.py:68): the input gradient of the three
 * parallel Conv1d(k = 3, 5, 7, padding (k-1)/2) in ONE launch that reads dy once (csrc/conv_seq.hip):
 *   dinp[b,t,n,ci] = sum_j sum_tau sum_co dy[b, t - tau + (k_j-1)/2, n, j*Cout + co] * w_j[co, ci, tau]
 * dy: bf16 (B, Lc, N, 3*Cout) time-major -- the GroupNorm+GELU backward's output; dinp: fp32 (B, Lc, N, ld_in), columns
 * >= Cin are written as zeros; wpack: the three weights (Cout, Cin, k_j) fp32 rounded to bf16 in MFMA fragment order by
 * tecm_conv_dx_pack, 15*Cout * 32*ceil(ld_in/32) bf16 values.  Cout % 64 == 0, ld_in % 4 == 0, ld_in <= 64,
 * Lc % 8 == 0; 16-byte aligned pointers. */
typedef struct TecmConvDx {
  const void* dy;
  const void* wpack;
  float* dinp;
  int32_t B, Lc, N, Cout, ld_in, _pad;
} TecmConvDx;
int tecm_conv_dx_pack(const float* w3, const float* w5, const float* w7, void* wpack, int32_t Cout, int32_t Cin,
                      int32_t ld_in, void* stream);
int tecm_conv_dx_bf16(const TecmConvDx* p, void* stream);
/* Forward of the same three Conv1d in ONE launch, bf16 mode: y (B, Lc, N, 3*Cout) fp32 = bias + conv(inp), inp bf16
 * (B, Lc, N, ld_in) with the real channels first (columns >= Cin are ignored: their weights are packed as zeros); bias =
 * b3 | b5 | b7 (3*Cout floats); wpack from tecm_conv_fwd_pack: sum_j ceil(k_j*ld_in/16) * Cout/32 * 512 bf16 values.
 * Cout % 32 == 0, Cout <= 128, ld_in % 8 == 0, Lc % 8 == 0. */
typedef struct TecmConvFwd {
  const void* inp;
  const void* wpack;
  const float* bias;
  float* y;                    /* fp32; y_bf16 != 0 (tecm_conv_fwd_bf16 only): a bf16 (B, Lc, N, 3*Cout) tensor -- what a bf16
                                  Conv1d returns under autocast (train.py:68), read by the TECM_GN_Y_BF16 norm kernels */
  int32_t B, Lc, N, Cout, ld_in, y_bf16;
  float* stats;                /* optional (tecm_conv_fwd_bf16 with y_bf16, Lc <= 48): (B*N, 3, 2) = mean, rstd of GroupNorm(1, Cout)
                                  (modules.py:28) per sequence and branch, from the ROUNDED y the kernel holds in registers -- the
                                  norm forward then is elementwise (TECM_GN_STATS_GIVEN).  NULL: not computed */
  float eps;                   /* GroupNorm eps (1e-5) */
  int32_t _pad;
} TecmConvFwd;
int tecm_conv_fwd_pack(const float* w3, const float* w5, const float* w7, void* wpack, int32_t Cout, int32_t Cin,
                       int32_t ld_in, void* stream);
int tecm_conv_fwd_bf16(const TecmConvFwd* p, void* stream);
/* ... and in exact fp32 (BASELINE configs[1], also the eval path): inp fp32, v_mfma_f32_32x32x2_f32; wpack from
 * tecm_conv_fwd_pack_f32: sum_j (k_j*ld_in/8) * Cout/32 * 256 floats. */
int tecm_conv_fwd_pack_f32(const float* w3, const float* w5, const float* w7, float* wpack, int32_t Cout, int32_t Cin,
                           int32_t ld_in, void* stream);
int tecm_conv_fwd_f32(const TecmConvFwd* p, void* stream);
/* The same in exact fp32 (BASELINE configs[1]): dy fp32, v_mfma_f32_32x32x2_f32; wpack from tecm_conv_dx_pack_f32
 * (15*Cout * 32*ceil(ld_in/32) floats). */
int tecm_conv_dx_pack_f32(const float* w3, const float* w5, const float* w7, float* wpack, int32_t Cout, int32_t Cin,
                          int32_t ld_in, void* stream);
int tecm_conv_dx_f32(const TecmConvDx* p, void* stream);
/* 1 when tecm_conv_fwd_{bf16,f32} / tecm_conv_dx_{bf16,f32} serve the shape -- divisibility AND the LDS bytes of one
 * sequence tile (64 KiB forward, 160 KiB d-input) -- else 0.  Pure host arithmetic, nothing is launched.  Callers route
 * the shapes these kernels refuse to the window-view GEMMs (tecm_gemm_*) instead of failing with TECM_E_LDS. */
int tecm_conv_fwd_supported(int32_t Lc, int32_t Cout, int32_t ld_in, int32_t f32);
int tecm_conv_dx_supported(int32_t Lc, int32_t Cout, int32_t ld_in, int32_t f32);

/* Weight gradient of the same three Conv1d in ONE launch pair, bf16 mode (replaces the three split-K window GEMMs
 * dY^T . window(inp) behind nn.Conv1d's autograd, modules.py:27,36):
 *   dw_j[co, ci, tau] = sum_{b,t,n} dy[b, t, n, j*Cout + co] * inp[b, t + tau - (k_j-1)/2, n, ci]     (zero outside [0, Lc))
 * inp bf16 (B, Lc, N, ld_in), dy bf16 (B, Lc, N, 3*Cout), dw3 / dw5 / dw7 fp32 (Cout, Cin, 3 / 5 / 7) -- only the Cin
 * real channels are written.  A persistent kernel of num_blocks thread blocks (one per CU is the intended use) keeps
 * all 15 taps' accumulators in registers and leaves one slab per block in `workspace`
 * (tecm_conv_dw_workspace(Cout, ld_in, num_blocks) floats); a second kernel adds the slabs in a fixed order.
 * (ld_in, Cout) in {24, 64} x {64, 128}; Lc % 4 == 0; 16-byte aligned tensors. */
typedef struct TecmConvDw {
  const void* inp;
  const void* dy;
  float* workspace;
  float* dw3;
  float* dw5;
  float* dw7;
  int32_t B, Lc, N, Cout, Cin, ld_in, num_blocks, _pad;
} TecmConvDw;
/* The same loss over a logical (B, H, N) index space with one stride triple (elements) per tensor: the model's
 * prediction is the permuted view (B, L_out, N, 1) of (B, N, L_out) storage (tec_mollm.py:122-123) and the target has its
 * own layout (train.py:76-78) -- no contiguous copies.  dpred is written with the PREDICTION's strides. */
int tecm_huber_fwd_bwd_strided(const float* pred, const int64_t* pred_strides /* 3 */, const float* target,
                               const int64_t* target_strides /* 3 */, float* dpred, float* loss_out, int32_t B, int32_t H,
                               int32_t N, float delta, float grad_scale, float* workspace /* >= 1024 */, void* stream);

int64_t tecm_conv_dw_workspace(int32_t Cout, int32_t ld_in, int32_t num_blocks);
int tecm_conv_dw_bf16(const TecmConvDw* p, void* stream);
/* The same in exact fp32 (BASELINE configs[1]): inp and dy fp32, v_mfma_f32_32x32x2_f32. */
int tecm_conv_dw_f32(const TecmConvDw* p, void* stream);

/* dst (bf16) [r][c] = round-to-nearest-even(src [r][c]) for a (rows, cols) block; cols % 4 == 0.  The cast torch.autocast
 * inserts in front of a Linear / Conv1D input (reference train.py:68), done once for a tensor a bf16 GEMM will read (here:
 * the 32 LoRA columns z = drop(LN1(h)) A^T, reference modules.py:177-186). */
int tecm_cast_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int32_t cols, void* stream);

/* bf16 forms of a trainable fp32 weight W [rows][cols] in one launch: same_bf16 = W rounded ([rows][ld_same]) and / or
 * transposed_bf16 = W^T rounded ([cols][ld_tr]) -- the values a bf16 contraction rounds W to in its loader, as tensors the
 * LDS-DMA GEMM can take as its [row][k] operand in the forward (nn.Linear / Conv1d(k=1): y = x W^T) and in the
 * input-gradient contraction (dx = dy W).  Either pointer may be NULL. */
int tecm_weight_bf16(const float* src, int64_t ld_src, void* same_bf16, int64_t ld_same, void* transposed_bf16, int64_t ld_tr,
                     int32_t rows, int32_t cols, void* stream);

/* dst[r][c] = src[r][c] * keep(seed, r*drop.ld + c) / (1 - p): the counter-based dropout mask every kernel of this
 * library recomputes (F.dropout of tec_mollm.py:115, GPT-2's embd dropout), materialised once where the masked
 * tensor is consumed several times.  cols, ld_src, ld_dst multiples of 4; 16-byte aligned. */
/* dst_bf16 != 0: dst is a bf16 tensor -- the cast autocast applies to the dropped value in front of a bf16 Linear
 * (train.py:68), done once where the masked tensor is only ever read by bf16 contractions. */
/* dst2_bf16 != NULL (dst fp32): the same masked values a second time as a bf16 tensor [r][ld_dst2] -- a gradient whose
 * column sums want the fp32 values and whose two bf16 contractions (dW, dX of the patch projection) read the rounded ones. */
int tecm_dropout_apply(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t dst_bf16, void* dst2_bf16,
                       int64_t ld_dst2, int64_t rows, int32_t cols, const TecmDrop* drop, void* stream);

/* dst[r*ldd + c] = scale * src[c*lds + r]  (r < rows, c < cols): builds the K-extended c_attn weight
 * [W ; (alpha/r) * B^T] (modules.py:177-183) and other small transposes. */
int tecm_transpose_scale(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t rows,
                         int32_t cols, float scale, void* stream);

/* LoRA fold (peft Linear on c_attn, reference modules.py:177-186): writes scale * lora_B (n_out, r) into the r trainable
 * rows of the K-extended backward operand w_kn = [ W ; scale B^T ] ([k_off + r][n_out], leading dimension ld_kn) and / or
 * the r trainable columns of the forward operand w_nk = [ W^T | scale B ] ([n_out][k_off + r], ld_nk); either may be
 * NULL; *_bf16 != 0: that operand is bf16.  The frozen part of both operands is the caller's (filled once). */
int tecm_lora_fold(const float* lora_B, int32_t n_out, int32_t r, float scale, void* w_kn, int64_t ld_kn, int32_t kn_bf16,
                   void* w_nk, int64_t ld_nk, int32_t nk_bf16, int32_t k_off, void* stream);

/* dst = srcs[0] | srcs[1] | ... (count <= 12 fp32 vectors of lens[i] elements, host arrays of device pointers): the
 * per-branch bias / gamma / beta of a Multi_Scale_Conv_Block (modules.py:27-29) as the 3*Cout vectors the fused kernels
 * read, in one launch. */
int tecm_pack_vectors(const float* const* srcs, const int32_t* lens, int32_t count, float* dst, void* stream);

/* ------------------------------------------------------------------ the shell around the step
 * (2) optimizer step of train.py:92-109 + :358-366 on FLAT buffers: global-norm clip + AdamW in two
 * launches, no host sync.  Semantics of torch.nn.utils.clip_grad_norm_(max_norm) followed by
 * torch.optim.AdamW(betas, eps, weight_decay).step() (decoupled decay, bias-corrected):
 *   g      = grad * grad_scale                          (grad_scale folds the data-parallel 1/world mean)
 *   norm   = ||g||_2 over all n values ; coef = min(1, max_norm / (norm + 1e-6))   (max_norm <= 0: coef = 1)
 *   g     *= coef ; p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
 *   p     -= (lr / (1-b1^step)) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * total_norm_out[0] = norm (device scalar, optional).  zero_grad != 0 clears grad in the same pass
 * (optimizer.zero_grad(), train.py:105).  partials: >= TECM_NORM_BLOCKS doubles of workspace. */
#define TECM_NORM_BLOCKS 512
typedef struct {
  int64_t n;
  float* param; float* grad; float* exp_avg; float* exp_avg_sq;
  double* partials;
  float* total_norm_out;
  float lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale;
  int32_t step;        /* 1-based count of this update (bias correction) */
  int32_t zero_grad;
  int32_t _pad;
} TecmAdamW;
int tecm_adamw_clip_step(const TecmAdamW* a, void* stream);
/* The parameter-divergence check that rides in the step's ONE gradient all-reduce (DDP's implicit invariant, train.py:353-354:
 * every rank holds the same parameters).  tecm_checksum_tail: f64 sum of `param[0..n)` (fixed order: identical bits on ranks
 * with identical parameters), split into two floats, written into slots 2*rank, 2*rank+1 of `tail` (2*world floats: the tail
 * of the flat gradient buffer), every other slot zeroed; `ws` >= 256 doubles of scratch.  After the SUM all-reduce
 * tecm_checksum_verify ORs `bit` into the device error word when any rank's pair differs from rank 0's.  Three launches for
 * what were eleven torch-dispatched ones (tecmollm/train.py). */
int tecm_checksum_tail(const float* param, int64_t n, float* tail, int32_t world, int32_t rank, double* ws, void* stream);
int tecm_checksum_verify(const float* tail, int32_t world, int32_t* err_word, int32_t bit, void* stream);
/* *word += inc on the stream (one thread).  The step's dropout word (TecmDrop::seed_dev): recorded as the first node of a
 * captured training step, it gives every replay its own masks -- what the reference gets from torch's generator advancing
 * under F.dropout (modules.py:307, modeling_gpt2.py attn/resid/embd dropout, train.py:68-93). */
int tecm_seed_advance(uint64_t* word, uint64_t inc, void* stream);

/* (3) evaluation metrics on device (src/evaluation/metrics.py:10-89, :119-183): per prediction horizon h
 * accumulate, over all (sample, node) pairs of one batch, the sufficient statistics of
 * evaluate_metrics(): after the non-finite guard on scaled predictions (:139-145, -> 0), the
 * StandardScaler inverse transform t = y*scale + mean (:36-37, rounded to f32 twice as sklearn does
 * on f32 input), nan_to_num(nan 0, +inf 100, -inf 0) (:40-46) and the clip of predictions to
 * [clip_lo, clip_hi] = [0, 200] TECU (:50-51).  pred/target are addressed as
 * base[s*stride_s + h*stride_h + i*stride_i] (element strides), so the model's permuted output view
 * (tec_mollm.py:123) is read in place.  stats: (H, 8) doubles, ACCUMULATED across calls:
 *   [count, sum t, sum p, sum t^2, sum p^2, sum t*p, sum |t-p|, sum (t-p)^2]. */
#define TECM_METRIC_STATS 8
typedef struct {
  const float* pred; int64_t p_stride_s, p_stride_h, p_stride_i;
  const float* target; int64_t t_stride_s, t_stride_h, t_stride_i;
  int64_t S; int32_t H; int32_t _pad; int64_t I;      /* samples, horizons, values per (sample, horizon) */
  double mean, scale;                                   /* scaler.mean_[0], scaler.scale_[0]; (0,1) = already unscaled */
  float clip_lo, clip_hi; int32_t clip;                /* clip != 0: clamp predictions */
  int32_t _pad2;
  double* stats;
} TecmMetrics;
int tecm_metrics_accumulate(const TecmMetrics* m, void* stream);

/* (4) sliding-window batch assembly from device-resident series (SlidingWindowSamplerDataset.__getitem__
 * src/data/dataset.py:65-99 + the harness reshapes train.py:62-65, :76): for sample b with window
 * start a = starts[b] (already multiplied by the dataset stride):
 *   x_out[b, t, :, :]   = X[a + t, :, :]                       t < L_in     (B, L_in, N*C) contiguous
 *   tf_out[b, t, :]     = TF[a + t, :]                                       (B, L_in, F_t)
 *   y_out[b, h, i]      = Y[a + L_in - 1, i, h]                h < L_out    (B, L_out, N): the target
 *                         already in the (B, L_out, N, 1) order of train.py:76
 * X (T, N*C), TF (T, F_t), Y (T, N, L_out), starts (B) int64 on the device.  Out-of-range windows are
 * rejected on the host only when starts_host_check is given (same values, host memory); the kernel
 * clamps nothing. */
typedef struct {
  const float* X; const float* TF; const float* Y; const int64_t* starts;
  const int64_t* starts_host_check;
  int64_t T; int64_t row; /* N*C floats per time step */
  int32_t N, L_in, L_out, F_t, B, _pad;
  float* x_out; float* tf_out; float* y_out;
} TecmWindowBatch;
int tecm_window_batch(const TecmWindowBatch* w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TECMOLLM_H */
