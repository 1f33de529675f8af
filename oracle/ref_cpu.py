"""CPU oracle for the TEC-MoLLM forward/backward hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-CPU *restatement* of the reference algorithm
(`/root/reference/src/model/tec_mollm.py:59-125` and `src/model/modules.py`).
It exists so that the HIP path can be checked against an independent statement of
the same arithmetic.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product package
(`tec-mollm_amd/`) never does and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * embed / temporal encoder / prediction head / full wiring -- PINNED against the
    reference's own classes, imported in the authoring container by
    `oracle/make_golden.py`; vectors committed under `tests/golden/`.
  * GPT-2 block math -- PINNED against the installed `transformers` 5.15.0
    `GPT2Model` (the reference's call sites: modules.py:165,170,208), same script.
  * GATv2Conv (torch_geometric) and LoRA (peft) -- third-party packages that are
    not installed anywhere in this environment and whose versions the reference
    does not pin: **parity unpinned**.  They are restated from the published
    algorithm (Brody et al. 2021 / PyG `GATv2Conv`; Hu et al. 2021 / peft `Linear`)
    and cross-checked by an independent dense-adjacency formulation in
    `tests/test_oracle.py`.

Everything is functional: parameters come in as a dict keyed by the reference's
state-dict names, activations are ordinary tensors, gradients
come from torch autograd on CPU.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------- names
P_EMB = "spatio_temporal_embedding."
P_GAT = "spatial_encoder.gat_conv."
P_CONV = "temporal_encoder.conv_embedder.embedder."
P_PATCH = "temporal_encoder.patcher.projection."
P_GPT = "llm_backbone.model.base_model.model."
P_HEAD = "prediction_head.mlp."

LORA_R = 32
LORA_ALPHA = 64
LORA_SCALE = LORA_ALPHA / LORA_R  # modules.py:177-183  (alpha / r = 2.0)
GAT_NEG_SLOPE = 0.2
GPT2_HEADS = 12
LN_EPS = 1e-5


# ------------------------------------------------------------------- stage a-1
def embed(x: torch.Tensor, tf: torch.Tensor, p: Params) -> torch.Tensor:
    """SpatioTemporalEmbedding.forward, modules.py:230-266.

    x (B,L,N,C_in) f32, tf (B,L,N,4) f32 integer-valued -> (B,L,N,C_in+d_emb).
    Sum order follows modules.py:260-261: ((tod+doy)+year)+season, then node + temporal.
    """
    B, L, N, _ = x.shape
    node = p[P_EMB + "node_embedding.weight"][torch.arange(N)].view(1, 1, N, -1)
    tod = p[P_EMB + "tod_embedding.weight"][tf[..., 0].long()]
    doy = p[P_EMB + "doy_embedding.weight"][tf[..., 1].long()]
    year = p[P_EMB + "year_embedding.weight"][tf[..., 2].long()]
    season = p[P_EMB + "season_embedding.weight"][tf[..., 3].long()]
    temporal = tod + doy + year + season
    return torch.cat([x, node + temporal], dim=-1)


# ------------------------------------------------------------------- stage a-2
def gatv2_conv(x: torch.Tensor, edge_index: torch.Tensor, p: Params, heads: int,
               alpha_keep: Optional[torch.Tensor] = None, drop_p: float = 0.0,
               alpha_mult: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch_geometric.nn.GATv2Conv(in, out, heads, concat=True, add_self_loops=True,
    share_weights=False, negative_slope=0.2) restated (call site modules.py:329-336,:356).

    x (M, C); edge_index (2,E) int64 with [0]=source j, [1]=target i.  Self loops in
    edge_index are removed, then one self loop per row is appended for ALL M rows
    (PyG: `add_self_loops(edge_index, num_nodes=x.size(0))`).  Softmax per target with
    max-subtraction and `+1e-16` in the denominator (PyG `utils.softmax`).
    `alpha_keep` (E',heads) in {0,1}: optional dropout keep-mask applied to alpha as
    alpha*keep/(1-drop_p); edge order = [non-self edges in given order, then self loops 0..M-1].
    `alpha_mult` (E',heads): the same thing as a ready multiplier (0 or 1/(1-p)).
    """
    M, C = x.shape
    Wl, bl = p[P_GAT + "lin_l.weight"], p[P_GAT + "lin_l.bias"]
    Wr, br = p[P_GAT + "lin_r.weight"], p[P_GAT + "lin_r.bias"]
    att, bias = p[P_GAT + "att"], p[P_GAT + "bias"]          # att (1,H,Ch)
    H = heads
    Ch = Wl.shape[0] // H
    xl = (x @ Wl.t() + bl).view(M, H, Ch)                    # source side
    xr = (x @ Wr.t() + br).view(M, H, Ch)                    # target side
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    loops = torch.arange(M, dtype=edge_index.dtype)
    src = torch.cat([src[keep], loops])
    dst = torch.cat([dst[keep], loops])
    s = xl[src] + xr[dst]                                    # (E',H,Ch)
    e = (F.leaky_relu(s, GAT_NEG_SLOPE) * att.view(1, H, Ch)).sum(-1)   # (E',H)
    emax = torch.full((M, H), float("-inf")).scatter_reduce(
        0, dst.view(-1, 1).expand(-1, H), e, reduce="amax", include_self=True)
    pexp = (e - emax[dst]).exp()
    denom = torch.zeros(M, H).index_add_(0, dst, pexp) + 1e-16
    alpha = pexp / denom[dst]
    if alpha_keep is not None:
        alpha = alpha * alpha_keep / (1.0 - drop_p)
    if alpha_mult is not None:
        alpha = alpha * alpha_mult
    out = torch.zeros(M, H, Ch).index_add_(0, dst, alpha.unsqueeze(-1) * xl[src])
    return out.reshape(M, H * Ch) + bias


def batched_edge_index(edge_index: torch.Tensor, num_nodes: int, num_graphs: int) -> torch.Tensor:
    """edge_index replicated with offsets g*N for g < num_graphs (what PyG batching would
    have produced had the reference built a Batch; it does not -- SURVEY.md section 0)."""
    offs = (torch.arange(num_graphs, dtype=edge_index.dtype) * num_nodes).view(1, -1, 1)
    return (edge_index.unsqueeze(1) + offs).reshape(2, -1)


def spatial(h: torch.Tensor, edge_index: torch.Tensor, p: Params, heads: int,
            graphs_with_edges: Optional[int] = 1, alpha_mult: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tec_mollm.py:84-94 + modules.py:340-359: permute to (L*B, N, C), GATv2 on the
    flattened (L*B*N, C) rows, residual add.  Returns x_spatial (L*B, N, C).

    graphs_with_edges=1 is the reference's literal behaviour: edge ids < N only touch
    graph 0 (t=0,b=0); every other row sees just its self loop.  None = every one of the
    L*B graphs gets the edges (the per-timestep behaviour the reference's comments intend).
    alpha_mult (E', heads): training-mode dropout of the attention coefficients (modules.py:333, GATv2Conv
    `dropout=0.1`) as a ready multiplier 0 | 1/(1-p), in gatv2_conv's edge order.
    """
    B, L, N, C = h.shape
    xg = h.permute(1, 0, 2, 3).reshape(-1, N, C)
    G = L * B if graphs_with_edges is None else graphs_with_edges
    ei = batched_edge_index(edge_index, N, G)
    gat = gatv2_conv(xg.reshape(-1, C), ei, p, heads, alpha_mult=alpha_mult).view(L * B, N, C)
    return xg + gat


# ------------------------------------------------------------------- precision emulation (BASELINE configs[2])
def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (RNE) and back: what the bf16 MFMA path does to GEMM operands (autocast semantics)."""
    return t.bfloat16().float()


def _ident(t: torch.Tensor) -> torch.Tensor:
    return t


class Rounding:
    """Operand rounding of the dense contractions as a PAIR: `fwd` is applied to both operands of a forward
    contraction, `bwd` to both operands of the two contractions its backward consists of (dX = dY.W^T and
    dW = X^T.dY).  Accumulation, bias, activation, dropout, norms, softmax and the GATv2 stage are fp32 in every mode;
    the tensors that are rounded where they are STORED (`stored`, below) are the GPT-2 c_fc output GELU is evaluated at
    and the c_attn output qkv the attention reads -- bf16 Linear outputs under autocast.
    FP32 = the reference's CPU arithmetic.  BF16 = what `torch.autocast('cuda', bfloat16)` (train.py:68) does to the
    operands of Linear / Conv1d / matmul, forward AND backward, as the MI355X bf16 mode implements it: fp32 outputs,
    bf16 operands.  A tensor the HIP path stores in HBM as bf16 (LN outputs, attention context, gelu(c_fc), conv
    activations; in the backward dqkv, d gelu-input, the dropout-masked LayerNorm-backward outputs, the conv dy) is a
    tensor whose readers are bf16 contractions only, so rounding at the store and rounding in the reader are the same
    arithmetic -- with the exceptions written next to the call sites below (`dx=` / `dw=` flags)."""

    def __init__(self, fwd=None, bwd=None, name="fp32", conv_policy=None):
        self.fwd, self.bwd, self.name = fwd, bwd, name
        # conv_policy(L, Cout, Cin) -> (acts16, y16): which conv-block tensors are STORED as bf16 (see conv_block).  None =
        # torch.autocast's own semantics: a bf16 Conv1d returns a bf16 tensor, always.  An implementation that keeps some of
        # them fp32 for some shapes (the HIP path at L_in = 336) hands its policy in from the TEST (tests/parity.py:
        # device_rounding); the oracle holds no copy of any kernel's eligibility rules.
        self.conv_policy = conv_policy

    def with_conv_policy(self, conv_policy) -> "Rounding":
        return Rounding(self.fwd, self.bwd, self.name, conv_policy)

    def __repr__(self):
        return f"Rounding({self.name})"


FP32 = Rounding(None, None, "fp32")
BF16 = Rounding(bf16_round, bf16_round, "bf16")
BF16_FORWARD_ONLY = Rounding(bf16_round, None, "bf16-forward-only")   # forward emulation with an exact backward


def _r(fn, t):
    return t if fn is None else fn(t)


class _StoreRounded(torch.autograd.Function):
    """A tensor kept in bf16 between its producer and a NON-contraction reader (the c_fc output GELU is evaluated
    at, forward and backward): rounded where it is stored, gradient passed through unchanged."""

    @staticmethod
    def forward(ctx, t, fn):
        return fn(t)

    @staticmethod
    def backward(ctx, g):
        return g, None


def stored(t: torch.Tensor, q: "Rounding") -> torch.Tensor:
    return t if q.fwd is None else _StoreRounded.apply(t, q.fwd)


class _GradStoreRounded(torch.autograd.Function):
    """Identity whose GRADIENT is rounded where it is stored: a backward tensor kept in bf16 between the contraction that
    produces it and a non-contraction reader (the gradient at the strided 1x1 conv's input, read by the GroupNorm + GELU
    backward -- what the backward of a bf16 Conv1d hands to fp32 ops under autocast)."""

    @staticmethod
    def forward(ctx, t, fn):
        ctx.fn = fn
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return ctx.fn(g), None


def grad_stored(t: torch.Tensor, q: "Rounding") -> torch.Tensor:
    return t if q.bwd is None else _GradStoreRounded.apply(t, q.bwd)


class _MatMul(torch.autograd.Function):
    """y = a @ b with a (..., K), b (K, N): the three contractions of a Linear, each with its own operand rounding."""

    @staticmethod
    def forward(ctx, a, b, rf, rdx, rdw):
        ctx.save_for_backward(a, b)
        ctx.r = (rdx, rdw)
        return _r(rf, a) @ _r(rf, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        rdx, rdw = ctx.r
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _r(rdx, g) @ _r(rdx, b).t()
        if ctx.needs_input_grad[1]:
            a2 = _r(rdw, a).reshape(-1, a.shape[-1])
            db = a2.t() @ _r(rdw, g).reshape(-1, g.shape[-1])
        return da, db, None, None, None


class _Conv1d(torch.autograd.Function):
    """F.conv1d without bias, the same three-way operand rounding (the HIP path runs it as a window-view GEMM)."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, rf, rdx, rdw):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, rdx, rdw)
        return F.conv1d(_r(rf, x), _r(rf, w), None, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, padding, rdx, rdw = ctx.cfg
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.nn.grad.conv1d_input(x.shape, _r(rdx, w), _r(rdx, g), stride=stride, padding=padding)
        if ctx.needs_input_grad[1]:
            dw = torch.nn.grad.conv1d_weight(_r(rdw, x), w.shape, _r(rdw, g), stride=stride, padding=padding)
        return dx, dw, None, None, None, None, None


def mm(a: torch.Tensor, b: torch.Tensor, q: Rounding = FP32, f: bool = True, dx: bool = True, dw: bool = True):
    """a @ b under rounding policy q.  f / dx / dw = False: THAT contraction runs on the exact fp32 kernel in the HIP
    path even in bf16 mode (fewer than 64 output columns and no bf16-resident operand: tecmollm/ops.py:uses_bf16)."""
    if q.fwd is None and q.bwd is None:
        return a @ b
    return _MatMul.apply(a, b, q.fwd if f else None, q.bwd if dx else None, q.bwd if dw else None)


def conv1d(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, q: Rounding = FP32, stride: int = 1, padding: int = 0):
    if q.fwd is None and q.bwd is None:
        return F.conv1d(x, w, bias, stride=stride, padding=padding)
    return _Conv1d.apply(x, w, stride, padding, q.fwd, q.bwd, q.bwd) + bias.view(1, -1, 1)


# ------------------------------------------------------------------- stage a-4/a-5
def conv_block(x: torch.Tensor, p: Params, idx: int, stride: int, q: Rounding = FP32) -> torch.Tensor:
    """Multi_Scale_Conv_Block.forward, modules.py:43-60: x (S, C_in, L).
    bf16 mode: all four convolutions and their dX / dW contractions have >= 64 output columns or read the bf16-resident
    dy (block 0's dX has 24 columns but its A operand is the bf16 dy), so every one is a bf16 contraction."""
    outs = []
    cout = p[f"{P_CONV}{idx}.convs.0.0.weight"].shape[0]
    acts16, y16 = q.conv_policy(x.shape[-1], cout, x.shape[1]) if q.conv_policy is not None else (True, True)
    for j, k in enumerate((3, 5, 7)):
        pre = f"{P_CONV}{idx}.convs.{j}."
        y = conv1d(x, p[pre + "0.weight"], p[pre + "0.bias"], q, padding=(k - 1) // 2)
        if y16:
            y = stored(y, q)                                 # autocast's Conv1d output IS a bf16 tensor; the norm is fp32
        y = F.group_norm(y, 1, p[pre + "1.weight"], p[pre + "1.bias"], eps=1e-5)
        outs.append(F.gelu(y))
    cat = torch.cat(outs, dim=1)
    if acts16:
        cat = grad_stored(cat, q)                            # the bf16 gradient a bf16 Conv1d returns for its input
    pre = f"{P_CONV}{idx}.final_conv."
    return conv1d(cat, p[pre + "weight"], p[pre + "bias"], q, stride=stride)


def temporal_encoder(x: torch.Tensor, p: Params, strides, patch_len: int, q: Rounding = FP32) -> torch.Tensor:
    """TemporalEncoder.forward modules.py:134-154 + LatentPatchingProjection :100-119.
    x (S, L, C) -> (S, P, d_llm).  Patch vector index = l*D + d (einops 'b (p l) d -> b p (l d)')."""
    y = x.permute(0, 2, 1)
    for i, s in enumerate(strides):
        y = conv_block(y, p, i, s, q)
    y = y.permute(0, 2, 1)                                   # (S, L', D)
    S, Lc, D = y.shape
    y = y.reshape(S, Lc // patch_len, patch_len * D)
    return mm(y, p[P_PATCH + "weight"].t(), q) + p[P_PATCH + "bias"]


# ------------------------------------------------------------------- stage a-6
def gelu_new(x: torch.Tensor) -> torch.Tensor:
    """transformers NewGELUActivation (GPT-2 `activation_function='gelu_new'`)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def _mul(t: torch.Tensor, masks, key: str) -> torch.Tensor:
    """Training-mode dropout with a GIVEN mask: masks[key] is the multiplier 0 | 1/(1-p), same shape as t
    (None / missing key = eval mode).  The masks are inputs of the oracle, not drawn here: tests feed the
    mirror of the device's counter-based masks (tecmollm/rng.py) so both sides drop the same elements."""
    if masks is None or masks.get(key) is None:
        return t
    m = masks[key]
    assert m.shape == t.shape, (key, tuple(m.shape), tuple(t.shape))
    return t * m


def gpt2_lora(h: torch.Tensor, p: Params, n_layers: int, q: Rounding = FP32, masks=None) -> torch.Tensor:
    """LLMBackbone.forward modules.py:205-209 -> peft(GPT2Model)(inputs_embeds=h, all-ones mask).
    h (S, T, 768).  c_attn' = base Conv1D + 2.0 * B(A(lora_dropout(u))).
    Dropout sites (all p = 0.1 in the reference; masks=None is eval mode): GPT2Model `drop` on
    inputs_embeds + wpe ("embd", modeling_gpt2.py embd_pdrop), peft's lora_dropout on the LoRA branch input
    ("lora{i}", modules.py:181), attention-probability dropout ("attn{i}", (S, heads, T, T), attn_pdrop),
    residual dropouts after attn.c_proj and mlp.c_proj ("res1_{i}", "res2_{i}", resid_pdrop).
    bf16 mode (q = BF16), as the HIP path runs it: c_attn + LoRA-B is ONE contraction with K = 768 + 32 over
    [LN1(h) | z] and [W ; 2 B^T]; the LoRA-A product z reads the bf16 copy of drop(LN1(h)) the LayerNorm kernel stores
    (what autocast casts in front of lora_A) -- a bf16 contraction like its two backward ones (round 4; it ran on the
    exact kernel before)."""
    S, T, D = h.shape
    hd = D // GPT2_HEADS
    h = _mul(h + p[P_GPT + "wpe.weight"][:T], masks, "embd")
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    for i in range(n_layers):
        pre = f"{P_GPT}h.{i}."
        u = F.layer_norm(h, (D,), p[pre + "ln_1.weight"], p[pre + "ln_1.bias"], LN_EPS)
        A = p[pre + "attn.c_attn.lora_A.default.weight"]     # (r, 768)
        Bm = p[pre + "attn.c_attn.lora_B.default.weight"]    # (2304, r)
        z = mm(grad_stored(_mul(u, masks, f"lora{i}"), q), A.t(), q)      # lora_A's input gradient: a bf16 tensor too
        wcat = torch.cat([p[pre + "attn.c_attn.base_layer.weight"], LORA_SCALE * Bm.t()], 0)     # (768 + r, 2304)
        # (grad_stored: the gradient a bf16 Linear returns for its input is a bf16 tensor under autocast -- d [LN1-out | z],
        #  d ctx and d LN2-out are stored as such by the device's d-input GEMMs and read by fp32 kernels)
        qkv = stored(mm(grad_stored(torch.cat([u, z], -1), q), wcat, q) + p[pre + "attn.c_attn.base_layer.bias"], q)   # a bf16 tensor under autocast
        qq, k, v = qkv.split(D, dim=-1)
        qq = qq.view(S, T, GPT2_HEADS, hd).transpose(1, 2)
        k = k.view(S, T, GPT2_HEADS, hd).transpose(1, 2)
        v = v.view(S, T, GPT2_HEADS, hd).transpose(1, 2)
        w = (qq @ k.transpose(-1, -2)) / math.sqrt(hd)
        w = _mul(w.masked_fill(~causal, float("-inf")).softmax(-1), masks, f"attn{i}")
        ctx = (w @ v).transpose(1, 2).reshape(S, T, D)
        h = h + _mul(mm(grad_stored(ctx, q), p[pre + "attn.c_proj.weight"], q) + p[pre + "attn.c_proj.bias"], masks, f"res1_{i}")
        u = F.layer_norm(h, (D,), p[pre + "ln_2.weight"], p[pre + "ln_2.bias"], LN_EPS)
        # autocast's c_fc output IS a bf16 tensor (train.py:68): GELU and its derivative see the rounded value
        f = gelu_new(stored(mm(grad_stored(u, q), p[pre + "mlp.c_fc.weight"], q) + p[pre + "mlp.c_fc.bias"], q))
        h = h + _mul(mm(f, p[pre + "mlp.c_proj.weight"], q) + p[pre + "mlp.c_proj.bias"], masks, f"res2_{i}")
    return F.layer_norm(h, (D,), p[P_GPT + "ln_f.weight"], p[P_GPT + "ln_f.bias"], LN_EPS)


# ------------------------------------------------------------------- stage a-8
def head(x: torch.Tensor, p: Params, q: Rounding = FP32, masks=None) -> torch.Tensor:
    """PredictionHead.forward modules.py:295-313: (S,T,768) -> (S, L_out); nn.Dropout after the GELU
    (modules.py:289, mask "head" (S, hidden)).
    bf16 mode: the L_out-column output layer runs on the exact kernel in the forward (fewer than 64 output columns);
    its dX and dW contractions have `hidden` (>= 64) output columns and are bf16."""
    z = x.reshape(x.shape[0], -1)
    z = _mul(F.gelu(mm(z, p[P_HEAD + "0.weight"].t(), q) + p[P_HEAD + "0.bias"]), masks, "head")
    hidden = p[P_HEAD + "0.weight"].shape[0]
    small = p[P_HEAD + "3.weight"].shape[0] < 64
    return mm(z, p[P_HEAD + "3.weight"].t(), q, f=not small, dx=hidden >= 64, dw=hidden >= 64) + p[P_HEAD + "3.bias"]


# ------------------------------------------------------------------- full path
def forward(x: torch.Tensor, tf: torch.Tensor, edge_index: torch.Tensor, p: Params, cfg: dict,
            graphs_with_edges: Optional[int] = 1, q: Rounding = FP32, masks=None) -> torch.Tensor:
    """TEC_MoLLM.forward tec_mollm.py:59-125.  Returns (B, L_out, N, 1).
    q=BF16 emulates the bf16 MFMA mode, forward and backward (GATv2's 22x22 transforms stay fp32 there).
    masks=None: eval mode.  masks = {site: multiplier tensor}: training mode with the given dropout masks at
    every site of the reference -- "gat" (modules.py:333), "embd"/"lora{i}"/"attn{i}"/"res1_{i}"/"res2_{i}"
    (GPT-2 + peft, see gpt2_lora), "post" (F.dropout tec_mollm.py:115, (S,T,768)), "head" (modules.py:289)."""
    B, L, N, _ = x.shape
    h = embed(x, tf, p)
    xs = spatial(h, edge_index, p, cfg["spatial_heads"], graphs_with_edges,
                 alpha_mult=None if masks is None else masks.get("gat"))
    C = xs.shape[-1]
    xt = xs.view(L, B, N, C).permute(1, 2, 0, 3).reshape(B * N, L, C)
    tok = temporal_encoder(xt, p, cfg["temporal_strides"], cfg["patch_len"], q)
    # (grad_stored: the gradient the head's first Linear -- a bf16 contraction under autocast -- returns for its input is a
    #  bf16 tensor; the device stores it as such in front of the dropout's and ln_f's backward)
    hid = grad_stored(_mul(gpt2_lora(tok, p, cfg["llm_layers"], q, masks), masks, "post"), q)
    pred = head(hid, p, q, masks)
    return pred.view(B, N, -1).permute(0, 2, 1).unsqueeze(-1)


def random_masks(cfg: dict, B: int, edge_index: torch.Tensor, graphs_with_edges: Optional[int], p: float = 0.1,
                 generator: Optional[torch.Generator] = None) -> dict:
    """Training-mode dropout masks (multiplier 0 | 1/(1-p)) for every site of `forward`, drawn with torch's own
    Bernoulli generator -- what the reference does inside each step (p = 0.1 everywhere: modules.py:181, :272, :333,
    tec_mollm.py:115, GPT-2 config defaults).  Used by the timed CPU baseline; parity tests feed the device's masks."""
    N, L = cfg["num_nodes"], cfg["temporal_seq_len"]
    D, H = cfg["d_llm"], cfg["spatial_heads"]
    T = (L // (cfg["temporal_strides"][0] * cfg["temporal_strides"][1])) // cfg["patch_len"]
    S = B * N
    G = L * B if graphs_with_edges is None else graphs_with_edges
    n_edges = int((edge_index[0] != edge_index[1]).sum()) * G + L * B * N

    def draw(*shape):
        return torch.bernoulli(torch.full(shape, 1.0 - p), generator=generator) / (1.0 - p)

    m = {"gat": draw(n_edges, H), "embd": draw(S, T, D), "post": draw(S, T, D), "head": draw(S, (D * T) // 4)}
    for i in range(cfg["llm_layers"]):
        m[f"lora{i}"] = draw(S, T, D)
        m[f"attn{i}"] = draw(S, GPT2_HEADS, T, T)
        m[f"res1_{i}"] = draw(S, T, D)
        m[f"res2_{i}"] = draw(S, T, D)
    return m


def huber(out: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """nn.HuberLoss(delta=1.0), mean reduction (train.py:372)."""
    return F.huber_loss(out, y, delta=1.0)


# ------------------------------------------------------------------- parameters
def default_config(L_in: int = 48, L_out: int = 12, num_nodes: int = 2911, c_in: int = 6,
                   d_emb: int = 16, llm_layers: int = 3) -> dict:
    """model_config dict as built at train.py:249-269 (patch_len fallback 4 -> 2 -> 1)."""
    conv_len = L_in // 4
    patch_len = 4
    if conv_len % patch_len != 0:
        patch_len = 2 if conv_len % 2 == 0 else 1
    return {
        "num_nodes": num_nodes, "d_emb": d_emb, "spatial_in_channels_base": c_in,
        "spatial_out_channels": (c_in + d_emb) // 2, "spatial_heads": 2,
        "temporal_channel_list": [64, 128], "temporal_strides": [2, 2], "patch_len": patch_len,
        "d_llm": 768, "llm_layers": llm_layers, "prediction_horizon": L_out,
        "temporal_seq_len": L_in, "num_years": 13,
    }


def init_params(cfg: dict, seed: int = 0, include_wte: bool = False) -> Params:
    """Deterministic config-style initialisation under the reference's state-dict names.
    GPT-2 weights N(0, 0.02) (pretrained weights are unavailable offline); LoRA B is
    NON-zero so the LoRA path is exercised; LayerNorm/GroupNorm affine perturbed off 1/0."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    def lin(out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        return ((torch.rand(out_f, in_f, generator=g) * 2 - 1) * b,
                (torch.rand(out_f, generator=g) * 2 - 1) * b)

    p: Params = {}
    d = cfg["d_emb"]
    N = cfg["num_nodes"]
    for name, rows in (("node", N), ("tod", 12), ("doy", 366), ("year", cfg.get("num_years", 13)),
                       ("season", 4)):
        p[f"{P_EMB}{name}_embedding.weight"] = rn(rows, d)
    C = cfg["spatial_in_channels_base"] + d
    H, Ch = cfg["spatial_heads"], cfg["spatial_out_channels"]
    assert H * Ch == C, "residual needs base + d_emb == out_channels*heads (tec_mollm.py:94)"
    p[P_GAT + "att"] = rn(1, H, Ch, std=0.4)
    p[P_GAT + "bias"] = rn(C, std=0.1)
    for s in ("l", "r"):
        p[f"{P_GAT}lin_{s}.weight"] = rn(H * Ch, C, std=1.0 / math.sqrt(C))
        p[f"{P_GAT}lin_{s}.bias"] = rn(H * Ch, std=0.1)
    cin = C
    for i, cout in enumerate(cfg["temporal_channel_list"]):
        for j, k in enumerate((3, 5, 7)):
            b = 1.0 / math.sqrt(cin * k)
            p[f"{P_CONV}{i}.convs.{j}.0.weight"] = (torch.rand(cout, cin, k, generator=g) * 2 - 1) * b
            p[f"{P_CONV}{i}.convs.{j}.0.bias"] = (torch.rand(cout, generator=g) * 2 - 1) * b
            p[f"{P_CONV}{i}.convs.{j}.1.weight"] = 1.0 + rn(cout, std=0.1)
            p[f"{P_CONV}{i}.convs.{j}.1.bias"] = rn(cout, std=0.1)
        b = 1.0 / math.sqrt(3 * cout)
        p[f"{P_CONV}{i}.final_conv.weight"] = (torch.rand(cout, 3 * cout, 1, generator=g) * 2 - 1) * b
        p[f"{P_CONV}{i}.final_conv.bias"] = (torch.rand(cout, generator=g) * 2 - 1) * b
        cin = cout
    D = cfg["d_llm"]
    p[P_PATCH + "weight"], p[P_PATCH + "bias"] = lin(D, cfg["patch_len"] * cin)
    if include_wte:
        p[P_GPT + "wte.weight"] = rn(50257, D, std=0.02)
    p[P_GPT + "wpe.weight"] = rn(1024, D, std=0.02)
    for i in range(cfg["llm_layers"]):
        pre = f"{P_GPT}h.{i}."
        for ln in ("ln_1", "ln_2"):
            p[pre + ln + ".weight"] = 1.0 + rn(D, std=0.1)
            p[pre + ln + ".bias"] = rn(D, std=0.1)
        p[pre + "attn.c_attn.base_layer.weight"] = rn(D, 3 * D, std=0.02)
        p[pre + "attn.c_attn.base_layer.bias"] = rn(3 * D, std=0.02)
        p[pre + "attn.c_attn.lora_A.default.weight"] = rn(LORA_R, D, std=1.0 / math.sqrt(D))
        p[pre + "attn.c_attn.lora_B.default.weight"] = rn(3 * D, LORA_R, std=0.02)
        p[pre + "attn.c_proj.weight"] = rn(D, D, std=0.02)
        p[pre + "attn.c_proj.bias"] = rn(D, std=0.02)
        p[pre + "mlp.c_fc.weight"] = rn(D, 4 * D, std=0.02)
        p[pre + "mlp.c_fc.bias"] = rn(4 * D, std=0.02)
        p[pre + "mlp.c_proj.weight"] = rn(4 * D, D, std=0.02)
        p[pre + "mlp.c_proj.bias"] = rn(D, std=0.02)
    p[P_GPT + "ln_f.weight"] = 1.0 + rn(D, std=0.1)
    p[P_GPT + "ln_f.bias"] = rn(D, std=0.1)
    n_patches = (cfg["temporal_seq_len"] // (cfg["temporal_strides"][0] * cfg["temporal_strides"][1])) \
        // cfg["patch_len"]
    hin = D * n_patches
    p[P_HEAD + "0.weight"], p[P_HEAD + "0.bias"] = lin(hin // 4, hin)
    p[P_HEAD + "3.weight"], p[P_HEAD + "3.bias"] = lin(cfg["prediction_horizon"], hin // 4)
    return p


def is_trainable(name: str) -> bool:
    """Freeze rule modules.py:195-203: inside the HF model only lora_/ln_/wpe train."""
    if name.startswith("llm_backbone."):
        return ("lora_" in name) or ("ln_" in name) or ("wpe" in name)
    return True


# ------------------------------------------------------------------- synthetic inputs
def grid_graph(n_lat: int = 41, n_lon: int = 71, lat0: float = 15.0, lon0: float = 70.0,
               step: float = 1.0, threshold_km: float = 150.0):
    """Haversine <= threshold adjacency on a regular lat/lon grid, no self loops,
    symmetric-normalised weights -- graph_constructor.py:46-56, :75-78, :112-125, :141-144.
    Node id = lat_index * n_lon + lon_index (meshgrid + ravel order, :46-47)."""
    import numpy as np
    lat = np.radians(lat0 + step * np.arange(n_lat))
    lon = np.radians(lon0 + step * np.arange(n_lon))
    lon_g, lat_g = np.meshgrid(lon, lat)
    la, lo = lat_g.ravel(), lon_g.ravel()
    n = la.size
    src, dst = [], []
    reach = int(math.ceil(threshold_km / (111.0 * step * math.cos(math.radians(lat0 + step * n_lat))))) + 1
    for i in range(n):
        r, c = divmod(i, n_lon)
        r0, r1 = max(0, r - reach), min(n_lat, r + reach + 1)
        c0, c1 = max(0, c - reach), min(n_lon, c + reach + 1)
        cand = (np.arange(r0, r1)[:, None] * n_lon + np.arange(c0, c1)[None, :]).ravel()
        dlat = la[cand] - la[i]
        dlon = lo[cand] - lo[i]
        a = np.sin(dlat / 2) ** 2 + np.cos(la[i]) * np.cos(la[cand]) * np.sin(dlon / 2) ** 2
        dist = 2 * np.arcsin(np.sqrt(a)) * 6371.0
        nb = cand[(dist <= threshold_km) & (cand != i)]
        src.extend([i] * len(nb))
        dst.extend(nb.tolist())
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    deg = np.bincount(src, minlength=n).astype(np.float64)
    inv = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0.0)
    w = (inv[src] * inv[dst]).astype(np.float32)
    return torch.from_numpy(np.stack([src, dst])), torch.from_numpy(w)


def synthetic_batch(B: int, L_in: int, N: int, c_in: int, L_out: int, seed: int = 1234):
    """Synthetic batch shaped like train.py:58-65 / dataset.py:86-99: x~N(0,1); integer time
    features (tod<12, doy<366, year<13, season<4) as f32, stride-0 expanded over N; y~N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L_in, N, c_in, generator=g)
    tf = torch.stack([torch.randint(0, hi, (B, L_in), generator=g) for hi in (12, 366, 13, 4)], -1).float()
    tf = tf.unsqueeze(-2).expand(B, L_in, N, 4)
    y = torch.randn(B, L_out, N, 1, generator=g)
    return x, tf, y
