"""MI355X-native mirror of the reference's `src/model/modules.py` (PANXIONG-CN/TEC-MoLLM).

Same class names, constructor arguments and state-dict keys as the reference
(`/root/reference/src/model/modules.py:13-359`); the arithmetic runs in hand-written HIP kernels
(libtecmollm_hip.so) through `tecmollm.functions`.  The modules here mostly *own parameters*: the fused
data path (embedding + GATv2 + residual in one kernel, time-major activations, no permute copies)
is driven from `TEC_MoLLM.forward` in `tec_mollm.py`.  Sub-modules that are meaningful on their
own (TemporalEncoder, LLMBackbone, PredictionHead) keep a working stand-alone `forward` with the
reference's tensor shapes; the two that only exist fused say so.

There is no PyTorch fallback: tensors must be CUDA (ROCm) fp32, otherwise the call raises.
"""
from __future__ import annotations

import logging
import math
from typing import List, Optional

import torch
import torch.nn as nn

import os

from tecmollm import functions as F_
from tecmollm import ops as ops_
from tecmollm import graph as graph_
from tecmollm._lib import TecmError

log = logging.getLogger(__name__)

_seed_counter = [0]


def autocast_bf16() -> bool:
    """True inside `torch.autocast('cuda', dtype=torch.bfloat16)` -- how the reference trains (train.py:68)."""
    return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16


def make_plan(module: nn.Module, p: float = 0.1, precision: str = "auto") -> F_.DropPlan:
    """One plan per forward call: dropout seeds (base seed = torch seed + call count) and the GEMM precision:
    "fp32" = exact-f32 MFMA, "bf16" = bf16 MFMA with fp32 accumulate, "auto" = bf16 iff under bf16 autocast,
    "bf16x3" = fp32 emulated by three bf16 MFMAs per product on the plain GPT-2 GEMMs (opt-in, ~1e-5)."""
    _seed_counter[0] += 1
    bf16 = precision == "bf16" or (precision == "auto" and autocast_bf16())
    if precision == "bf16x3":
        bf16 = 2                                   # ops.PREC_BF16X3
    elif precision == "bf16x6":
        bf16 = 3                                   # ops.PREC_BF16X6
    return F_.DropPlan(training=module.training, p=p, base_seed=(torch.initial_seed() + 7919 * _seed_counter[0]),
                       bf16=bf16)


def _need_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise TecmError(f"{name} is on {t.device}: the MI355X path has no CPU fallback (move model and inputs to cuda)")
    if t.dtype != torch.float32:
        raise TecmError(f"{name} must be float32 (got {t.dtype})")


class Multi_Scale_Conv_Block(nn.Module):
    """modules.py:13-60.  Parameters live in the reference's layout: convs.{j}.0 = Conv1d, convs.{j}.1 = GroupNorm."""

    def __init__(self, in_channels: int, out_channels: int, stride: int, kernel_sizes: list = [3, 5, 7]):
        super().__init__()
        if list(kernel_sizes) != [3, 5, 7]:
            raise ValueError("the HIP path is built for kernel_sizes [3, 5, 7] (reference default)")
        if out_channels not in (64, 128, 256):
            raise ValueError("out_channels must be 64, 128 or 256 (GroupNorm kernel works on 64-channel chunks)")
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.convs = nn.ModuleList([
            nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size=k, padding=(k - 1) // 2),
                          nn.GroupNorm(1, out_channels), nn.GELU()) for k in kernel_sizes])
        self.final_conv = nn.Conv1d(out_channels * len(kernel_sizes), out_channels, kernel_size=1, stride=stride)

    def forward_tm(self, inp: torch.Tensor, cin: int, need_dinp: bool = True, bf16: bool = False,
                   inp16: Optional[torch.Tensor] = None):
        """inp (B, Lc, N, ld) time-major with `cin` real channels -> (out (B, Lc/stride, N, Cout), out16): out16 is a
        bf16 copy of out in bf16 mode (None otherwise) for the window GEMMs of the next stage; inp16 likewise."""
        args = []
        for seq in self.convs:
            args += [seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias]
        if inp16 is None and int(bf16) == ops_.PREC_BF16 and inp.shape[-1] % 8 == 0 and os.environ.get("TECM_XS16", "1")[:1] != "0":
            # a block called on its own in bf16 mode (the embedder hands its blocks the bf16 copy): round the input once here,
            # so that the sequence-tile kernels -- and with them the bf16 storage of y -- serve this call as they serve the model
            inp16 = torch.empty(inp.shape, device=inp.device, dtype=torch.bfloat16)
            ops_.cast_bf16(inp.detach(), inp.shape[-1], inp16, inp.shape[-1], inp.numel() // inp.shape[-1], inp.shape[-1])
        return F_.ConvBlockFn.apply(inp, inp16, cin, self.stride, need_dinp, bf16, *args, self.final_conv.weight,
                                    self.final_conv.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Reference signature: x (S, C_in, L) -> (S, C_out, L_out)."""
        _need_cuda(x, "x")
        S, Cc, L = x.shape
        tm = x.permute(0, 2, 1).contiguous().view(S, L, 1, Cc)
        out, _ = self.forward_tm(tm, Cc, bf16=autocast_bf16())
        return out.view(S, out.shape[1], self.out_channels).permute(0, 2, 1)


class MultiScaleConvEmbedder(nn.Module):
    """modules.py:62-88."""

    def __init__(self, in_channels: int, channel_list: list, strides: list):
        super().__init__()
        assert len(channel_list) == len(strides), "Channel list and strides list must have the same length."
        layers = []
        cur = in_channels
        for out_channels, stride in zip(channel_list, strides):
            layers.append(Multi_Scale_Conv_Block(cur, out_channels, stride))
            cur = out_channels
        self.embedder = nn.Sequential(*layers)

    def forward_tm(self, inp: torch.Tensor, cin: int, need_dinp: bool = True, bf16: bool = False):
        inp16 = None
        if int(bf16) == ops_.PREC_BF16 and inp.shape[-1] % 8 == 0 and os.environ.get("TECM_XS16", "1")[:1] != "0":
            # bf16 mode: the first block's window GEMMs (three forward convs, three weight gradients: 15 tap reads of
            # the spatial stage's output) read a bf16 copy rounded ONCE here -- the same bits their loaders would produce
            inp16 = torch.empty(inp.shape, device=inp.device, dtype=torch.bfloat16)
            rows = inp.numel() // inp.shape[-1]
            ops_.cast_bf16(inp.detach(), inp.shape[-1], inp16, inp.shape[-1], rows, inp.shape[-1])
        for i, blk in enumerate(self.embedder):
            inp, inp16 = blk.forward_tm(inp, cin, need_dinp or i > 0, bf16, inp16)
            cin = blk.out_channels
        return inp, inp16

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for blk in self.embedder:
            x = blk(x)
        return x


class LatentPatchingProjection(nn.Module):
    """modules.py:90-119."""

    def __init__(self, latent_dim: int, patch_len: int, d_llm: int):
        super().__init__()
        self.patch_len = patch_len
        self.projection = nn.Linear(patch_len * latent_dim, d_llm)

    def forward_tm(self, conv: torch.Tensor, wpe: Optional[torch.Tensor], plan: F_.DropPlan,
                   conv16: Optional[torch.Tensor] = None) -> torch.Tensor:
        return F_.PatchEmbedFn.apply(conv, conv16, self.projection.weight, self.projection.bias, wpe, self.patch_len, plan)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Reference signature: x (S, L, D_latent) -> (S, num_patches, d_llm)."""
        _need_cuda(x, "x")
        S, L, D = x.shape
        out = self.forward_tm(x.contiguous().view(S, L, 1, D), None, F_.DropPlan(False, 0.0, 0, autocast_bf16()))
        return out.view(S, out.shape[1], -1)


class TemporalEncoder(nn.Module):
    """modules.py:121-154."""

    def __init__(self, in_channels: int, channel_list: list, strides: list, patch_len: int, d_llm: int):
        super().__init__()
        self.in_channels = in_channels
        self.conv_embedder = MultiScaleConvEmbedder(in_channels, channel_list, strides)
        self.patcher = LatentPatchingProjection(channel_list[-1], patch_len, d_llm)

    def forward_tm(self, inp, cin, wpe, plan, need_dinp=True):
        conv, conv16 = self.conv_embedder.forward_tm(inp, cin, need_dinp, plan.bf16)
        return self.patcher.forward_tm(conv, wpe, plan, conv16)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Reference signature: x (S, L_in, C_in) -> (S, num_patches, d_llm).  A sequence-major (S, L, C)
        tensor IS time-major with B=S, N=1, so no permute is needed (the reference's :143/:149 copies vanish)."""
        _need_cuda(x, "x")
        S, L, Cc = x.shape
        out = self.forward_tm(x.contiguous().view(S, L, 1, Cc), Cc, None, F_.DropPlan(False, 0.0, 0, autocast_bf16()),
                              need_dinp=x.requires_grad)
        return out.view(S, out.shape[1], -1)


# ------------------------------------------------------------------------------------------- GPT-2 + LoRA
class _Conv1D(nn.Module):
    """transformers.pytorch_utils.Conv1D parameter holder: weight (in, out), bias (out)."""

    def __init__(self, nf: int, nx: int):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.empty(nx, nf).normal_(std=0.02))
        self.bias = nn.Parameter(torch.zeros(nf))


class _LoraLinearHolder(nn.Module):
    def __init__(self, in_f: int, out_f: int):
        super().__init__()
        self.default = nn.Linear(in_f, out_f, bias=False)


class _LoraConv1D(nn.Module):
    """peft.tuners.lora.Linear wrapped around c_attn: base_layer + lora_A.default + lora_B.default."""

    def __init__(self, nf: int, nx: int, r: int):
        super().__init__()
        self.base_layer = _Conv1D(nf, nx)
        self.lora_A = _LoraLinearHolder(nx, r)         # weight (r, nx), kaiming-uniform like peft
        self.lora_B = _LoraLinearHolder(r, nf)         # weight (nf, r), zeros like peft
        nn.init.kaiming_uniform_(self.lora_A.default.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B.default.weight)


class _Attn(nn.Module):
    def __init__(self, d: int, r: int):
        super().__init__()
        self.c_attn = _LoraConv1D(3 * d, d, r)
        self.c_proj = _Conv1D(d, d)


class _MLP(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.c_fc = _Conv1D(4 * d, d)
        self.c_proj = _Conv1D(d, 4 * d)


class _Block(nn.Module):
    def __init__(self, d: int, r: int):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d, eps=1e-5)
        self.attn = _Attn(d, r)
        self.ln_2 = nn.LayerNorm(d, eps=1e-5)
        self.mlp = _MLP(d)


class _GPT2Trunk(nn.Module):
    """GPT2Model parameter tree: wte, wpe, h[i], ln_f."""

    def __init__(self, n_layers: int, d: int = 768, r: int = 32, n_positions: int = 1024, vocab: int = 50257,
                 include_wte: bool = True):
        super().__init__()
        if include_wte:
            self.wte = nn.Embedding(vocab, d)          # unused by the inputs_embeds path; kept for strict load
            nn.init.normal_(self.wte.weight, std=0.02)
        self.wpe = nn.Embedding(n_positions, d)
        nn.init.normal_(self.wpe.weight, std=0.02)
        self.h = nn.ModuleList([_Block(d, r) for _ in range(n_layers)])
        self.ln_f = nn.LayerNorm(d, eps=1e-5)


class _BaseModel(nn.Module):
    def __init__(self, trunk: _GPT2Trunk):
        super().__init__()
        self.model = trunk


class _PeftGPT2(nn.Module):
    """Stands where peft.PeftModel stood: keys `base_model.model.*`, plus the two methods callers touch
    (train.py:70-73 `gradient_checkpointing_enable`, modules.py:193 `print_trainable_parameters`)."""

    def __init__(self, trunk: _GPT2Trunk):
        super().__init__()
        self.base_model = _BaseModel(trunk)

    def gradient_checkpointing_enable(self, *args, **kwargs):
        # Nothing to recompute: activations are kept (288 GB HBM); the reference re-enables this every step.
        return None

    def print_trainable_parameters(self):
        tr = sum(p.numel() for p in self.parameters() if p.requires_grad)
        al = sum(p.numel() for p in self.parameters())
        print(f"trainable params: {tr:,} || all params: {al:,} || trainable%: {100 * tr / max(al, 1):.4f}")


def hf_key_of(own_key: str) -> str:
    """Name of one of our trunk parameters inside a plain transformers GPT2Model state dict: peft wraps c_attn, so
    `attn.c_attn.base_layer.{weight,bias}` here is `attn.c_attn.{weight,bias}` there (modules.py:177-186)."""
    return own_key.replace("attn.c_attn.base_layer.", "attn.c_attn.")


def copy_hf_gpt2_weights(trunk: _GPT2Trunk, hf_state: dict) -> int:
    """Copy a transformers GPT2Model state dict into the trunk (first `len(trunk.h)` blocks, modules.py:170).
    Every non-LoRA parameter of the trunk MUST be found with the same shape: a silent partial copy would train LoRA on
    top of a half-random backbone.  Returns the number of tensors copied."""
    own = trunk.state_dict()
    missing, wrong, n = [], [], 0
    for k, dst in own.items():
        if ".lora_A." in k or ".lora_B." in k:
            continue
        src = hf_state.get(hf_key_of(k))
        if src is None:
            missing.append(k)
        elif tuple(src.shape) != tuple(dst.shape):
            wrong.append((k, tuple(src.shape), tuple(dst.shape)))
        else:
            dst.copy_(src)
            n += 1
    if missing or wrong:
        raise RuntimeError(f"GPT-2 checkpoint does not cover the trunk: missing {missing[:5]} (+{max(len(missing) - 5, 0)}), "
                           f"shape mismatches {wrong[:5]}")
    return n


def _load_pretrained_gpt2(trunk: _GPT2Trunk) -> None:
    """AutoModel.from_pretrained('gpt2') exactly as the reference does (modules.py:165): it downloads the checkpoint or
    fails loudly.  A machine without the checkpoint (and without network) therefore raises here; pass
    model_config['load_pretrained_gpt2'] = False to ask for config-style initialisation explicitly."""
    from transformers import AutoModel
    hf = AutoModel.from_pretrained("gpt2")
    copy_hf_gpt2_weights(trunk, hf.state_dict())


class LLMBackbone(nn.Module):
    """modules.py:156-209: GPT-2 truncated to `num_layers_to_keep` blocks, LoRA(r=32, alpha=64, dropout 0.1) on
    c_attn, everything frozen except names containing lora_/ln_/wpe."""

    def __init__(self, num_layers_to_keep: int = 3, include_wte: bool = True, load_pretrained: bool = True):
        super().__init__()
        trunk = _GPT2Trunk(num_layers_to_keep, include_wte=include_wte)
        if load_pretrained:
            with torch.no_grad():
                _load_pretrained_gpt2(trunk)
        self.num_layers = num_layers_to_keep
        self.model = _PeftGPT2(trunk)
        self._freeze_parameters()

    def _freeze_parameters(self):
        for p in self.model.parameters():
            p.requires_grad = False
        for name, p in self.model.named_parameters():
            if "lora_" in name or "ln_" in name or "wpe" in name:
                p.requires_grad = True

    @property
    def trunk(self) -> _GPT2Trunk:
        return self.model.base_model.model

    def stack_params(self) -> List[torch.Tensor]:
        ps: List[torch.Tensor] = []
        t = self.trunk
        for blk in t.h:
            ps += [blk.ln_1.weight, blk.ln_1.bias, blk.attn.c_attn.base_layer.weight, blk.attn.c_attn.base_layer.bias,
                   blk.attn.c_attn.lora_A.default.weight, blk.attn.c_attn.lora_B.default.weight,
                   blk.attn.c_proj.weight, blk.attn.c_proj.bias, blk.ln_2.weight, blk.ln_2.bias,
                   blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias]
        ps += [t.ln_f.weight, t.ln_f.bias]
        return ps

    def forward_tm(self, h0: torch.Tensor, plan: F_.DropPlan) -> torch.Tensor:
        """h0 (B, T, N, 768) = inputs_embeds + wpe (already added by the patch GEMM epilogue)."""
        return F_.GPT2StackFn.apply(h0, self.num_layers, plan, *self.stack_params())

    def forward(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reference signature (modules.py:205-209): (S, T, 768) -> last_hidden_state (S, T, 768).
        attention_mask must be all ones (tec_mollm.py:111) -- the kernel is purely causal."""
        _need_cuda(inputs_embeds, "inputs_embeds")
        S, T, D = inputs_embeds.shape
        plan = make_plan(self)
        wpe = self.trunk.wpe.weight
        h0 = inputs_embeds + wpe[:T]                     # stand-alone use only; fused path adds wpe in the GEMM
        if plan.training and plan.p > 0:
            h0 = torch.nn.functional.dropout(h0, plan.p, True)
        out = self.forward_tm(h0.contiguous().view(S, T, 1, D), plan)
        return out.view(S, T, D)


class SpatioTemporalEmbedding(nn.Module):
    """modules.py:211-266.  Parameter holder: the gather + sum + concat runs inside the fused spatial kernel."""

    def __init__(self, d_emb: int, num_nodes: int = 2911, num_years: int = 13):
        super().__init__()
        self.d_emb = d_emb
        self.node_embedding = nn.Embedding(num_nodes, d_emb)
        self.tod_embedding = nn.Embedding(12, d_emb)
        self.doy_embedding = nn.Embedding(366, d_emb)
        self.year_embedding = nn.Embedding(num_years, d_emb)
        self.season_embedding = nn.Embedding(4, d_emb)

    def tables(self):
        return (self.node_embedding.weight, self.tod_embedding.weight, self.doy_embedding.weight,
                self.year_embedding.weight, self.season_embedding.weight)

    def forward(self, x: torch.Tensor, time_features: torch.Tensor) -> torch.Tensor:
        """Reference signature (modules.py:230-266): x (B, L, N, C_in), time_features (B, L, N, 4) -> (B, L, N, C_in +
        d_emb).  Stand-alone use only: TEC_MoLLM.forward runs this fused with GATv2 + residual in one kernel."""
        _need_cuda(x, "x")
        _need_cuda(time_features, "time_features")
        if x.dim() != 4 or time_features.shape != (*x.shape[:3], 4):
            raise ValueError("x must be (B, L, N, C_in) and time_features (B, L, N, 4)")
        if x.shape[2] != self.node_embedding.num_embeddings:
            raise ValueError(f"x has {x.shape[2]} nodes, the node table {self.node_embedding.num_embeddings}")
        return F_.EmbedFn.apply(x, time_features, *self.tables())


class _GATv2Params(nn.Module):
    """torch_geometric.nn.GATv2Conv parameter tree: att (1,H,C), bias (H*C), lin_l / lin_r Linear (glorot / zeros)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int):
        super().__init__()
        self.lin_l = nn.Linear(in_channels, heads * out_channels)
        self.lin_r = nn.Linear(in_channels, heads * out_channels)
        self.att = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        for lin in (self.lin_l, self.lin_r):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)
        nn.init.xavier_uniform_(self.att)


class SpatialEncoder(nn.Module):
    """modules.py:315-359.  Parameter holder (gat_conv.*), fused into the spatial kernel."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 2, dropout: float = 0.1):
        super().__init__()
        self.gat_conv = _GATv2Params(in_channels, out_channels, heads)
        self.heads = heads
        self.dropout = dropout
        self.output_channels = out_channels * heads
        # "reference": a single-graph edge_index only connects rows 0..N-1 of the flattened (G*N) input, i.e. graph 0
        # (what the reference computes); "per_timestep": every graph of the batch
        self.gat_graphs = "reference"

    def params(self):
        g = self.gat_conv
        return (g.lin_l.weight, g.lin_l.bias, g.lin_r.weight, g.lin_r.bias, g.att, g.bias)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reference signature (modules.py:340-359): x (num_graphs, N, C_in) -> (num_graphs, N, heads * out_channels);
        edge_weight is ignored as in the reference.  Stand-alone use only (TEC_MoLLM.forward fuses this with the
        embedding and the residual); gradients reach the GATv2 parameters, x must not require grad."""
        _need_cuda(x, "x")
        if x.dim() != 3 or x.shape[2] != self.output_channels:
            raise ValueError(f"x must be (num_graphs, N, {self.output_channels})")
        G, N, _ = x.shape
        meta = graph_.get(edge_index, N, x.device, 0)
        plan = make_plan(self, self.dropout, precision="fp32")
        R = 1 if self.gat_graphs == "reference" else G
        return F_.GatFn.apply(x, *self.params(), meta, self.heads, R, plan)


class PredictionHead(nn.Module):
    """modules.py:268-313."""

    def __init__(self, input_dim: int, output_dim: int, hidden_dim_ratio: int = 4, dropout_rate: float = 0.1):
        super().__init__()
        hidden = input_dim // hidden_dim_ratio
        self.dropout_rate = dropout_rate
        self.mlp = nn.Sequential(nn.Linear(input_dim, hidden), nn.GELU(), nn.Dropout(dropout_rate),
                                 nn.Linear(hidden, output_dim))

    def forward_tm(self, hid: torch.Tensor, plan: F_.DropPlan) -> torch.Tensor:
        """hid (B, T, N, 768) -> (B, N, L_out); the F.dropout of tec_mollm.py:115 is applied in the GEMM's A loader."""
        return F_.HeadFn.apply(hid, self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight, self.mlp[3].bias, plan)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Reference signature: x (S, T, hidden) -> (S, output_dim) (no post-LLM dropout here, as in the reference)."""
        _need_cuda(x, "x")
        S, T, D = x.shape
        plan = make_plan(self, self.dropout_rate)
        out = F_.HeadFn.apply(x.contiguous().view(S, T, 1, D), self.mlp[0].weight, self.mlp[0].bias,
                              self.mlp[3].weight, self.mlp[3].bias, _NoPostDrop(plan))
        return out.view(S, -1)


class _NoPostDrop(F_.DropPlan):
    """Plan that keeps the head's own dropout but not the post-LLM one (stand-alone PredictionHead)."""

    def __init__(self, plan: F_.DropPlan):
        super().__init__(plan.training, plan.p, plan.base_seed, plan.bf16)

    def spec(self, site: int, ld: int):
        return None if site == F_.SITE_POST else super().spec(site, ld)


__all__ = ["Multi_Scale_Conv_Block", "MultiScaleConvEmbedder", "LatentPatchingProjection", "TemporalEncoder",
           "LLMBackbone", "SpatioTemporalEmbedding", "PredictionHead", "SpatialEncoder", "graph_"]
