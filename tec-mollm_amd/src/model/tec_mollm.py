"""MI355X-native mirror of the reference's `src/model/tec_mollm.py` (`/root/reference/src/model/tec_mollm.py:15-125`).

Drop-in for `from src.model.tec_mollm import TEC_MoLLM` (train.py:18, test.py:13): same constructor
dict, same forward signature and output shape, same parameter names.  Differences by design:

  * the data path is time-major (B, T, N, C) end to end, so the two permute copies of
    tec_mollm.py:84 and :100-106 do not exist;
  * stages 1-4 of the reference forward (embedding, GATv2, residual, layout) are ONE kernel;
  * `edge_weight` is optional (it is ignored by the reference too, modules.py:355-356), which also
    makes test.py:37's 3-argument call valid;
  * model_config may carry `precision`: "auto" (default: bf16 matrix cores iff called under
    `torch.autocast('cuda', torch.bfloat16)` as train.py:68 does, exact fp32 otherwise), "fp32", "bf16", or
    "bf16x3" (opt-in: the plain GPT-2 GEMMs evaluate every fp32 product as three bf16 matrix-core products of the
    hi/lo bf16 split of its factors -- relative error ~1e-5, far inside the 1e-3 parity bar, but not exact fp32;
    "bf16x6": three-way split, six products, fp32-grade accuracy at 6/16 of the exact matrix time);
  * model_config may carry `gat_graphs`: "reference" (default: a single-graph edge_index only
    connects graph 0 = (t=0, b=0), exactly what the reference computes -- SURVEY.md section 0) or
    "per_timestep" (every (b, t) graph aggregates neighbours, what the reference's comments intend).
"""
from __future__ import annotations

import dataclasses
import logging
import os
from typing import Optional

import torch
import torch.nn as nn

from tecmollm import functions as F_
from tecmollm import graph as graph_
from .modules import (LLMBackbone, PredictionHead, SpatialEncoder, SpatioTemporalEmbedding, TemporalEncoder,
                      _need_cuda, make_plan)

log = logging.getLogger(__name__)


def _fuse_head() -> bool:
    """TECM_FUSE_HEAD=0: ln_f and the head as separate stages (A/B diagnostics)."""
    return os.environ.get("TECM_FUSE_HEAD", "1") != "0"

_REQUIRED = ("num_nodes", "d_emb", "spatial_in_channels_base", "spatial_out_channels", "spatial_heads",
             "temporal_channel_list", "temporal_strides", "patch_len", "d_llm", "llm_layers", "temporal_seq_len",
             "prediction_horizon")


class TEC_MoLLM(nn.Module):
    """The main TEC-MoLLM model (tec_mollm.py:15-57)."""

    def __init__(self, model_config: dict):
        super().__init__()
        missing = [k for k in _REQUIRED if k not in model_config]
        if missing:
            raise KeyError(f"model_config is missing {missing}")
        cfg = model_config
        self.num_nodes = cfg["num_nodes"]
        self.c_in = cfg["spatial_in_channels_base"]
        c_spatial = cfg["spatial_out_channels"] * cfg["spatial_heads"]
        if self.c_in + cfg["d_emb"] != c_spatial:
            raise ValueError("residual connection needs spatial_in_channels_base + d_emb == "
                             "spatial_out_channels * spatial_heads (tec_mollm.py:94)")
        if cfg["d_llm"] != 768:
            raise ValueError("d_llm must be 768 (GPT-2 hidden size)")
        if len(cfg["temporal_strides"]) != 2 or len(cfg["temporal_channel_list"]) != 2:
            raise ValueError("exactly two temporal conv blocks are supported (tec_mollm.py:51)")
        self.precision = cfg.get("precision", "auto")
        if self.precision not in ("auto", "fp32", "bf16", "bf16x3", "bf16x6"):
            raise ValueError("precision must be 'auto' (follow torch.autocast), 'fp32', 'bf16', 'bf16x3' or 'bf16x6'")
        self.gat_graphs = cfg.get("gat_graphs", "reference")
        if self.gat_graphs not in ("reference", "per_timestep"):
            raise ValueError("gat_graphs must be 'reference' or 'per_timestep'")

        self.spatio_temporal_embedding = SpatioTemporalEmbedding(
            d_emb=cfg["d_emb"], num_nodes=self.num_nodes, num_years=cfg.get("num_years", 13))
        self.spatial_encoder = SpatialEncoder(in_channels=c_spatial, out_channels=cfg["spatial_out_channels"],
                                              heads=cfg["spatial_heads"])
        self.spatial_encoder.gat_graphs = self.gat_graphs
        self.temporal_encoder = TemporalEncoder(in_channels=c_spatial, channel_list=cfg["temporal_channel_list"],
                                                strides=cfg["temporal_strides"], patch_len=cfg["patch_len"],
                                                d_llm=cfg["d_llm"])
        self.llm_backbone = LLMBackbone(num_layers_to_keep=cfg["llm_layers"],
                                        include_wte=cfg.get("include_wte", True),
                                        load_pretrained=cfg.get("load_pretrained_gpt2", True))
        conv_len = cfg["temporal_seq_len"] // (cfg["temporal_strides"][0] * cfg["temporal_strides"][1])
        if conv_len % cfg["patch_len"] != 0:
            raise ValueError("patch_len must divide temporal_seq_len // (s0*s1) (train.py:255-260)")
        self.num_patches = conv_len // cfg["patch_len"]
        self.prediction_head = PredictionHead(input_dim=cfg["d_llm"] * self.num_patches,
                                              output_dim=cfg["prediction_horizon"])
        self.c_spatial = c_spatial
        self.heads = cfg["spatial_heads"]

    def forward(self, x: torch.Tensor, time_features: torch.Tensor, edge_index: torch.Tensor,
                edge_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x (B, L_in, N, C_in) f32, time_features (B, L_in, N, 4) f32 (may be a stride-0 expanded view),
        edge_index (2, E) int64, edge_weight ignored -> (B, L_out, N, 1)."""
        _need_cuda(x, "x")
        _need_cuda(time_features, "time_features")
        B, L, N, Cin = x.shape
        if N != self.num_nodes or Cin != self.c_in:
            raise ValueError(f"x has (N, C_in) = ({N}, {Cin}), model was built for ({self.num_nodes}, {self.c_in})")
        if time_features.shape != (B, L, N, 4):
            raise ValueError(f"time_features must be (B, L, N, 4), got {tuple(time_features.shape)}")
        plan = make_plan(self, precision=self.precision)
        meta = graph_.get(edge_index, N, x.device, self.spatio_temporal_embedding.d_emb)
        R = 1 if self.gat_graphs == "reference" else B * L
        # 1-4. embedding + GATv2 + residual  -> (B, L, N, 24) time-major
        xs = F_.SpatialFn.apply(x, time_features, *self.spatio_temporal_embedding.tables(),
                                *self.spatial_encoder.params(), meta, self.heads, R, plan)
        # 5. temporal encoder (+ wpe and embd dropout of the GPT-2 front end) -> (B, P, N, 768)
        wpe = self.llm_backbone.trunk.wpe.weight
        h0 = self.temporal_encoder.forward_tm(xs, self.c_spatial, wpe, plan, need_dinp=True)
        # 6. GPT-2 blocks with LoRA; 7. dropout + prediction head -> (B, N, L_out)
        # (bf16 mode: ln_f hands the head its operand directly -- dropped, rounded, sequence-major; F_.GPT2StackFn)
        if int(plan.bf16) == F_.ops.PREC_BF16 and _fuse_head():
            plan = dataclasses.replace(plan, fuse_head=True)
        hid = self.llm_backbone.forward_tm(h0, plan)
        pred = self.prediction_head.forward_tm(hid, plan)
        # 8. (B, N, L_out) -> (B, L_out, N, 1), a permuted view like tec_mollm.py:123
        return pred.permute(0, 2, 1).unsqueeze(-1)
