#!/usr/bin/env python3
"""Training-loss curves of the precision modes on the same synthetic task (diagnostics): identical weights, data and
dropout masks (counter-based, so they do not depend on the arithmetic), N steps of the full train step at B=8."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from src.model import modules as M_
from src.model.tec_mollm import TEC_MoLLM
from tecmollm.synthetic import grid_graph, synthetic_batch
from tecmollm.train import TrainStep

steps = int(os.environ.get("STEPS", 120))
B, L, Lo, cin = 8, 48, 12, 10
dev = torch.device("cuda")
ei = grid_graph()[0].to(dev)
batches = []
for k in range(4):                                   # four batches cycled: a learnable (memorisable) target
    x, tf, y = synthetic_batch(B, L, 2911, cin, Lo, seed=100 + k)
    batches.append((x.to(dev), tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, L, 2911, 4), (0.5 * x[:, -Lo:, :, :1] + 0.1 * y).to(dev)))
curves = {}
for mode in ("fp32", "bf16x6", "bf16x3", "bf16"):
    cfg = {"num_nodes": 2911, "d_emb": 22 - cin, "spatial_in_channels_base": cin, "spatial_out_channels": 11,
           "spatial_heads": 2, "temporal_channel_list": [64, 128], "temporal_strides": [2, 2], "patch_len": 4, "d_llm": 768,
           "llm_layers": 3, "prediction_horizon": Lo, "temporal_seq_len": L, "num_years": 13, "gat_graphs": "per_timestep",
           "include_wte": False, "load_pretrained_gpt2": False, "precision": mode}
    torch.manual_seed(0)
    M_._seed_counter[0] = 0                          # same dropout masks in every mode
    model = TEC_MoLLM(cfg)
    with torch.no_grad():
        for blk in model.llm_backbone.trunk.h:
            blk.attn.c_attn.lora_B.default.weight.normal_(std=0.02)
    model = model.to(dev).train()
    ts = TrainStep(model, lr=3e-4)
    losses = []
    for s in range(steps):
        x, tf, y = batches[s % 4]
        losses.append(ts.step(x, tf, ei, None, y))
    curves[mode] = [float(v) for v in torch.stack(losses).cpu()]
    print(mode, " ".join(f"{v:.5f}" for v in curves[mode][::max(1, steps // 12)]), flush=True)
ref = torch.tensor(curves["fp32"])
for mode in ("bf16x6", "bf16x3", "bf16"):
    d = (torch.tensor(curves[mode]) - ref).abs() / ref.abs()
    print(f"{mode}: max relative deviation of the loss from the fp32 run over {steps} steps = {float(d.max()):.2e} (at step {int(d.argmax())}), last step {float(d[-1]):.2e}")
