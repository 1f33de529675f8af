#!/usr/bin/env python3
"""Per-call-site GEMM time of one B=8 training step (diagnostics): shape, views, epilogue flags, TFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from tecmollm import ops
from tecmollm.train import TrainStep
from src.model.tec_mollm import TEC_MoLLM
from tecmollm.synthetic import grid_graph, synthetic_batch

prec = os.environ.get("PRECISION", "fp32")
B, L, Lo, cin = 8, 48, 12, 10
cfg = {"num_nodes": 2911, "d_emb": 22 - cin, "spatial_in_channels_base": cin, "spatial_out_channels": 11,
       "spatial_heads": 2, "temporal_channel_list": [64, 128], "temporal_strides": [2, 2], "patch_len": 4, "d_llm": 768,
       "llm_layers": 3, "prediction_horizon": Lo, "temporal_seq_len": L, "num_years": 13, "gat_graphs": "per_timestep",
       "include_wte": False, "load_pretrained_gpt2": False, "precision": prec}
dev = torch.device("cuda")
torch.manual_seed(0)
model = TEC_MoLLM(cfg).to(dev).train()
x, tf, y = synthetic_batch(B, L, 2911, cin, Lo, seed=1)
x, y = x.to(dev), y.to(dev)
tf = tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, L, 2911, 4)
ei = grid_graph()[0].to(dev)
ts = TrainStep(model)
for _ in range(3):
    ts.step(x, tf, ei, None, y)
torch.cuda.synchronize()
rec = ops.enable_gemm_timing(detail=True)
steps = 3
for _ in range(steps):
    ts.step(x, tf, ei, None, y)
agg = ops.summarize_gemm_timing(rec)
ops.disable_gemm_timing()
tot = sum(a["ms"] for a in agg.values()) / steps
print(f"total GEMM {tot:.2f} ms/step")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    ms = a["ms"] / steps
    print(f"{ms:7.3f} ms {a['n']//steps:3d}x {a['flops']/a['ms']/1e9:7.1f} TF  {name}")
