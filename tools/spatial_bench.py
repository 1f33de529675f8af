#!/usr/bin/env python3
"""Spatial (embedding + GATv2 + residual) forward/backward kernels at B=8 (diagnostics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from src.model.tec_mollm import TEC_MoLLM
from tecmollm import functions as F_, graph as graph_
from src.model.modules import make_plan
from tecmollm.synthetic import grid_graph, synthetic_batch
B, L, cin = int(os.environ.get("BATCH", "8")), 48, 10
cfg = {"num_nodes": 2911, "d_emb": 22 - cin, "spatial_in_channels_base": cin, "spatial_out_channels": 11,
       "spatial_heads": 2, "temporal_channel_list": [64, 128], "temporal_strides": [2, 2], "patch_len": 4, "d_llm": 768,
       "llm_layers": 1, "prediction_horizon": 12, "temporal_seq_len": L, "num_years": 13, "gat_graphs": "per_timestep",
       "include_wte": False, "load_pretrained_gpt2": False, "precision": "fp32"}
dev = torch.device("cuda")
model = TEC_MoLLM(cfg).to(dev).train(os.environ.get('EVAL', '0') != '1')
x, tf, y = synthetic_batch(B, L, 2911, cin, 12, seed=1)
x = x.to(dev); tf = tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, L, 2911, 4)
ei = grid_graph()[0].to(dev)
meta = graph_.get(ei, 2911, dev, cfg["d_emb"])
plan = make_plan(model, precision="fp32")
def run():
    xs = F_.SpatialFn.apply(x, tf, *model.spatio_temporal_embedding.tables(), *model.spatial_encoder.params(), meta, 2,
                            B * L, plan)
    return xs
go = torch.randn(B, L, 2911, 24, device=dev)
for _ in range(3):
    run().backward(go)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf_, tb_ = 0.0, 0.0
for _ in range(5):
    e[0].record(); o = run(); e[1].record(); o.backward(go); e[2].record(); torch.cuda.synchronize()
    tf_ += e[0].elapsed_time(e[1]); tb_ += e[1].elapsed_time(e[2])
print(f"spatial fwd {tf_/5*1e3:.0f} us   bwd (incl. partial reductions) {tb_/5*1e3:.0f} us", flush=True)
import ctypes
from tecmollm import _lib
h = ctypes.CDLL(_lib.LIB_PATH)
for sym, names in (("tecm_debug_spf_stamps", ["prologue", "stage", "dense-wait", "edge-wait", "store", "dense-own", "pf-commit", "edge-own", "e:xr", "e:loop", "e:out"]),
                   ("tecm_debug_spb_stamps", ["prologue", "stage", "dense", "B1", "B2", "outer", "flush"])):
    if hasattr(h, sym):
        buf = (ctypes.c_ulonglong * 16)()
        getattr(h, sym)(buf, 1)
        with torch.no_grad():
            for _ in range(4):
                run()
        if sym.endswith("spb_stamps"):
            for _ in range(4):
                run().backward(go)
        torch.cuda.synchronize()
        getattr(h, sym)(buf, 0)
        tot = sum(buf[:8]) or 1
        print(sym, {n: f"{buf[i] / tot:.2f}" for i, n in enumerate(names)}, "cycles(100MHz ticks?) total", tot, flush=True)
