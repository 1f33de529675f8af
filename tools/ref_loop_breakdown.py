#!/usr/bin/env python3
"""Where the reference's loop body (train.py:57-112) loses time against tecmollm.train.TrainStep around the same drop-in
model: bench.reference_loop with single statements left out (B = 8 and B = 2, bf16 autocast), and the native step beside it.
    python tools/ref_loop_breakdown.py > gpurun_out/ref_loop_breakdown.txt"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
from tecmollm.synthetic import grid_graph

args = argparse.Namespace(batch=8, L_in=48, L_out=12, c_in=10, llm_layers=3, gat="per_timestep", precision="bf16", eval_mode=False)
cfg = bench.make_config(args)
dev = torch.device("cuda", 0)
ei, ew = grid_graph(); ei, ew = ei.to(dev), ew.to(dev)
for B in (8, 2):
    nat = bench.extra_config(cfg, args, dev, "bf16", ei, ew, "native", with_roofline=False, batch=B, steps=20)
    print(f"B={B} native TrainStep                      {nat['ms_per_step']:7.2f} ms/step")
    for skip in ((), ("empty_cache",), ("empty_cache", "item"), ("empty_cache", "item", "scaler"), ("empty_cache", "item", "scaler", "clip")):
        r = bench.reference_loop(cfg, args, dev, ei, ew, B, steps=20, skip=skip)
        print(f"B={B} reference loop without {','.join(skip) or '-':32s} {r['ms_per_step']:7.2f} ms/step", flush=True)
