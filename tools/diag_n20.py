#!/usr/bin/env python3
"""Distribution of the element-wise bf16 model bar over seeds on the smallest test problem (N = 20, B = 2, train mode): is a
miss noise (the worst tensor moves) or systematic (one tensor, every seed)?   GRAD16 / Y16 = 0 switch storage points off on
BOTH sides (device env + oracle policy) for comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
from oracle import ref_cpu as R
from tests.parity import compare_forward_backward
for seed in (31, 32, 33, 34, 35):
    cfg = R.default_config(L_in=48, L_out=12, num_nodes=20)
    res = compare_forward_backward(cfg, B=2, grid=(4, 5), threshold_km=170.0, gat_graphs="per_timestep", seed=seed, train=True,
                                   precision="bf16")
    top = sorted(res["per_param"].items(), key=lambda kv: -kv[1][1])[:4]
    print(seed, "fwd", round(res["fwd_rel"], 4), "grad max-norm", round(res["grad_rel_max"], 4),
          [(k[-38:], round(v[1], 2)) for k, v in top], flush=True)
