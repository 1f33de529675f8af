#!/usr/bin/env python3
"""Which tensors carry the device <-> bf16-emulating-oracle gap of the L_in = 16 sanity configuration, with the conv
output y stored as bf16 (default) and as fp32 (TECM_Y16=0 on the device; the oracle follows through the device policy it is handed)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from oracle import ref_cpu as R
from tests.parity import build_model, device_rounding, oracle_step, rel_err
dev = torch.device("cuda")
for L_in, seed in ((16, 12), (48, 12), (16, 13)):
    cfg = R.default_config(L_in=L_in, L_out=12, num_nodes=12)
    p = R.init_params(cfg, seed=seed)
    x, tf, y = R.synthetic_batch(2, L_in, 12, 6, 12, seed=seed + 100)
    ei, _ = R.grid_graph(3, 4, threshold_km=170.0)
    for y16 in (True, False):
        os.environ["TECM_Y16"] = "1" if y16 else "0"
        model = build_model(cfg, p, dev, "per_timestep").eval()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x.to(dev), tf.to(dev), ei.to(dev))
            loss = torch.nn.functional.huber_loss(out.float(), y.to(dev))
        loss.backward()
        named = dict(model.named_parameters())
        o16, _, g16 = oracle_step(cfg, p, x, tf, ei, y, None, q=device_rounding(12))
        errs = sorted(((rel_err(named[k].grad, g), k) for k, g in g16.items() if g.abs().max() > 0), reverse=True)
        print(f"L_in={L_in} seed={seed} y16={y16}: fwd {rel_err(out, o16):.2e}  worst grads:", [(f"{e:.1e}", k[-45:]) for e, k in errs[:4]], flush=True)
