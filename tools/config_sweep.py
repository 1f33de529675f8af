#!/usr/bin/env python3
"""Parity of one full training step (forward, loss, all gradients) against the CPU oracle over a sweep of model
configurations the reference's ctor accepts, in fp32 and bf16 mode -- a net for loud failures and mismatches in
configurations the timed ones do not touch (diagnostics; the shapes that matter are pinned by tests/)."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
from oracle import ref_cpu as R
from tests.parity import compare_forward_backward, assert_parity

cases = []
ONLY = os.environ.get("ONLY", "")
for L_in, L_out in ((48, 12), (16, 4), (32, 12), (40, 12), (24, 6), (80, 12), (96, 24), (8, 2)):
    cases.append((f"L_in={L_in} L_out={L_out}", dict(L_in=L_in, L_out=L_out), {}))
cases.append(("channels [128, 256]", dict(), {"temporal_channel_list": [128, 256]}))
cases.append(("channels [64, 64]", dict(), {"temporal_channel_list": [64, 64]}))
# shapes whose fp32 sequence tiles do NOT fit the LDS (tecm_conv_*_supported says no): the window-GEMM fallback serves them
cases.append(("channels [64, 256] L_in=96", dict(L_in=96, L_out=24), {"temporal_channel_list": [64, 256]}))
cases.append(("channels [128, 128] L_in=96", dict(L_in=96, L_out=24), {"temporal_channel_list": [128, 128]}))
cases.append(("strides [1, 2]", dict(), {"temporal_strides": [1, 2], "patch_len": 4}))
cases.append(("patch_len 2", dict(), {"patch_len": 2}))
cases.append(("c_in 10 d_emb 12", dict(c_in=10, d_emb=12), {}))
cases.append(("c_in 4 d_emb 8", dict(c_in=4, d_emb=8), {}))
cases.append(("llm_layers 1", dict(llm_layers=1), {}))
bad = 0
for name, kw, over in cases:
    if ONLY and ONLY not in name:
        continue
    for prec in (os.environ.get("PRECS", "fp32,bf16").split(",")):
        for train in (False, True):
            try:
                cfg = R.default_config(num_nodes=12, **kw)
                cfg.update(over)
                res = compare_forward_backward(cfg, B=2, grid=(3, 4), threshold_km=170.0, gat_graphs="per_timestep", seed=5,
                                               train=train, precision=prec)
                line = (f"{name:28s} {prec} {'train' if train else 'eval '}  fwd {res['fwd_rel']:.1e} (elem {res['fwd_elem']:.2f}) "
                        f"grad {res['grad_rel_max']:.1e} elem {res['grad_elem_max']:.2f} [{res['grad_elem_worst'][-40:]}]")
                try:
                    assert_parity(res)
                    print("ok    " + line, flush=True)
                except AssertionError:
                    bad += 1
                    print("OVER  " + line, flush=True)
            except Exception as e:  # noqa: BLE001
                bad += 1
                msg = str(e).splitlines()[0][:160] if str(e) else type(e).__name__
                print(f"FAIL  {name:28s} {prec} {'train' if train else 'eval '}  {type(e).__name__}: {msg}", flush=True)
print("failures:", bad)
