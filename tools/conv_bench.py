#!/usr/bin/env python3
"""Per-GEMM timing of the two Multi_Scale_Conv_Blocks at the bench shapes (B=8, N=2911): forward and backward launches
with torch events, keyed by shape (diagnostics; TECM_LIB selects an ablation build, see tools/build_variant.py)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from tecmollm import ops, functions as F_
from src.model.modules import Multi_Scale_Conv_Block

dev = torch.device("cuda")
prec = int(os.environ.get("BF16", "0"))
B, N = 8, 2911
blocks = [(24, 22, 64, 48), (64, 64, 128, 24)]          # (ld_in, cin, Cout, L)
torch.manual_seed(0)
for ld_in, cin, Cout, L in blocks:
    blk = Multi_Scale_Conv_Block(cin, Cout, 2).to(dev)
    inp = torch.randn(B, L, N, ld_in, device=dev)
    if ld_in != cin:
        inp[..., cin:] = 0
    inp.requires_grad_(True)
    for it in range(3):
        if it == 2:
            rec = ops.enable_gemm_timing(detail=True)
        out, _ = blk.forward_tm(inp, cin, True, prec)
        out.backward(torch.randn_like(out))
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, flops, e0, e1 in (r[:4] for r in rec):
        a = agg.setdefault(name, [0.0, 0.0])
        a[0] += e0.elapsed_time(e1) * 1e3
        a[1] += flops
    ops.disable_gemm_timing()
    tot = 0.0
    print(f"--- block Cin={cin} Cout={Cout} L={L}")
    for k, (us, fl) in agg.items():
        tot += us
        print(f"{us:8.1f} us {fl / us / 1e6:7.1f} TF  {k[:150]}")
    print(f"{tot:8.1f} us total GEMM")
