#!/usr/bin/env python3
"""The fused conv d-input kernel (csrc/conv_seq.hip) alone at the bench shapes (diagnostics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from tecmollm import ops
dev = torch.device("cuda")
F32 = os.environ.get('F32', '0') == '1'
for (B, Lc, N, cin, ld, Cout) in ((8, 48, 2911, 22, 24, 64), (8, 24, 2911, 64, 64, 128)):
    dy = torch.randn(B, Lc, N, 3 * Cout, device=dev)
    dy = dy if F32 else dy.bfloat16()
    ws = [torch.randn(Cout, cin, k, device=dev) * 0.05 for k in (3, 5, 7)]
    out = torch.empty(B, Lc, N, ld, device=dev)
    for _ in range(3):
        ops.conv_dx_bf16(dy, *ws, out, B, Lc, N, Cout, cin, ld)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv_dx_bf16(dy, *ws, out, B, Lc, N, Cout, cin, ld)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = (dy.numel() * dy.element_size() + out.numel() * 4) / 1e9
    tf = 2.0 * B * Lc * N * ld * 15 * Cout / us / 1e6
    print(f"conv_dx Lc={Lc} Cout={Cout} ld_in={ld}: {us:7.1f} us  (pack included)  {gb / us * 1e6 / 1e3:5.2f} TB/s of dy + dinp  {tf:6.1f} TFLOP/s ({'fp32' if F32 else 'bf16'})", flush=True)
