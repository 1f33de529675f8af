#!/usr/bin/env python3
"""conv_dw_seq.hip at the B = 8 shapes of both conv blocks: time per launch pair, implied HBM rate (diagnostics).
Phase ablations: python tools/build_variant.py conv_dw_seq.hip abl3 -DCDW_ABLATE=3; TECM_LIB=<that .so> python tools/conv_dw_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from tecmollm import ops

dev = torch.device("cuda")
B, N = 8, 2911
dt = torch.float32 if os.environ.get("PRECISION", "bf16") == "fp32" else torch.bfloat16
for Lc, cin, ld_in, Cout in ((48, 22, 24, 64), (24, 64, 64, 128)):
    x = torch.randn(B, Lc, N, ld_in, device=dev).to(dt)
    dy = torch.randn(B, Lc, N, 3 * Cout, device=dev).to(dt)
    for _ in range(3):
        ops.conv_dw(x, dy, B, Lc, N, Cout, cin, ld_in)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.conv_dw(x, dy, B, Lc, N, Cout, cin, ld_in)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    mb = (x.numel() + dy.numel()) * x.element_size() / 1e6
    tf = 2.0 * B * Lc * N * 15 * cin * Cout / us * 1e-6
    print(f"ld_in={ld_in} Cout={Cout} Lc={Lc}: {us:7.1f} us per call (kernel + reduce), operands {mb:.0f} MB -> {mb / us * 1e-3 * 1e3:.2f} GB/ms"
          f" = {mb / us:.2f} TB/s, {tf:.1f} TFLOP/s (real channels)   lib={os.path.basename(os.environ.get('TECM_LIB', 'default'))}")
