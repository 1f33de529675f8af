#!/usr/bin/env python3
"""conv_fwd_seq kernels alone at the B = 8 shapes (PRECISION=bf16|fp32; TECM_LIB selects a variant build, e.g.
python tools/build_variant.py conv_seq.hip pd1 -DCFW_PD=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from tecmollm import ops

dev = torch.device("cuda")
f32 = os.environ.get("PRECISION", "bf16") == "fp32"
B, N = 8, 2911
for ld_in, cin, Cout, L in ((24, 22, 64, 48), (64, 64, 128, 24)):
    torch.manual_seed(0)
    inp = torch.randn(B, L, N, ld_in, device=dev)
    inp = inp if f32 else inp.bfloat16()
    w = [torch.randn(Cout, cin, k, device=dev) * 0.1 for k in (3, 5, 7)]
    bias = torch.randn(3 * Cout, device=dev)
    y = torch.empty(B, L, N, 3 * Cout, device=dev, dtype=torch.bfloat16 if os.environ.get('Y16', '0') == '1' else torch.float32)
    for _ in range(3):
        ops.conv_fwd(inp, w[0], w[1], w[2], bias, y, B, L, N, Cout, cin, ld_in)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.conv_fwd(inp, w[0], w[1], w[2], bias, y, B, L, N, Cout, cin, ld_in)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * B * L * N * Cout * 15 * cin
    print(f"{'fp32' if f32 else 'bf16'} Cout={Cout:3d} L={L}: {us:7.1f} us (incl. the weight pack)  {fl / us / 1e6:6.1f} TF  y {y.numel() * y.element_size() / us / 1e6:5.2f} TB/s", flush=True)
