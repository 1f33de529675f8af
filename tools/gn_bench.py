#!/usr/bin/env python3
"""GroupNorm(1)+GELU forward/backward kernels at the B=8 shapes (diagnostics): MODE=fp32 (all fp32), bf16out (fp32 y, bf16
act / dy, fp32 dact: round 3), y16 (everything bf16, compact act: round 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from tecmollm import ops
dev = torch.device("cuda")
B, N = 8, 2911
mode = os.environ.get("MODE", "y16")
for (L, Cout, ds) in [(48, 64, 2), (24, 128, 2)]:
    CT = 3 * Cout
    y = torch.randn(B, L, N, CT, device=dev)
    gamma, beta = torch.randn(CT, device=dev), torch.randn(CT, device=dev)
    stats = torch.empty(B * N * 3 * 2, device=dev)
    o16 = mode != "fp32"
    odt = torch.bfloat16 if o16 else torch.float32
    if mode == "y16":
        y = y.bfloat16()
        act = torch.empty(B, L // ds, N, CT, device=dev, dtype=odt)
        astr = ds
        dact = torch.randn(B, L // ds, N, CT, device=dev).bfloat16()
    else:
        act = torch.empty(B, L, N, CT, device=dev, dtype=odt)
        astr = 1
        dact = torch.randn(B, L // ds, N, CT, device=dev)
    dy = torch.empty(B, L, N, CT, device=dev, dtype=odt)
    for _ in range(3):
        ops.groupnorm_gelu_fwd(y, gamma, beta, act, stats, B, L, N, Cout, act_stride=astr)
        ops.groupnorm_gelu_bwd(dact, ds, y, gamma, beta, stats, dy, B, L, N, Cout)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(10): ops.groupnorm_gelu_fwd(y, gamma, beta, act, stats, B, L, N, Cout, act_stride=astr)
    e[1].record()
    for _ in range(10): ops.groupnorm_gelu_bwd(dact, ds, y, gamma, beta, stats, dy, B, L, N, Cout)
    e[2].record()
    torch.cuda.synchronize()
    f, b = e[0].elapsed_time(e[1]) / 10, e[1].elapsed_time(e[2]) / 10
    fb = y.numel() * y.element_size() + act.numel() * act.element_size()
    bb = y.numel() * y.element_size() + dact.numel() * dact.element_size() + dy.numel() * dy.element_size()
    print(f"{mode:8s} L={L} Cout={Cout}: fwd {f*1e3:7.1f} us ({fb/1e6:6.0f} MB, {fb/f/1e9:5.2f} TB/s)   bwd {b*1e3:7.1f} us "
          f"({bb/1e6:6.0f} MB, {bb/b/1e9:5.2f} TB/s)", flush=True)
