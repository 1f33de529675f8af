#!/usr/bin/env python3
"""GroupNorm(1)+GELU forward/backward kernels at the B=8 shapes (diagnostics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from tecmollm import ops
dev = torch.device("cuda")
B, N = 8, 2911
for (L, Cout, ds) in [(48, 64, 2), (24, 128, 2)]:
    CT = 3 * Cout
    y = torch.randn(B, L, N, CT, device=dev)
    gamma, beta = torch.randn(CT, device=dev), torch.randn(CT, device=dev)
    act = torch.empty_like(y); stats = torch.empty(B * N * 3 * 2, device=dev)
    dact = torch.randn(B, L // ds, N, CT, device=dev); dy = torch.empty_like(y)
    for _ in range(3):
        ops.groupnorm_gelu_fwd(y, gamma, beta, act, stats, B, L, N, Cout)
        ops.groupnorm_gelu_bwd(dact, ds, y, gamma, beta, stats, dy, B, L, N, Cout)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(10): ops.groupnorm_gelu_fwd(y, gamma, beta, act, stats, B, L, N, Cout)
    e[1].record()
    for _ in range(10): ops.groupnorm_gelu_bwd(dact, ds, y, gamma, beta, stats, dy, B, L, N, Cout)
    e[2].record()
    torch.cuda.synchronize()
    f, b = e[0].elapsed_time(e[1]) / 10, e[1].elapsed_time(e[2]) / 10
    nb = y.numel() * 4
    print(f"L={L} Cout={Cout}: fwd {f*1e3:7.1f} us ({2*nb/f/1e9:6.2f} TB/s)   bwd {b*1e3:7.1f} us ({(2*nb+dact.numel()*4)/b/1e9:6.2f} TB/s)", flush=True)
