#!/usr/bin/env python3
"""How fast does this GPU absorb plain stores / copies?  (context for the GEMM epilogue numbers in DESIGN.md)"""
import torch
dev = torch.device("cuda")
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb in (215, 429, 858, 1716):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    us = t(lambda: x.zero_())
    uc = t(lambda: y.copy_(x))
    print(f"{mb:5d} MB  fill {us:8.1f} us = {mb/us*1e-3*1e3:6.2f} TB/s   copy {uc:8.1f} us = {2*mb/uc:6.2f} TB/s (r+w)", flush=True)
