#!/usr/bin/env python3
"""tecm_gemm_f32/bf16 on arbitrary shapes: SHAPES="M,N,K,kn|nk;..." (diagnostics, not the bench)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from tecmollm import ops

dev = torch.device("cuda")
BF16 = int(os.environ.get('BF16', '0'))       # 0 exact fp32, 1 bf16, 2 bf16x3
ops.GROUP_M = int(os.environ.get('GROUP_M', '0'))
for spec in os.environ.get("SHAPES", "16384,2048,16384,kn").split(";"):
    M, N, K, lay = spec.split(",")
    M, N, K = int(M), int(N), int(K)
    bl = ops.B_KN if lay == "kn" else ops.B_NK
    A = torch.randn(M, K, device=dev)
    B = torch.randn((K, N) if bl == ops.B_KN else (N, K), device=dev) * 0.05
    C = torch.empty(M, N, device=dev)
    R16 = os.environ.get("RES16", "")            # which tensors live in HBM as bf16: any of "a", "b", "c" (bf16 mode, nk only)
    if "a" in R16: A = A.bfloat16()
    if "b" in R16: B = B.bfloat16()
    if "c" in R16: C = C.bfloat16()
    ldb = N if bl == ops.B_KN else K
    EPI = os.environ.get("EPI", "")
    kw = {}
    if "bias" in EPI: kw["bias"] = torch.randn(N, device=dev)
    if "gelu" in EPI: kw["act"] = ops.ACT_GELU_TANH
    p16 = torch.bfloat16 if "p" in R16 else torch.float32     # "p": the pre-activation / GELU' source lives as bf16 (the step's form)
    if "preact" in EPI: kw["preact"] = (torch.empty(M, N, device=dev, dtype=p16), N)
    if "dact" in EPI: kw["dact_src"] = (torch.randn(M, N, device=dev).to(p16), N)
    if "resid" in EPI: kw["residual"] = (torch.randn(M, N, device=dev), N)
    if "drop" in EPI: kw["out_drop"] = ops.drop(0.1, 1234, N)
    for _ in range(2):
        ops.gemm(M, N, K, A, K, B, ldb, C, N, b_layout=bl, bf16=BF16, **kw)
    torch.cuda.synchronize()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(M, N, K, A, K, B, ldb, C, N, b_layout=bl, bf16=BF16, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{lay} M={M} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
