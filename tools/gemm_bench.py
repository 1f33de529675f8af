#!/usr/bin/env python3
"""Micro-benchmark of tecm_gemm_f32 on the GPT-2 shapes of the TEC-MoLLM step (diagnostics, not the bench)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
import ctypes
from tecmollm import _lib
if os.environ.get('TECM_LIB'):
    h = ctypes.CDLL(os.environ['TECM_LIB']); h.tecm_gemm_f32.restype = ctypes.c_int; h.tecm_gemm_f32.argtypes = _lib.EXPORTS['tecm_gemm_f32'][1]; h.tecm_last_error.restype = ctypes.c_char_p; _lib._lib = h  # TECM_SKIP_BIND
from tecmollm import ops

dev = torch.device("cuda")
BF16 = os.environ.get('BF16', '0') == '1'
M = int(os.environ.get("M", 69864))
shapes = [("fc   KN", 3072, 768, ops.B_KN), ("proj KN", 768, 3072, ops.B_KN), ("qkv  KN", 2304, 800, ops.B_KN),
          ("cprj KN", 768, 768, ops.B_KN), ("da   NK", 3072, 768, ops.B_NK), ("du2  NK", 768, 3072, ops.B_NK),
          ("du   NK", 800, 2304, ops.B_NK)]
for name, N, K, bl in shapes:
    A = torch.randn(M, K, device=dev)
    B = torch.randn((K, N) if bl == ops.B_KN else (N, K), device=dev) * 0.05
    C = torch.empty(M, N, device=dev)
    ldb = N if bl == ops.B_KN else K
    for _ in range(2):
        ops.gemm(M, N, K, A, K, B, ldb, C, N, b_layout=bl, bf16=BF16)
    torch.cuda.synchronize()
    reps = 8
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(M, N, K, A, K, B, ldb, C, N, b_layout=bl, bf16=BF16)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name}  M={M} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
