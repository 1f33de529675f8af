#!/usr/bin/env python3
"""Can plain stores overlap a GEMM's K loop on this chip AT ALL?  Runs the epilogue-free ablation of the bf16 LDS-DMA
GEMM (build: python tools/build_variant.py gemm_bf16_dma.hip noepi5 -DDMA_ABLATE_NOEPI; TECM_LIB selects it) on one
stream and a plain fill of the bytes its epilogue would have written on a second stream, alone and together.
together ~ max(alone) -> the memory system lets stores ride under K loops, the kernel's structure is what serialises them;
together ~ sum(alone) -> store traffic itself stalls the operand stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from tecmollm import ops

dev = torch.device("cuda")
M, N, K = (int(v) for v in os.environ.get("SHAPE", "69864,3072,768").split(","))
MB = int(os.environ.get("FILL_MB", "858"))
R = int(os.environ.get("REPS", "20"))
A = torch.randn(M, K, device=dev).bfloat16()
B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
fill = torch.empty(MB * 1000 * 1000 // 4, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def gemm():
    ops.gemm(M, N, K, A, K, B, K, Cc, N, bf16=1)


def wall(fn_a, reps_a, fn_b, reps_b):
    torch.cuda.synchronize()
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    if fn_a:
        with torch.cuda.stream(s1):
            for _ in range(reps_a): fn_a()
            ea.record()
    if fn_b:
        with torch.cuda.stream(s2):
            for _ in range(reps_b): fn_b()
            eb.record()
    torch.cuda.synchronize()
    ta = e0.elapsed_time(ea) / reps_a * 1e3 if fn_a else 0.0
    tb = e0.elapsed_time(eb) / reps_b * 1e3 if fn_b else 0.0
    return ta, tb


zero = lambda: fill.zero_()
for _ in range(2):                                     # warm both streams (first use of a stream costs milliseconds)
    wall(gemm, 3, zero, 3)
ga, _ = wall(gemm, R, None, 0)
_, fb = wall(None, 0, zero, R)
rb = max(1, int(R * ga / fb))                          # enough fills to keep the second stream busy for the whole GEMM run
ta, tb = wall(gemm, R, zero, rb)
print(f"lib={os.environ.get('TECM_LIB', 'default')} DMA={os.environ.get('TECM_BF16_DMA', '')} M={M} N={N} K={K} fill={MB} MB")
print(f"  GEMM alone      {ga:8.1f} us/launch")
print(f"  fill alone      {fb:8.1f} us/launch  ({MB / fb:.2f} TB/s)")
print(f"  together        GEMM {ta:8.1f} us/launch (x{ta / ga:.2f}), fill {tb:8.1f} us/launch (x{tb / fb:.2f}, {MB / tb:.2f} TB/s) "
      f"[{R} GEMMs beside {rb} fills]")
print(f"  one GEMM + one fill of its output: serial {ga + fb:.1f} us, perfectly overlapped {max(ga, fb):.1f} us, "
      f"measured-rate equivalent {1.0 / (1.0 / ta + 0.0):.1f} us per GEMM while {MB / tb * ta / 1e0:.0f} MB are filled beside it")
