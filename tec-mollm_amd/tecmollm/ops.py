"""Thin, typed wrappers over the C ABI (include/tecmollm.h).  Every function launches HIP kernels
asynchronously on torch's current stream; tensors are only used as device-memory handles."""
from __future__ import annotations

import os
import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import TecmDrop, TecmGemm, TecmWin, check, lib, ptr, stream_ptr

A_MK, A_KM = 0, 1
B_NK, B_KN = 0, 1
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH = 0, 1, 2

_MASK64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


def win(N: int, Lin: int, Lout: int, stride_t: int, taps: int, Cw: int, pad: int) -> TecmWin:
    return TecmWin(1, N, Lin, Lout, stride_t, taps, Cw, pad)


NO_WIN = TecmWin(0, 0, 0, 0, 0, 0, 0, 0)


# The step's dropout word (TecmDrop::seed_dev): a device uint64 every mask-drawing kernel adds to its recorded seed.  None
# outside a captured step (tecmollm/train.py: TrainStep.step_graphed sets it while it records and replays).
SEED_WORD: Optional[torch.Tensor] = None


def drop(p: float, seed: int, ld: int) -> TecmDrop:
    return TecmDrop(float(p), 0, seed & _MASK64, int(ld), SEED_WORD.data_ptr() if SEED_WORD is not None else None)


NO_DROP = TecmDrop(0.0, 0, 0, 0, None)


def _off(t: torch.Tensor, col_off: int = 0) -> int:
    return t.data_ptr() + t.element_size() * col_off


def gemm(M: int, N: int, K: int, A: torch.Tensor, lda: int, B: torch.Tensor, ldb: int, Cout: torch.Tensor, ldc: int,
         *, a_layout: int = A_MK, b_layout: int = B_NK, a_off: int = 0, b_off: int = 0, c_off: int = 0,
         a_win: Optional[TecmWin] = None, b_win: Optional[TecmWin] = None, c_win: Optional[TecmWin] = None,
         a_drop: Optional[TecmDrop] = None, b_drop: Optional[TecmDrop] = None, out_drop: Optional[TecmDrop] = None,
         alpha: float = 1.0, act: int = ACT_NONE, bias: Optional[torch.Tensor] = None,
         rowbias: Optional[Tuple[torch.Tensor, int, int, int]] = None,
         preact: Optional[Tuple[torch.Tensor, int]] = None, dact_src: Optional[Tuple[torch.Tensor, int]] = None,
         residual: Optional[Tuple[torch.Tensor, int]] = None, accumulate: bool = False, split_k: int = 1,
         bf16: bool = False) -> None:
    # A / B / Cout may be torch.bfloat16 tensors (bf16 mode only, plain MK x NK): operands that already live in HBM as
    # bf16 (activations rounded once by their producer, cached bf16 weights) and / or a bf16 output for such a consumer
    """C[M,N] = epilogue(alpha * A_view[M,K] . B_view[K,N]); see TecmGemm in include/tecmollm.h.
    *_off are element offsets added to the base pointers (column slices of wider buffers).
    bf16 is the precision code: 0/False exact fp32, 1/True bf16 matrix cores, 2 bf16x3 (split-bf16, ~1e-5),
    3 bf16x6 (three-way split, fp32-grade).
    bf16=True asks for the bf16 matrix cores (operands rounded to bf16, fp32 accumulate); calls the bf16
    kernel cannot serve (N < 64 output columns or operands that are not 16-byte friendly) run on the exact
    fp32 kernel instead -- the choice is a pure function of shapes/alignment (`uses_bf16`), never silent
    with respect to the tests, which mirror it."""
    g = TecmGemm()
    g.M, g.N, g.K = M, N, K
    io = (IO_A_BF16 if A.dtype == torch.bfloat16 else 0) | (IO_B_BF16 if B.dtype == torch.bfloat16 else 0) | \
         (IO_C_BF16 if Cout.dtype == torch.bfloat16 else 0)
    pre_dt = {t[0].dtype for t in (preact, dact_src) if t is not None}     # ONE flag covers both pointers: they must agree
    if len(pre_dt) > 1:
        raise ValueError("gemm: preact and dact_src must share a dtype")
    if pre_dt == {torch.bfloat16}:
        io |= IO_PRE_BF16
    if io:
        if int(bf16) != PREC_BF16:
            raise _lib.TecmError("bf16 tensors are accepted by the bf16 GEMM only")
        g.io_bf16 = io
    g.A, g.lda, g.a_layout = _off(A, a_off), lda, a_layout
    g.B, g.ldb, g.b_layout = _off(B, b_off), ldb, b_layout
    g.C, g.ldc = _off(Cout, c_off), ldc
    g.a_win = a_win or NO_WIN
    g.b_win = b_win or NO_WIN
    g.c_win = c_win or NO_WIN
    g.a_drop = a_drop or NO_DROP
    g.b_drop = b_drop or NO_DROP
    g.out_drop = out_drop or NO_DROP
    g.alpha, g.act = alpha, act
    g.bias = ptr(bias)
    if rowbias is not None:
        rb, rb_ld, rb_div, rb_mod = rowbias
        g.rowbias, g.rb_ld, g.rb_div, g.rb_mod = rb.data_ptr(), rb_ld, rb_div, rb_mod
    if preact is not None:
        g.preact, g.ldp = preact[0].data_ptr(), preact[1]
    if dact_src is not None:
        g.dact_src, g.ldd = dact_src[0].data_ptr(), dact_src[1]
    if residual is not None:
        g.residual, g.ldr = residual[0].data_ptr(), residual[1]
    g.accumulate = 1 if accumulate else 0
    g._p1 = GROUP_M
    ws = None
    if split_k > 1:
        ws = torch.empty(split_k * M * N, device=Cout.device, dtype=torch.float32)
        g.split_k, g.workspace = split_k, ws.data_ptr()
    else:
        g.split_k = 1
    mode = int(bf16)                                    # 0 exact fp32, 1 bf16 (autocast semantics), 2 bf16x3
    use16 = mode == PREC_BF16 and (io != 0 or _bf16_ok(g))
    use3 = mode in (PREC_BF16X3, PREC_BF16X6) and _x3_ok(g)
    fn, what = ((lib().tecm_gemm_bf16x6, "tecm_gemm_bf16x6") if (use3 and mode == PREC_BF16X6) else
                (lib().tecm_gemm_bf16x3, "tecm_gemm_bf16x3") if use3 else
                (lib().tecm_gemm_bf16, "tecm_gemm_bf16") if use16 else (lib().tecm_gemm_f32, "tecm_gemm_f32"))
    if _timing is None:
        check(fn(C.byref(g), stream_ptr()), what)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(C.byref(g), stream_ptr()), what)
    e1.record()
    name = ("gemm_x3_kernel<3,16>" if mode == PREC_BF16X6 else "gemm_x3_kernel<2,32>") if use3 else _kernel_name(g, use16)
    if _timing_detail:
        name += (f" M={M} N={N} K={K} win={g.a_win.enabled}{g.b_win.enabled}{g.c_win.enabled}"
                 f" drop={int(g.a_drop.p > 0)}{int(g.b_drop.p > 0)}{int(g.out_drop.p > 0)} split={g.split_k}"
                 f" act={g.act} acc={g.accumulate} dact={int(dact_src is not None)} res={int(residual is not None)}"
                 f" pre={int(preact is not None)}")
    # algorithmic HBM bytes of the call: every operand / result element once (a windowed operand: its source tensor once)
    ea, eb, ec = A.element_size(), B.element_size(), Cout.element_size()
    nbytes = (min(A.numel(), M * K) if g.a_win.enabled else M * K) * ea + (min(B.numel(), N * K) if g.b_win.enabled else N * K) * eb
    nbytes += M * N * ec * (2 if accumulate else 1)
    for t in (preact, dact_src, residual):
        if t is not None:
            nbytes += M * N * t[0].element_size()
    _timing.append((name, 2.0 * M * N * K, e0, e1, float(nbytes)))


PREC_FP32, PREC_BF16, PREC_BF16X3, PREC_BF16X6 = 0, 1, 2, 3
IO_A_BF16, IO_B_BF16, IO_C_BF16, IO_PRE_BF16 = 1, 2, 4, 8       # TecmGemm.io_bf16
GROUP_M = int(os.environ.get("TECM_GROUP_M", "0"))   # m-tiles per L2 super-tile (0 = each kernel's default); tools/ sweep it
BF16_MIN_N = 64


def uses_bf16(N: int, K: int, lda: int, ldb: int, a_layout: int = A_MK, b_layout: int = B_NK, cw_a: int = 4,
              cw_b: int = 4) -> bool:
    """Which GEMMs run on the bf16 matrix cores when bf16 is requested (mirrored by the oracle's emulation)."""
    if N < BF16_MIN_N or lda % 4 or ldb % 4 or cw_a % 4 or cw_b % 4:
        return False
    if a_layout == A_MK and K % 4:
        return False
    return True


def uses_x3(N: int, K: int, lda: int, ldb: int, a_layout: int = A_MK, b_layout: int = B_NK, a_win: bool = False,
            b_win: bool = False, a_drop: bool = False, b_drop: bool = False) -> bool:
    """Which GEMMs the bf16x3 (split-bf16) kernel serves when that mode is requested: the plain MK x NK contraction
    with at least 64 output columns; every other call runs on the exact fp32 kernel."""
    if a_layout != A_MK or b_layout != B_NK or a_win or b_win or a_drop or b_drop:
        return False
    return N >= BF16_MIN_N and lda % 4 == 0 and ldb % 4 == 0 and K % 4 == 0


def _x3_ok(g: TecmGemm) -> bool:
    if not uses_x3(g.N, g.K, g.lda, g.ldb, g.a_layout, g.b_layout, bool(g.a_win.enabled), bool(g.b_win.enabled),
                   g.a_drop.p > 0, g.b_drop.p > 0):
        return False
    return g.A % 16 == 0 and g.B % 16 == 0


def _bf16_ok(g: TecmGemm) -> bool:
    if not uses_bf16(g.N, g.K, g.lda, g.ldb, g.a_layout, g.b_layout, g.a_win.Cw if g.a_win.enabled else 4,
                     g.b_win.Cw if g.b_win.enabled else 4):
        return False
    return g.A % 16 == 0 and g.B % 16 == 0


# ------------------------------------------------------------------ per-launch timing (bench.py roofline)
_timing = None
_timing_detail = False


def _vec(p: int, ld: int, w: TecmWin, inner_is_k: bool, K: int) -> int:
    v = 4
    while v > 1:
        ok = p % (4 * v) == 0 and ld % v == 0
        if w.enabled:
            ok = ok and w.Cw % v == 0
        if inner_is_k:
            ok = ok and K % v == 0
        if ok:
            break
        v >>= 1
    return v


def _kernel_name(g: TecmGemm, use16: bool = False) -> str:
    """Name of the template instance csrc/gemm.hip dispatches to (mirrors pick_vec / dispatch_vec)."""
    if use16:
        win16 = "true" if (g.a_win.enabled or g.b_win.enabled) else "false"
        drp16 = "true" if (g.a_drop.p > 0 or g.b_drop.p > 0) else "false"
        if (g.io_bf16 & (IO_A_BF16 | IO_B_BF16)) and not (g.a_layout == A_MK and g.b_layout == B_NK and win16 == "false"):
            win16 = "true"                             # the bf16-resident window / transposed instances are built WIN = true
        both16 = (g.io_bf16 & IO_A_BF16) and (g.io_bf16 & IO_B_BF16)
        # the straight-line epilogues (gemm_impl.h epi_fast_mode >= 0; mirrors tecm_gemm16_dma_try's fast_epi)
        streams = (1 if g.residual else 0) + (1 if g.dact_src else 0) + (1 if g.accumulate else 0)
        if g.c_win.enabled:
            fast_epi = streams == 0 and not g.rowbias and not g.preact and not (g.io_bf16 & (IO_C_BF16 | IO_PRE_BF16))
        else:
            fast_epi = streams == 0 if g.rowbias else streams <= 1
        p8 = os.environ.get("TECM_BF16_P8", "")[:1]                            # mirrors tecm_gemm16_p8_try (gemm_bf16_p8.hip)

        def p8_takes(a_rows: int) -> bool:
            return (fast_epi and not os.environ.get("TECM_BF16_DMA", "") and p8 != "0" and g.K >= 128 and g.K % 32 == 0
                    and g.M >= 256 and g.N >= 128 and (p8 == "1" or g.N >= 768 or not 1 <= g.N % 256 <= 128)
                    and (a_rows * g.lda + 64) * 2 < 2 ** 32 and (g.N * g.ldb + 64) * 2 < 2 ** 32)
        if (both16 and g.a_layout == A_MK and g.b_layout == B_NK and g.a_win.enabled and not g.b_win.enabled and drp16 == "false"
                and g.a_win.pad == 0 and g.a_win.Cw % 64 == 0 and g.split_k <= 1 and g.M >= 256 and g.N >= 128
                and os.environ.get("TECM_BF16_DMA", "")[:1] != "0"):
            w = g.a_win                                    # the window route of tecm_gemm16_dma_try
            if ((w.Lout - 1) * w.stride_t + w.taps <= w.Lin and g.M % (w.Lout * w.N) == 0
                    and p8_takes(g.M // (w.Lout * w.N) * w.Lin * w.N)):
                return f"gemm_bf16_p8_kernel<{lib().tecm_p8_rows(g.M, g.N)}>"
            return "gemm_bf16_dma_kernel"
        # mirrors tecm_gemm16_dma_try (csrc/gemm_bf16_dma.hip); the float4-epilogue condition holds for every bf16 call
        sel = os.environ.get("TECM_BF16_DMA", "")[:1]
        if (both16 and g.a_layout == A_MK and g.b_layout == B_NK and not g.a_win.enabled and not g.b_win.enabled
                and g.split_k <= 1 and g.N % 4 == 0 and g.K % 32 == 0 and g.K >= 32 and g.M >= 256
                and (g.N >= 128 or g.N == 32) and sel != "0"):
            k32, n_small = g.K < 64, g.N < 128
            can16 = not g.c_win.enabled and not g.rowbias and streams <= 1
            if p8_takes(g.M):
                return f"gemm_bf16_p8_kernel<{lib().tecm_p8_rows(g.M, g.N)}>"
            narrow = n_small or (sel == "2" if (sel and not k32) else 1 <= g.N % 256 <= 128)   # the 256 x 128 geometry (N = 800)
            if narrow:
                return "gemm_bf16_dma2_kernel"
            ring = sel == "4"                                                 # the four-slot ring, anti-phase wave groups (A/B)
            can16 = not g.c_win.enabled and not g.rowbias and \
                (1 if g.residual else 0) + (1 if g.dact_src else 0) + (1 if g.accumulate else 0) <= 1
            if can16 and sel in ("6", "7"):                                   # four-wave blocks, two per CU (A/B)
                return "gemm_bf16_dma6_kernel<256,128,2,2>" if sel == "6" else "gemm_bf16_dma6_kernel<128,256,1,4>"
            want16 = sel in ("5", "8") if (sel and not k32) else True          # the ring with 16x16x32 MFMAs
            if can16 and want16:
                tn_ = (g.N + 255) // 256
                cost = lambda bm: ((((g.M + bm - 1) // bm) * tn_ + 255) // 256) * bm     # rounds of blocks on 256 CUs x tile height
                tall = True if sel == "8" else (False if (sel == "5" or os.environ.get("TECM_BF16_TALL", "")[:1] == "0")
                                                else cost(288) * 100 < cost(256) * 95)
                return "gemm_bf16_dma5w_kernel" if tall else "gemm_bf16_dma5_kernel"
            return "gemm_bf16_dma4_kernel" if ring else ("gemm_bf16_dma3_kernel" if sel == "3" else "gemm_bf16_dma_kernel")
        # mirrors tecm_gemm16_tn_try (csrc/gemm_bf16_tn.hip): weight gradients from bf16 tensors in their natural orientation
        if (both16 and g.a_layout == A_KM and g.b_layout == B_KN and g.split_k >= 2 and not g.a_win.enabled
                and (not g.b_win.enabled or g.b_win.pad == 0) and os.environ.get("TECM_BF16_TN", "")[:1] != "0"
                and lib().tecm_gemm_tn_splits(g.M, g.N, g.K) > 0):
            return "gemm_bf16_tn_kernel"
        return f"gemm_bf16_kernel<{g.a_layout},{g.b_layout},{win16},{drp16}>"
    av = _vec(g.A, g.lda, g.a_win, g.a_layout == A_MK, g.K)
    bv = _vec(g.B, g.ldb, g.b_win, g.b_layout == B_NK, g.K)
    if av == 4 and bv == 4:
        pair = (4, 4)
    elif g.a_layout == A_MK:
        pair = (2, 1) if av >= 2 else (1, 1)
    else:
        pair = (4, 2) if (av == 4 and bv >= 2) else (1, 1)
    bn = 32 if g.N <= 32 else (64 if g.N <= 64 else 128)
    m64 = ",64" if (g.a_layout == A_KM and g.M <= 64 and bn > 32 and pair == (4, 4)) else ",128"   # block rows
    if pair == (4, 4):                      # float4 loaders: one instance per (WIN, DROP); narrower vectors: the general one
        win = bool(g.a_win.enabled or g.b_win.enabled)
        drp = bool(g.a_drop.p > 0 or g.b_drop.p > 0)
    else:
        win = drp = True
    flags = f",{'true' if win else 'false'},{'true' if drp else 'false'}"
    return f"gemm_kernel<{g.a_layout},{g.b_layout},{pair[0]},{pair[1]},{bn}{flags}{m64}>"


def enable_gemm_timing(detail: bool = False) -> list:
    """Bracket every GEMM launch with events on the launch stream (torch's current stream).
    detail=True keys the records by shape / view / epilogue flags as well (tools/gemm_breakdown.py)."""
    global _timing, _timing_detail
    _timing = []
    _timing_detail = detail
    return _timing


def disable_gemm_timing() -> None:
    global _timing
    _timing = None


def summarize_gemm_timing(records: list) -> dict:
    torch.cuda.synchronize()
    agg: dict = {}
    for rec in records:
        name, flops, e0, e1 = rec[:4]
        a = agg.setdefault(name, {"ms": 0.0, "flops": 0.0, "n": 0, "bytes": 0.0})
        a["ms"] += e0.elapsed_time(e1)
        a["flops"] += flops
        a["bytes"] += rec[4] if len(rec) > 4 else 0.0
        a["n"] += 1
    return agg


def tn_ok(Mo: int, No: int, K: int) -> bool:
    """Does the natural-orientation bf16 weight-gradient kernel (csrc/gemm_bf16_tn.hip) serve an Mo x No output over K rows?"""
    return os.environ.get("TECM_BF16_TN", "")[:1] != "0" and lib().tecm_gemm_tn_splits(Mo, No, K) >= 2


def pick_split_k(Mo: int, No: int, K: int, target_blocks: int = 512, min_chunk: int = 512,
                 max_splits: int = 256, prec: int = 0) -> int:
    """Split the (huge) reduction dim of a weight-gradient GEMM so the grid fills 256 CUs; capped because every
    split costs one slab of partial sums that the reducer has to read back.  The fp32 kernel has 128 x 128 tiles and
    two blocks per CU; the bf16 kernel (prec = PREC_BF16 and at least 64 output columns) 256 x 128 tiles and ONE block
    per CU, so its grid should be one round of at most 256 blocks, not 270."""
    if int(prec) == PREC_BF16:
        tn = lib().tecm_gemm_tn_splits(Mo, No, K)      # the natural-orientation kernel's own choice for shapes it serves
        if tn >= 2:
            return int(tn)
    if int(prec) == PREC_BF16 and No >= BF16_MIN_N:
        tiles = ((Mo + 255) // 256) * ((No + 127) // 128)
        s = max(1, 256 // max(tiles, 1))
        return int(min(s, max(1, K // min_chunk), max_splits))
    tiles = ((Mo + 127) // 128) * ((No + 127) // 128 if No > 64 else 1)
    s = max(1, target_blocks // max(tiles, 1))
    s = min(s, max(1, K // min_chunk), max_splits)
    return int(s)


def layernorm_fwd(x: torch.Tensor, ldx: int, gamma: torch.Tensor, beta: torch.Tensor, y: Optional[torch.Tensor], ldy: int,
                  stats: torch.Tensor, M: int, D: int, eps: float = 1e-5, y_off: int = 0,
                  y16: Optional[torch.Tensor] = None, ldy16: int = 0, y16d: Optional[torch.Tensor] = None, ldy16d: int = 0,
                  drop16d: Optional[TecmDrop] = None, seq_major: Optional[Tuple[int, int]] = None) -> None:
    """y (fp32) and / or y16 (bf16 copy for a bf16 matrix-core GEMM, BASELINE configs[2]); y16d: a third, optional bf16
    output = dropout(y, drop16d) rounded -- the LoRA branch's input (peft lora_dropout, modules.py:181).  seq_major = (T, N):
    y16d is written sequence-major -- row (b, n, t) for the time-major row (b, t, n) -- the head's view(batch, -1) of ln_f's
    (dropped) output as a plain matrix; drop16d may then be NO_DROP (eval mode)."""
    for t, what in ((y16, "y16"), (y16d, "y16d")):
        if t is not None and t.dtype != torch.bfloat16:
            raise _lib.TecmError(f"layernorm_fwd: {what} must be a bfloat16 tensor")
    if y16d is not None and drop16d is None:
        raise _lib.TecmError("layernorm_fwd: y16d needs its dropout spec")
    sT, sN = seq_major if seq_major is not None else (0, 0)
    check(lib().tecm_layernorm_fwd(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(),
                                   None if y is None else _off(y, y_off), ldy, ptr(y16), ldy16, ptr(y16d), ldy16d,
                                   C.byref(drop16d) if drop16d is not None else None, sT, sN,
                                   stats.data_ptr(), M, D, eps, stream_ptr()), "tecm_layernorm_fwd")


def _ln_add(add) -> Optional["_lib.TecmLnAdd"]:
    """add = (dy2 (M, D) fp32 / bf16, ld, drop spec or None)."""
    if add is None:
        return None
    dy2, ld, dspec = add
    if dy2.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.TecmError("layernorm_bwd: dy2 is fp32 or bf16")
    return _lib.TecmLnAdd(dy2=dy2.data_ptr(), ld=ld, bf16=1 if dy2.dtype == torch.bfloat16 else 0,
                          drop=dspec if dspec is not None else NO_DROP)


def layernorm_bwd_blocks(M: int, D: int) -> int:
    nb = C.c_int32(0)
    check(lib().tecm_layernorm_bwd(None, 0, None, 0, None, None, None, None, None, 0, None, None, C.byref(nb), M, D,
                                   None, 0, None, None), "tecm_layernorm_bwd(query)")
    return nb.value


def layernorm_bwd(dy: torch.Tensor, lddy: int, x: torch.Tensor, ldx: int, gamma: torch.Tensor, stats: torch.Tensor,
                  dres: Optional[torch.Tensor], dx: torch.Tensor, M: int, D: int,
                  dx_masked: Optional[torch.Tensor] = None,
                  mask_drop: Optional[TecmDrop] = None, need_dgb: bool = True, add=None, dy_seq_major=None):
    """dx = dres + LN'(dy); optional dx_masked = dropout(dx, mask_drop) -- fp32, or bf16 when its only reader is a bf16
    GEMM.  dy: fp32 or bf16.  dy_seq_major = (T, N, drop spec or None): dy is the bf16 sequence-major matrix (rows (b, n, t))
    the head's first Linear returned, still in front of the post-LLM dropout whose mask is applied here.  add = (dy2, ld, drop): a second gradient stream of the same tensor, dy += dropmask * dy2 before
    the LayerNorm backward (the LoRA branch's input gradient through lora_dropout's backward).  Returns (dgamma, dbeta), or
    (None, None) when need_dgb is False (frozen LayerNorm: the per-block partials are not reduced)."""
    m16 = 1 if (dx_masked is not None and dx_masked.dtype == torch.bfloat16) else 0
    ad = _ln_add(add)
    dy16 = 1 if dy.dtype == torch.bfloat16 else 0        # bf16 mode: the gradient a bf16 GEMM returned for its input
    nb = layernorm_bwd_blocks(M, D)
    partials = torch.empty(nb, 2 * D, device=dx.device, dtype=torch.float32)
    nbc = C.c_int32(0)
    od = mask_drop if mask_drop is not None else NO_DROP
    dm = None
    if dy_seq_major is not None:
        mT, mN, mdrop = dy_seq_major
        dm = _lib.TecmLnDyMap(T=mT, N=mN, drop=mdrop if mdrop is not None else NO_DROP)
    check(lib().tecm_layernorm_bwd(dy.data_ptr(), lddy, x.data_ptr(), ldx, gamma.data_ptr(), stats.data_ptr(),
                                   ptr(dres), dx.data_ptr(), ptr(dx_masked), m16, C.byref(od), partials.data_ptr(),
                                   C.byref(nbc), M, D, C.byref(ad) if ad is not None else None, dy16,
                                   C.byref(dm) if dm is not None else None, stream_ptr()),
          "tecm_layernorm_bwd")
    if not need_dgb:
        return None, None
    dgb = colsum(partials, 2 * D, nb, 1, 1, 2 * D)
    return dgb[0, :D], dgb[0, D:]


GN_Y_BF16, GN_OUT_BF16, GN_DACT_BF16 = 1, 2, 4


def gn_reg_ok(L: int, N: int, Cout: int) -> bool:
    """True when the register-resident GroupNorm+GELU kernels (csrc/norm.hip: gn_reg_pairs with 4 or 8 waves per sequence,
    at most 9 float4 per lane) serve BOTH directions -- the only kernels that read / write bf16 activations.  Longer
    sequences (the reference's default L_in = 336) take the multi-pass fp32 kernels, and the conv block then keeps its
    activations fp32 in bf16 mode too."""
    quads = L * (3 * Cout // 4)
    if L * N * 3 * Cout >= 1 << 31:
        return False
    return any(quads % (64 * wps) == 0 and quads // (64 * wps) <= 9 for wps in (4, 8))


def _gn_io(y: torch.Tensor, out: torch.Tensor) -> int:
    if y.dtype == torch.bfloat16:
        if out.dtype != torch.bfloat16:
            raise _lib.TecmError("groupnorm_gelu: a bf16 y comes with bf16 act / dact / dy")
        return GN_Y_BF16 | GN_OUT_BF16
    if y.dtype != torch.float32:
        raise _lib.TecmError("groupnorm_gelu: y is fp32 or bf16")
    return GN_OUT_BF16 if out.dtype == torch.bfloat16 else 0


def gn_y16_ok(L: int, N: int, Cout: int) -> bool:
    """True when the all-bf16 GroupNorm + GELU kernels serve the sequence (the conv block then writes y as bf16)."""
    return os.environ.get("TECM_Y16", "1")[:1] != "0" and lib().tecm_gn_y16_supported(L, N, Cout) == 1


GN_STATS_GIVEN = 8


def groupnorm_gelu_fwd(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, act: torch.Tensor,
                       stats: torch.Tensor, B: int, L: int, N: int, Cout: int, eps: float = 1e-5, act_stride: int = 1,
                       stats_given: bool = False) -> None:
    """act: fp32, or bf16 in bf16 mode (its dtype says which); y is fp32.  act_stride s > 1: act is the COMPACT
    (B, ceil(L / s), N, 3*Cout) tensor of the time steps t % s == 0 (all the stride-s 1x1 conv behind it reads).
    stats_given (all-bf16 form only): `stats` already holds (mean, rstd) -- conv_fwd(..., stats=) computed them from the y it
    wrote -- and the call is an elementwise pass over the time steps act keeps."""
    La = (L + act_stride - 1) // act_stride
    if act.numel() != B * La * N * 3 * Cout:
        raise _lib.TecmError(f"groupnorm_gelu_fwd: act must hold (B, {La}, N, 3*Cout) values for act_stride {act_stride}")
    io = _gn_io(y, act)
    if stats_given:
        if io != (GN_Y_BF16 | GN_OUT_BF16):
            raise _lib.TecmError("groupnorm_gelu_fwd: given statistics go with a bf16 y and a bf16 act")
        io |= GN_STATS_GIVEN
    check(lib().tecm_groupnorm_gelu_fwd(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act.data_ptr(),
                                        stats.data_ptr(), B, L, N, Cout, eps, io, act_stride, stream_ptr()),
          "tecm_groupnorm_gelu_fwd")


def groupnorm_gelu_bwd(dact: torch.Tensor, dstride: int, y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                       stats: torch.Tensor, dy: torch.Tensor, B: int, L: int, N: int,
                       Cout: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns (dgamma, dbeta, colsum(dy)), each (3*Cout,); the last is the conv-bias gradient.  dact: fp32, or (bf16 mode,
    together with a bf16 dy) the bf16 tensor the strided 1x1 conv's d-input GEMM wrote."""
    nb = C.c_int32(0)
    check(lib().tecm_groupnorm_gelu_bwd(None, dstride, None, None, None, None, None, None, C.byref(nb), B, L, N, Cout,
                                        0, None, None), "tecm_groupnorm_gelu_bwd(query)")
    CT = 3 * Cout
    partials = torch.empty(nb.value, 3 * CT, device=dy.device, dtype=torch.float32)
    io = _gn_io(y, dy)
    if dact.dtype == torch.bfloat16:
        if not io:
            raise _lib.TecmError("groupnorm_gelu_bwd: a bf16 dact comes with a bf16 dy")
        io |= GN_DACT_BF16
    elif io & GN_Y_BF16:
        raise _lib.TecmError("groupnorm_gelu_bwd: a bf16 y comes with a bf16 dact")
    # all-bf16 form: a small workspace for the per-sequence sums lets the library run the two streaming kernels
    sums = torch.empty(B * N * 6, device=dy.device, dtype=torch.float32) if (io & GN_Y_BF16) else None
    check(lib().tecm_groupnorm_gelu_bwd(dact.data_ptr(), dstride, y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                        stats.data_ptr(), dy.data_ptr(), partials.data_ptr(), C.byref(nb), B, L, N,
                                        Cout, io, ptr(sums), stream_ptr()), "tecm_groupnorm_gelu_bwd")
    dgb = colsum(partials, 3 * CT, nb.value, 1, 1, 3 * CT)
    return dgb[0, :CT], dgb[0, CT:2 * CT], dgb[0, 2 * CT:]


ATT_OUT_BF16, ATT_QKV_BF16, ATT_DCTX_BF16 = 1, 2, 4      # io_bf16 of tecm_attention_fwd / _bwd


def attention_fwd(qkv: torch.Tensor, ctx: torch.Tensor, B: int, T: int, N: int, heads: int, D: int,
                  prob_drop: Optional[TecmDrop] = None) -> None:
    pd = prob_drop if prob_drop is not None else NO_DROP
    io = (ATT_OUT_BF16 if ctx.dtype == torch.bfloat16 else 0) | (ATT_QKV_BF16 if qkv.dtype == torch.bfloat16 else 0)
    check(lib().tecm_attention_fwd(qkv.data_ptr(), ctx.data_ptr(), io, B, T, N, heads, D, C.byref(pd), stream_ptr()),
          "tecm_attention_fwd")


def attention_bwd(qkv: torch.Tensor, dctx: torch.Tensor, dqkv: torch.Tensor, B: int, T: int, N: int, heads: int,
                  D: int, prob_drop: Optional[TecmDrop] = None) -> None:
    pd = prob_drop if prob_drop is not None else NO_DROP
    io = (ATT_OUT_BF16 if dqkv.dtype == torch.bfloat16 else 0) | (ATT_QKV_BF16 if qkv.dtype == torch.bfloat16 else 0) | \
        (ATT_DCTX_BF16 if dctx.dtype == torch.bfloat16 else 0)
    check(lib().tecm_attention_bwd(qkv.data_ptr(), dctx.data_ptr(), dqkv.data_ptr(), io, B, T, N, heads, D, C.byref(pd),
                                   stream_ptr()), "tecm_attention_bwd")


def colsum(inp: torch.Tensor, ld: int, outer: int, inner: int, nseg: int, Cn: int, *, in_off: int = 0,
           scale: float = 1.0, in_drop: Optional[TecmDrop] = None, out: Optional[torch.Tensor] = None,
           accumulate: bool = False, twin: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[s][c] = scale * sum_{o<outer, j<inner} in[((o*nseg + s)*inner + j)*ld + c]  -> (nseg, Cn).
    twin: a contiguous bf16 (rows, Cn) tensor that receives the (masked) input values from the same pass."""
    if out is None:
        out = torch.empty(nseg, Cn, device=inp.device, dtype=torch.float32)
    ws = torch.empty(1024 * nseg * Cn, device=inp.device, dtype=torch.float32)     # contract: include/tecmollm.h
    idr = in_drop if in_drop is not None else NO_DROP
    if twin is not None and not (Cn % 4 == 0 and Cn <= 1024 and ld % 4 == 0):
        # wider than the one-pass kernel's 256 column quads (the head at L_in = 96: 1152 columns): two passes
        if in_drop is not None or ld != Cn or in_off:
            raise _lib.TecmError("colsum: a bf16 twin of a masked or strided input needs Cn % 4 == 0 and Cn <= 1024")
        cast_bf16(inp, Cn, twin, Cn, outer * nseg * inner, Cn)
        twin = None
    if twin is not None:
        if twin.dtype != torch.bfloat16 or twin.numel() != outer * nseg * inner * Cn:
            raise _lib.TecmError("colsum: twin must be a contiguous bf16 tensor of the input's rows x Cn")
        check(lib().tecm_colsum_twin(_off(inp, in_off), ld, outer, inner, nseg, Cn, out.data_ptr(), Cn, 1 if accumulate else 0,
                                     scale, C.byref(idr), ws.data_ptr(), twin.data_ptr(), Cn, stream_ptr()), "tecm_colsum_twin")
        return out
    check(lib().tecm_colsum(_off(inp, in_off), ld, outer, inner, nseg, Cn, out.data_ptr(), Cn, 1 if accumulate else 0,
                            scale, C.byref(idr), ws.data_ptr(), stream_ptr()), "tecm_colsum")
    return out


def cast_bf16(src: torch.Tensor, lds: int, dst: torch.Tensor, ldd: int, rows: int, cols: int, src_off: int = 0,
              dst_off: int = 0) -> None:
    """dst (bf16) [r][dst_off + c] = round(src (fp32) [r][src_off + c]) for c < cols."""
    check(lib().tecm_cast_bf16(src.data_ptr() + 4 * src_off, lds, dst.data_ptr() + 2 * dst_off, ldd, rows, cols,
                               stream_ptr()), "tecm_cast_bf16")


def dropout_apply(src: torch.Tensor, rows: int, cols: int, spec: TecmDrop, ld: Optional[int] = None,
                  out_bf16: bool = False, twin_bf16: bool = False):
    """dropout(src) with the library's counter-based mask (index = row*spec.ld + col) as a new contiguous tensor; out_bf16:
    a bf16 one (the masked value rounded once, for a tensor only bf16 contractions read); twin_bf16: (fp32 result, its bf16
    twin) from one pass."""
    dst = torch.empty_like(src, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    twin = torch.empty_like(src, dtype=torch.bfloat16) if twin_bf16 else None
    ld = cols if ld is None else ld
    check(lib().tecm_dropout_apply(src.data_ptr(), ld, dst.data_ptr(), ld, 1 if out_bf16 else 0, ptr(twin), ld, rows, cols,
                                   C.byref(spec), stream_ptr()), "tecm_dropout_apply")
    return (dst, twin) if twin_bf16 else dst


def weight_bf16(W: torch.Tensor, same: bool = True, transposed: bool = False):
    """(W rounded to bf16 [R][C] or None, W^T rounded to bf16 [C][R] or None) of a contiguous fp32 matrix, one launch."""
    R, Cc = W.shape
    s = torch.empty(R, Cc, device=W.device, dtype=torch.bfloat16) if same else None
    t = torch.empty(Cc, R, device=W.device, dtype=torch.bfloat16) if transposed else None
    check(lib().tecm_weight_bf16(W.data_ptr(), Cc, ptr(s), Cc, ptr(t), R, R, Cc, stream_ptr()), "tecm_weight_bf16")
    return s, t


def bf16_twin(src: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """A bf16 copy of a contiguous fp32 (rows, cols) matrix for bf16 contractions that would round it in their loaders
    anyway (bit-identical) -- the natural-orientation weight-gradient kernel takes its operands by LDS-DMA, bf16 only."""
    dst = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    cast_bf16(src, cols, dst, cols, rows, cols)
    return dst


def huber_fwd_bwd(pred: torch.Tensor, target: torch.Tensor, delta: float = 1.0, grad_scale: float = 1.0,
                  want_grad: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    n = pred.numel()
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred) if want_grad else None
    ws = torch.empty(1024, device=pred.device, dtype=torch.float32)
    check(lib().tecm_huber_fwd_bwd(pred.data_ptr(), target.data_ptr(), ptr(dpred), loss.data_ptr(), n, delta,
                                   grad_scale, ws.data_ptr(), stream_ptr()), "tecm_huber_fwd_bwd")
    return loss, dpred


def huber_fwd_bwd_strided(pred: torch.Tensor, target: torch.Tensor, delta: float = 1.0, grad_scale: float = 1.0,
                          want_grad: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """HuberLoss(delta, mean) of pred (B, H, N, 1) against target (same shape), each in its own layout (any strides): no
    contiguous copies.  Returns (loss (1,), dpred with pred's shape AND strides -- or None)."""
    if pred.shape != target.shape or pred.dim() != 4 or pred.shape[3] != 1:
        raise _lib.TecmError("huber_fwd_bwd_strided: pred and target are (B, L_out, N, 1) tensors of one shape")
    _lib.require_gpu_tensor(pred, "pred")
    _lib.require_gpu_tensor(target, "target")
    B, H, N, _ = pred.shape
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_strided(pred.shape, pred.stride(), device=pred.device, dtype=torch.float32) if want_grad else None
    if want_grad and max(st * (sz - 1) for st, sz in zip(pred.stride(), pred.shape)) + 1 > pred.numel():
        raise _lib.TecmError("huber_fwd_bwd_strided: pred must be a dense (permuted) tensor")
    ws = torch.empty(1024, device=pred.device, dtype=torch.float32)
    ps = (C.c_int64 * 3)(*pred.stride()[:3])
    tst = (C.c_int64 * 3)(*target.stride()[:3])
    check(lib().tecm_huber_fwd_bwd_strided(pred.data_ptr(), ps, target.data_ptr(), tst, ptr(dpred), loss.data_ptr(), B, H, N,
                                           delta, grad_scale, ws.data_ptr(), stream_ptr()), "tecm_huber_fwd_bwd_strided")
    return loss, dpred


def lora_fold(lB: torch.Tensor, scale: float, w_kn: Optional[torch.Tensor], w_nk: Optional[torch.Tensor], k_off: int) -> None:
    """The LoRA rows of [ W ; s B^T ] (w_kn: (k_off + r, n)) and / or the LoRA columns of [ W^T | s B ] (w_nk: (n, k_off + r))
    from lora_B (n, r); either operand fp32 or bf16, in place."""
    n, r = lB.shape
    for w, shape in ((w_kn, (k_off + r, n)), (w_nk, (n, k_off + r))):
        if w is not None and (tuple(w.shape) != shape or not w.is_contiguous() or w.dtype not in (torch.float32, torch.bfloat16)):
            raise _lib.TecmError(f"lora_fold: operand must be a contiguous {shape} fp32 / bf16 tensor")
    check(lib().tecm_lora_fold(lB.data_ptr(), n, r, float(scale), ptr(w_kn), n, int(w_kn is not None and w_kn.dtype == torch.bfloat16),
                               ptr(w_nk), k_off + r, int(w_nk is not None and w_nk.dtype == torch.bfloat16), k_off,
                               stream_ptr()), "tecm_lora_fold")


def pack_vectors(vecs, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch.cat of up to 12 small contiguous fp32 device vectors in ONE launch."""
    vecs = [v.detach() for v in vecs]
    if any(v.dtype != torch.float32 or not v.is_contiguous() or v.dim() != 1 for v in vecs) or not 1 <= len(vecs) <= 12:
        raise _lib.TecmError("pack_vectors: 1..12 contiguous fp32 vectors")
    total = sum(v.numel() for v in vecs)
    if out is None:
        out = torch.empty(total, device=vecs[0].device, dtype=torch.float32)
    srcs = (C.c_void_p * len(vecs))(*[v.data_ptr() for v in vecs])
    lens = (C.c_int32 * len(vecs))(*[v.numel() for v in vecs])
    check(lib().tecm_pack_vectors(srcs, lens, len(vecs), out.data_ptr(), stream_ptr()), "tecm_pack_vectors")
    return out


def conv_weight_pack(w: torch.Tensor, want_bwd: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    Cout, Cin, k = w.shape
    fwd = torch.empty(Cout, k * Cin, device=w.device, dtype=torch.float32)
    bwd = torch.empty(k * Cout, Cin, device=w.device, dtype=torch.float32) if want_bwd else None
    check(lib().tecm_conv_weight_pack(w.data_ptr(), fwd.data_ptr(), ptr(bwd), Cout, Cin, k, stream_ptr()),
          "tecm_conv_weight_pack")
    return fwd, bwd


def conv_weight_unpack(dpack: torch.Tensor, Cout: int, Cin: int, k: int) -> torch.Tensor:
    dw = torch.empty(Cout, Cin, k, device=dpack.device, dtype=torch.float32)
    check(lib().tecm_conv_weight_unpack(dpack.data_ptr(), dw.data_ptr(), Cout, Cin, k, stream_ptr()),
          "tecm_conv_weight_unpack")
    return dw


def conv_dx_seq_ok(Lc: int, Cout: int, ld_in: int, f32: bool = False) -> bool:
    """Shapes the sequence-tile d-input kernel (csrc/conv_seq.hip) serves IN THIS PRECISION -- divisibility and the LDS
    bytes of one tile, asked of the library itself (tecm_conv_dx_supported) so that the two can never disagree; everything
    else takes the window-view GEMMs."""
    return (os.environ.get("TECM_CONV_SEQ", "1")[:1] != "0"
            and lib().tecm_conv_dx_supported(Lc, Cout, ld_in, 1 if f32 else 0) == 1)


def conv_dx(dy: torch.Tensor, w3: torch.Tensor, w5: torch.Tensor, w7: torch.Tensor, dinp: torch.Tensor, B: int,
            Lc: int, N: int, Cout: int, cin: int, ld_in: int) -> None:
    """dinp (B, Lc, N, ld_in) fp32 = input gradient of the three parallel Conv1d of a Multi_Scale_Conv_Block
    (modules.py:43-60) from dy (B, Lc, N, 3*Cout), in one launch that reads dy once (csrc/conv_seq.hip).  dy bf16: the
    bf16 mode's arithmetic (weights rounded to bf16, fp32 accumulate); dy fp32: exact fp32."""
    if dinp.dtype != torch.float32 or dy.dtype not in (torch.bfloat16, torch.float32):
        raise _lib.TecmError("conv_dx: dy is bf16 or fp32, dinp fp32")
    f32 = dy.dtype == torch.float32
    nci = (ld_in + 31) // 32
    wpack = torch.empty(15 * Cout * nci * 32, device=dy.device, dtype=dy.dtype)
    pack, run, what = ((lib().tecm_conv_dx_pack_f32, lib().tecm_conv_dx_f32, "tecm_conv_dx_f32") if f32 else
                       (lib().tecm_conv_dx_pack, lib().tecm_conv_dx_bf16, "tecm_conv_dx_bf16"))
    check(pack(w3.data_ptr(), w5.data_ptr(), w7.data_ptr(), wpack.data_ptr(), Cout, cin, ld_in, stream_ptr()), what + "/pack")
    d = _lib.TecmConvDx(dy=dy.data_ptr(), wpack=wpack.data_ptr(), dinp=dinp.data_ptr(), B=B, Lc=Lc, N=N, Cout=Cout,
                        ld_in=ld_in)
    if _timing is None:
        check(run(C.byref(d), stream_ptr()), what)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(run(C.byref(d), stream_ptr()), what)
    e1.record()
    name = ("conv_dx_seq_f32_kernel" if f32 else "conv_dx_seq_kernel") + \
        (f" M={B * Lc * N} N={ld_in} K={15 * Cout}" if _timing_detail else "")
    _timing.append((name, 2.0 * B * Lc * N * ld_in * 15 * Cout, e0, e1))


conv_dx_bf16 = conv_dx      # round-3 name (tests)


def conv_dw_seq_ok(Lc: int, Cout: int, ld_in: int) -> bool:
    """Shapes the sequence-tile weight-gradient kernel (csrc/conv_dw_seq.hip) serves."""
    return (Lc % 4 == 0 and Lc > 0 and Cout in (64, 128) and ld_in in (24, 64)
            and os.environ.get("TECM_CONV_DW_SEQ", "1")[:1] != "0")


_cu_count = {}


def conv_dw(inp: torch.Tensor, dy: torch.Tensor, B: int, Lc: int, N: int, Cout: int, cin: int, ld_in: int):
    """(dw3, dw5, dw7), fp32 (Cout, cin, k): weight gradients of the three parallel Conv1d of a Multi_Scale_Conv_Block
    (modules.py:43-60) from the block input (B, Lc, N, ld_in) and dy (B, Lc, N, 3*Cout), in one persistent launch that
    reads both once (csrc/conv_dw_seq.hip) + a fixed-order reduction of the per-block slabs.  Both bf16: the bf16 mode's
    arithmetic; both fp32: exact fp32."""
    if inp.dtype != dy.dtype or dy.dtype not in (torch.bfloat16, torch.float32):
        raise _lib.TecmError("conv_dw: inp and dy are both bf16 or both fp32 tensors")
    f32 = dy.dtype == torch.float32
    run, what = (lib().tecm_conv_dw_f32, "tecm_conv_dw_f32") if f32 else (lib().tecm_conv_dw_bf16, "tecm_conv_dw_bf16")
    dev = dy.device
    nb = _cu_count.get(dev.index)
    if nb is None:
        nb = _cu_count[dev.index] = int(os.environ.get("TECM_CONV_DW_BLOCKS", 0)) or \
            torch.cuda.get_device_properties(dev).multi_processor_count
    nws = lib().tecm_conv_dw_workspace(Cout, ld_in, nb)
    if nws <= 0:
        raise _lib.TecmError(f"conv_dw: no kernel for Cout={Cout}, ld_in={ld_in}")
    ws = torch.empty(nws, device=dev, dtype=torch.float32)
    dws = [torch.empty(Cout, cin, k, device=dev, dtype=torch.float32) for k in (3, 5, 7)]
    d = _lib.TecmConvDw(inp=inp.data_ptr(), dy=dy.data_ptr(), workspace=ws.data_ptr(), dw3=dws[0].data_ptr(),
                        dw5=dws[1].data_ptr(), dw7=dws[2].data_ptr(), B=B, Lc=Lc, N=N, Cout=Cout, Cin=cin, ld_in=ld_in,
                        num_blocks=nb)
    if _timing is None:
        check(run(C.byref(d), stream_ptr()), what)
        return dws
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(run(C.byref(d), stream_ptr()), what)
    e1.record()
    name = ("conv_dw_seq_f32_kernel" if f32 else "conv_dw_seq_kernel") + (f" M={15 * ld_in} N={Cout} K={B * Lc * N}" if _timing_detail else "")
    _timing.append((name, 2.0 * B * Lc * N * ld_in * 15 * Cout, e0, e1))
    return dws


def conv_fwd_seq_ok(Lc: int, Cout: int, ld_in: int, f32: bool = False) -> bool:
    """As conv_dx_seq_ok, for the forward kernel (tecm_conv_fwd_supported: 64 KiB of LDS per tile)."""
    return (os.environ.get("TECM_CONV_SEQ", "1")[:1] != "0" and os.environ.get("TECM_CONV_FWD_SEQ", "1")[:1] != "0"
            and lib().tecm_conv_fwd_supported(Lc, Cout, ld_in, 1 if f32 else 0) == 1)


def conv_fwd_stats_ok(Lc: int) -> bool:
    """conv_fwd can compute the GroupNorm statistics when one tile holds a whole sequence (csrc/conv_seq.hip: 48 time steps)."""
    return Lc <= 48 and os.environ.get("TECM_CONV_STATS", "1")[:1] != "0"


def conv_fwd(inp: torch.Tensor, w3: torch.Tensor, w5: torch.Tensor, w7: torch.Tensor, bias: torch.Tensor,
             y: torch.Tensor, B: int, Lc: int, N: int, Cout: int, cin: int, ld_in: int,
             stats: Optional[torch.Tensor] = None, eps: float = 1e-5) -> None:
    """y (B, Lc, N, 3*Cout) fp32 = the three parallel Conv1d (k = 3, 5, 7) of a Multi_Scale_Conv_Block (modules.py:43-60)
    of inp (B, Lc, N, ld_in), bias (3*Cout) included, in one launch (csrc/conv_seq.hip).  inp bf16: the bf16 mode's
    arithmetic; inp fp32: exact fp32."""
    if inp.dtype not in (torch.bfloat16, torch.float32) or y.dtype not in (torch.bfloat16, torch.float32) or \
            (y.dtype == torch.bfloat16 and inp.dtype != torch.bfloat16):
        raise _lib.TecmError("conv_fwd: inp is bf16 or fp32, y fp32 (or bf16 next to a bf16 inp)")
    f32 = inp.dtype == torch.float32
    if f32:
        nval = sum(((3 + 2 * j) * ld_in // 8) * (Cout // 32) for j in range(3)) * 256
        pack, run, what = lib().tecm_conv_fwd_pack_f32, lib().tecm_conv_fwd_f32, "tecm_conv_fwd_f32"
    else:
        nval = sum((((3 + 2 * j) * ld_in + 15) // 16) * (Cout // 32) for j in range(3)) * 512
        pack, run, what = lib().tecm_conv_fwd_pack, lib().tecm_conv_fwd_bf16, "tecm_conv_fwd_bf16"
    wpack = torch.empty(nval, device=y.device, dtype=inp.dtype)
    check(pack(w3.data_ptr(), w5.data_ptr(), w7.data_ptr(), wpack.data_ptr(), Cout, cin, ld_in, stream_ptr()), what + "/pack")
    d = _lib.TecmConvFwd(inp=inp.data_ptr(), wpack=wpack.data_ptr(), bias=bias.data_ptr(), y=y.data_ptr(), B=B, Lc=Lc,
                         N=N, Cout=Cout, ld_in=ld_in, y_bf16=1 if y.dtype == torch.bfloat16 else 0,
                         stats=ptr(stats), eps=eps)     # stats: (B*N, 3, 2) GroupNorm(1) mean / rstd of the bf16 y (one tile per sequence)
    if _timing is None:
        check(run(C.byref(d), stream_ptr()), what)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(run(C.byref(d), stream_ptr()), what)
    e1.record()
    name = ("conv_fwd_seq_f32_kernel" if f32 else "conv_fwd_seq_kernel") + \
        (f" M={B * Lc * N} N={3 * Cout} K={5 * ld_in}(avg)" if _timing_detail else "")
    _timing.append((name, 2.0 * B * Lc * N * Cout * 15 * ld_in, e0, e1))


conv_fwd_bf16 = conv_fwd    # round-3 name (tests)


def transpose_scale(src: torch.Tensor, lds: int, dst: torch.Tensor, ldd: int, rows: int, cols: int, scale: float,
                    dst_off: int = 0) -> None:
    """dst[r*ldd + c] = scale * src[c*lds + r]."""
    check(lib().tecm_transpose_scale(src.data_ptr(), lds, _off(dst, dst_off), ldd, rows, cols, scale, stream_ptr()),
          "tecm_transpose_scale")
