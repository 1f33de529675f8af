"""torch.autograd.Function wrappers: one per stage of TEC_MoLLM.forward (tec_mollm.py:59-125).

PyTorch owns tensors, the autograd graph and the optimiser; every forward and backward here is a
sequence of launches of hand-written HIP kernels through the C ABI (ops.py).  Nothing in this file
computes on tensors with torch ops except allocation, tiny weight reshapes (zero-padding the 22->24
channel conv weights) and slicing of gradient buffers.

All activations after the spatial stage are time-major (B, T, N, C): row m = (b*T + t)*N + n.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
import weakref
from typing import List, Optional

import torch

from . import devcheck, ops
from ._lib import TecmSpatial, TecmSpatialGrads, check, lib, stream_ptr
from .graph import GraphMeta
from .ops import (A_KM, A_MK, ACT_GELU_ERF, ACT_GELU_TANH, B_KN, B_NK, colsum, drop, gemm, pick_split_k, win)

PRE16 = os.environ.get("TECM_PRE16", "1")[:1] != "0"     # diagnostics: "0" keeps the GPT-2 MLP pre-activation fp32
GRAD16 = os.environ.get("TECM_GRAD16", "1")[:1] != "0"   # diagnostics: "0" keeps d LN-out / d ctx fp32 in bf16 mode
QKV16 = os.environ.get("TECM_QKV16", "1")[:1] != "0"   # diagnostics: "0" keeps qkv fp32 in bf16 mode

import ctypes as C

CP = 24            # C = 22 feature channels padded to a multiple of 4 floats (16-byte rows)
C_FEAT = 22
LORA_R = 32
LORA_SCALE = 2.0   # lora_alpha / r = 64 / 32 (modules.py:177-183)
GPT_HEADS = 12

# dropout sites (each gets an independent seed derived from the per-forward base seed)
SITE_GAT, SITE_EMBD, SITE_POST, SITE_HEAD = 0, 1, 100, 101


def site_lora(i):
    return 10 + 4 * i


def site_attn(i):
    return 11 + 4 * i


def site_res1(i):
    return 12 + 4 * i


def site_res2(i):
    return 13 + 4 * i


@dataclass
class DropPlan:
    """Dropout configuration of one forward pass.  Masks are keep(seed_site, idx) recomputed in backward."""
    training: bool
    p: float
    base_seed: int
    bf16: bool = False      # run the dense contractions on the bf16 matrix cores (autocast semantics)
    fuse_head: bool = False  # bf16 mode, whole model: ln_f writes the head's operand directly (see GPT2StackFn / HeadFn)

    def spec(self, site: int, ld: int):
        if not self.training or self.p <= 0.0:
            return None
        return drop(self.p, ops.splitmix64(self.base_seed * 1000003 + site), ld)


def _lib_error(msg: str):
    from ._lib import TecmError
    return TecmError(msg)


def _empty(*shape, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


# ============================================================================ stage a-1..a-3
class SpatialFn(torch.autograd.Function):
    """SpatioTemporalEmbedding + GATv2Conv + residual (modules.py:230-266, :340-359; tec_mollm.py:84-94)."""

    @staticmethod
    def forward(ctx, x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab, Wl, bl, Wr, br, att, bias,
                meta: GraphMeta, heads: int, graphs_with_edges: int, plan: DropPlan):
        B, L, N, Cin = x.shape
        Demb = node_tab.shape[1]
        x = x.contiguous()
        out = _empty(B, L, N, CP, like=x)
        d = SpatialFn._desc(x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab, Wl, bl, Wr, br, att, bias, meta,
                            heads, graphs_with_edges, plan, B, L, N, Cin, Demb)
        d.out = out.data_ptr()
        errs = devcheck.error_word(x.device)
        errs.poll()                                     # a bad time index reported by an earlier forward (no sync)
        # round 5: the high-occupancy formulation (csrc/spatial_fwd2.hip) wherever it serves the call -- block-uniform time
        # features, the fused stage as TEC_MoLLM.forward runs it; TECM_SPATIAL_V2=0 keeps the persistent kernel (A/B)
        nws = lib().tecm_spatial_fwd2_ws_floats(C.byref(d)) if os.environ.get("TECM_SPATIAL_V2", "1")[:1] != "0" else 0
        if nws > 0:
            ws = torch.empty(nws, device=x.device, dtype=torch.float32)
            check(lib().tecm_spatial_fwd2(C.byref(d), ws.data_ptr(), stream_ptr()), "tecm_spatial_fwd2")
        else:
            check(lib().tecm_spatial_fwd(C.byref(d), stream_ptr()), "tecm_spatial_fwd")
        errs.post()
        ctx.save_for_backward(x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab, Wl, bl, Wr, br, att, bias)
        ctx.meta, ctx.heads, ctx.R, ctx.plan = meta, heads, graphs_with_edges, plan
        return out

    @staticmethod
    def _desc(x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab, Wl, bl, Wr, br, att, bias, meta, heads, R,
              plan, B, L, N, Cin, Demb) -> TecmSpatial:
        d = TecmSpatial()
        d.B, d.L, d.N, d.Cin, d.Demb, d.H = B, L, N, Cin, Demb, heads
        d.graphs_with_edges = min(int(R), B * L)
        d.num_tiles, d.tile_nodes, d.win_max = meta.num_tiles, meta.tile_nodes, meta.win_max
        d.tile_edges_max = meta.tile_edges_max
        d.x = x.data_ptr()
        if tf is not None:
            d.tf = tf.data_ptr()
            d.tf_sb, d.tf_sl, d.tf_sn, d.tf_sf = tf.stride()
        for name, t in (("node_tab", node_tab), ("tod_tab", tod_tab), ("doy_tab", doy_tab), ("year_tab", year_tab),
                        ("season_tab", season_tab), ("Wl", Wl), ("bl", bl), ("Wr", Wr), ("br", br), ("att", att),
                        ("bias", bias)):
            if t is not None:
                setattr(d, name, t.data_ptr())
        d.year_rows = year_tab.shape[0] if year_tab is not None else 0
        d.rowptr, d.colidx = meta.rowptr.data_ptr(), meta.colidx.data_ptr()
        d.tile_lo, d.tile_hi = meta.tile_lo.data_ptr(), meta.tile_hi.data_ptr()
        sp = plan.spec(SITE_GAT, meta.max_deg + 1)
        if sp is not None:
            d.alpha_drop = sp
        d.err_flag = devcheck.error_word(x.device).ptr()
        d.flags, d.out_ld = 0, CP
        return d

    @staticmethod
    def backward(ctx, dout):
        (x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab, Wl, bl, Wr, br, att, bias) = ctx.saved_tensors
        meta = ctx.meta
        B, L, N, Cin = x.shape
        Demb = node_tab.shape[1]
        dout = dout.contiguous()
        d = SpatialFn._desc(x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab, Wl, bl, Wr, br, att, bias, meta,
                            ctx.heads, ctx.R, ctx.plan, B, L, N, Cin, Demb)
        # round 5: the second formulation (csrc/spatial_bwd2.hip) wherever it serves the call; TECM_SPATIAL_V2=0 / TECM_SPATIAL_BWD2=0
        # keep the persistent kernel (A/B)
        v2 = os.environ.get("TECM_SPATIAL_V2", "1")[:1] != "0" and os.environ.get("TECM_SPATIAL_BWD2", "1")[:1] != "0"
        nb2 = lib().tecm_spatial_bwd2_blocks(C.byref(d)) if v2 else 0
        nblocks = nb2 if nb2 > 0 else lib().tecm_spatial_bwd_blocks(C.byref(d))   # rows of the partial-sum buffer
        if nblocks <= 0:
            check(nblocks, "tecm_spatial_bwd_blocks")
        Cc = C_FEAT
        pld = 2 * Cc * Cc + 4 * Cc
        partials = _empty(nblocks, pld, like=x)
        tabs = (node_tab, tod_tab, doy_tab, year_tab, season_tab)          # the kernel accumulates into them: ONE zero fill
        zbuf = torch.zeros(sum(t.numel() for t in tabs), device=x.device, dtype=torch.float32)
        offs = [0]
        for t in tabs:
            offs.append(offs[-1] + t.numel())
        d_node, d_tod, d_doy, d_year, d_season = (zbuf[a:b].view_as(t) for t, a, b in zip(tabs, offs[:-1], offs[1:]))
        g = TecmSpatialGrads()
        g.dout = dout.data_ptr()
        g.d_node_tab, g.d_tod_tab, g.d_doy_tab = d_node.data_ptr(), d_tod.data_ptr(), d_doy.data_ptr()
        g.d_year_tab, g.d_season_tab = d_year.data_ptr(), d_season.data_ptr()
        g.partials, g.partial_ld = partials.data_ptr(), pld
        g.t_chunk, g.num_blocks = 0, nblocks
        g.src_ptr, g.src_col = meta.src_ptr.data_ptr(), meta.src_col.data_ptr()
        g.src_ptr_off = meta.src_ptr_off.data_ptr()
        if nb2 > 0:
            ws = torch.empty(lib().tecm_spatial_fwd2_ws_floats(C.byref(d)), device=x.device, dtype=torch.float32)
            check(lib().tecm_spatial_bwd2(C.byref(d), C.byref(g), ws.data_ptr(), stream_ptr()), "tecm_spatial_bwd2")
        else:
            check(lib().tecm_spatial_bwd(C.byref(d), C.byref(g), stream_ptr()), "tecm_spatial_bwd")
        s = colsum(partials, pld, nblocks, 1, 1, pld)[0]
        o = 0
        dWl = s[o:o + Cc * Cc].view(Cc, Cc); o += Cc * Cc
        dbl = s[o:o + Cc]; o += Cc
        dWr = s[o:o + Cc * Cc].view(Cc, Cc); o += Cc * Cc
        dbr = s[o:o + Cc]; o += Cc
        datt = s[o:o + Cc].view_as(att)
        dbias = colsum(dout, CP, B * L * N, 1, 1, CP)[0, :Cc]   # d bias = column sums of dout (out = h + gat + bias);
        #                                                         all 24 padded columns: the float4 reduction path
        return (None, None, d_node, d_tod, d_doy, d_year, d_season, dWl, dbl, dWr, dbr, datt, dbias,
                None, None, None, None)


class EmbedFn(torch.autograd.Function):
    """Stand-alone SpatioTemporalEmbedding.forward (modules.py:230-266): (B, L, N, C_in) -> (B, L, N, C_in + d_emb),
    the embed-only mode of the fused spatial kernel.  Needs no graph: the node tiling fields are placeholders."""

    @staticmethod
    def forward(ctx, x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab):
        B, L, N, Cin = x.shape
        Demb = node_tab.shape[1]
        x = x.contiguous()
        out = _empty(B, L, N, Cin + Demb, like=x)
        d = EmbedFn._desc(x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab, B, L, N, Cin, Demb)
        d.out = out.data_ptr()
        errs = devcheck.error_word(x.device)
        errs.poll()
        check(lib().tecm_spatial_fwd(C.byref(d), stream_ptr()), "tecm_spatial_fwd(embed only)")
        errs.post()
        ctx.save_for_backward(x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab)
        return out

    @staticmethod
    def _desc(x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab, B, L, N, Cin, Demb) -> TecmSpatial:
        d = TecmSpatial()
        d.B, d.L, d.N, d.Cin, d.Demb, d.H = B, L, N, Cin, Demb, 2
        d.graphs_with_edges, d.num_tiles, d.tile_nodes, d.win_max, d.tile_edges_max = 0, N, 1, 1, 0
        d.x = x.data_ptr()
        d.tf = tf.data_ptr()
        d.tf_sb, d.tf_sl, d.tf_sn, d.tf_sf = tf.stride()
        for name, t in (("node_tab", node_tab), ("tod_tab", tod_tab), ("doy_tab", doy_tab), ("year_tab", year_tab),
                        ("season_tab", season_tab)):
            setattr(d, name, t.data_ptr())
        # the graph / GAT fields are not read in this mode; point them at a live buffer to satisfy the null checks
        for name in ("Wl", "bl", "Wr", "br", "att", "bias", "rowptr", "colidx", "tile_lo", "tile_hi"):
            setattr(d, name, node_tab.data_ptr())
        d.year_rows = year_tab.shape[0]
        d.err_flag = devcheck.error_word(x.device).ptr()
        d.flags, d.out_ld = 1, Cin + Demb                     # TECM_SPATIAL_EMBED_ONLY
        return d

    @staticmethod
    def backward(ctx, dout):
        x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab = ctx.saved_tensors
        B, L, N, Cin = x.shape
        Demb = node_tab.shape[1]
        dout = dout.contiguous()
        d = EmbedFn._desc(x, tf, node_tab, tod_tab, doy_tab, year_tab, season_tab, B, L, N, Cin, Demb)
        grads = [torch.zeros_like(t) for t in (node_tab, tod_tab, doy_tab, year_tab, season_tab)]
        g = TecmSpatialGrads()
        g.dout = dout.data_ptr()
        g.d_node_tab, g.d_tod_tab, g.d_doy_tab, g.d_year_tab, g.d_season_tab = (t.data_ptr() for t in grads)
        check(lib().tecm_spatial_bwd(C.byref(d), C.byref(g), stream_ptr()), "tecm_spatial_bwd(embed only)")
        dx = dout[..., :Cin] if ctx.needs_input_grad[0] else None
        return (dx, None, *grads)


class GatFn(torch.autograd.Function):
    """Stand-alone SpatialEncoder.forward (modules.py:340-359): x (G, N, 22) -> GATv2Conv(x) (G, N, 22), no residual,
    no embedding: the fused kernels with Demb = 0 (the input rows ARE h) and TECM_SPATIAL_NO_RESIDUAL.  Gradients for
    the GATv2 parameters; the input itself gets none (the fused path never needs d x), so x must not require grad."""

    @staticmethod
    def forward(ctx, x, Wl, bl, Wr, br, att, bias, meta: GraphMeta, heads: int, graphs_with_edges: int, plan: DropPlan):
        G, N, Cc = x.shape
        x = x.contiguous()
        out = _empty(G, 1, N, CP, like=x)
        d = GatFn._desc(x, Wl, bl, Wr, br, att, bias, meta, heads, graphs_with_edges, plan)
        d.out = out.data_ptr()
        check(lib().tecm_spatial_fwd(C.byref(d), stream_ptr()), "tecm_spatial_fwd(GAT only)")
        ctx.save_for_backward(x, Wl, bl, Wr, br, att, bias)
        ctx.meta, ctx.heads, ctx.R, ctx.plan = meta, heads, graphs_with_edges, plan
        return out.view(G, N, CP)[..., :C_FEAT]

    @staticmethod
    def _desc(x, Wl, bl, Wr, br, att, bias, meta, heads, R, plan) -> TecmSpatial:
        G, N, Cc = x.shape
        d = SpatialFn._desc(x, None, None, None, None, None, None, Wl, bl, Wr, br, att, bias, meta, heads, R, plan,
                            G, 1, N, Cc, 0)
        d.flags = 2                                           # TECM_SPATIAL_NO_RESIDUAL
        return d

    @staticmethod
    def backward(ctx, dout):
        x, Wl, bl, Wr, br, att, bias = ctx.saved_tensors
        G, N, Cc = x.shape
        if ctx.needs_input_grad[0]:
            raise _lib_error("SpatialEncoder.forward: the input needs no gradient in the MI355X path (x must not require grad)")
        dpad = torch.zeros(G, 1, N, CP, device=x.device, dtype=torch.float32)
        dpad[..., :Cc] = dout.reshape(G, 1, N, Cc)
        d = GatFn._desc(x, Wl, bl, Wr, br, att, bias, ctx.meta, ctx.heads, ctx.R, ctx.plan)
        nblocks = lib().tecm_spatial_bwd_blocks(C.byref(d))
        if nblocks <= 0:
            check(nblocks, "tecm_spatial_bwd_blocks")
        pld = 2 * Cc * Cc + 4 * Cc
        partials = _empty(nblocks, pld, like=x)
        g = TecmSpatialGrads()
        g.dout, g.partials, g.partial_ld, g.num_blocks = dpad.data_ptr(), partials.data_ptr(), pld, nblocks
        meta = ctx.meta
        g.src_ptr, g.src_col, g.src_ptr_off = meta.src_ptr.data_ptr(), meta.src_col.data_ptr(), meta.src_ptr_off.data_ptr()
        check(lib().tecm_spatial_bwd(C.byref(d), C.byref(g), stream_ptr()), "tecm_spatial_bwd(GAT only)")
        s = colsum(partials, pld, nblocks, 1, 1, pld)[0]
        o = 0
        dWl = s[o:o + Cc * Cc].view(Cc, Cc); o += Cc * Cc
        dbl = s[o:o + Cc]; o += Cc
        dWr = s[o:o + Cc * Cc].view(Cc, Cc); o += Cc * Cc
        dbr = s[o:o + Cc]; o += Cc
        datt = s[o:o + Cc].view_as(att)
        dbias = colsum(dpad, CP, G * N, 1, 1, CP)[0, :Cc]
        return (None, dWl, dbl, dWr, dbr, datt, dbias, None, None, None, None)


# ============================================================================ stage a-4
def conv_block_acts16(Lc: int, N: int, Cout: int) -> bool:
    """bf16 mode: does a conv block keep the activations behind its GroupNorm (and the gradient at the 1x1 conv's input) in
    HBM as bf16?  Only where the register-resident norm kernels -- the ones that read / write bf16 -- serve the sequence."""
    return Cout >= 64 and ops.gn_reg_ok(Lc, N, Cout)


def conv_block_y16(Lc: int, N: int, Cout: int, ld_in: int) -> bool:
    """bf16 mode: is the three-branch conv output y itself written as bf16 (by the sequence-tile forward kernel from a bf16
    input, read by the all-bf16 norm kernels)?  ld_in = channel pitch of the block's input tensor (spatial_ld(C) for block 0)."""
    CT = 3 * Cout
    return (conv_block_acts16(Lc, N, Cout) and ld_in % 8 == 0 and ops.conv_fwd_seq_ok(Lc, Cout, ld_in, f32=False)
            and ops.gn_y16_ok(Lc, N, Cout) and ops.uses_bf16(CT, Cout, Cout, CT, b_layout=B_KN))


def conv_storage_policy(N: int):
    """(L, Cout, Cin) -> (acts16, y16): the two functions above as ONE callable keyed the way the oracle sees a conv block
    (tests hand it to oracle.ref_cpu.Rounding.with_conv_policy; block 0's input is the padded spatial output)."""
    def policy(L: int, Cout: int, Cin: int):
        ld_in = CP if Cin == C_FEAT else Cin                  # block 0 reads the spatial output: 22 channels at a pitch of 24
        return conv_block_acts16(L, N, Cout), conv_block_y16(L, N, Cout, ld_in)
    return policy


class ConvBlockFn(torch.autograd.Function):
    """Multi_Scale_Conv_Block.forward (modules.py:43-60): three Conv1d(k=3,5,7)+GroupNorm(1)+GELU branches,
    channel concat, Conv1d(k=1, stride).  inp is (B, Lc, N, ld_in) time-major with `cin` real channels
    (ld_in - cin zero pad channels); returns (B, Lc//stride, N, Cout)."""

    @staticmethod
    def forward(ctx, inp, inp16, cin: int, stride: int, need_dinp: bool, bf16: bool,
                w3, b3, g3, be3, w5, b5, g5, be5, w7, b7, g7, be7, wf, bf):
        """inp16: optional bf16 copy of inp (bf16 mode: written by the producing block's last GEMM next to its fp32 output;
        only the forward window GEMMs read it).  Returns (out, out16): out16 is that copy of this block's output, or None."""
        B, Lc, N, ld_in = inp.shape
        Cout = w3.shape[0]
        CT = 3 * Cout
        M = B * Lc * N
        ws = (w3, w5, w7)
        bs = (b3, b5, b7)
        # bf16 mode: the activations behind the norm and (backward) dy live in HBM as bf16 -- they are only ever read by
        # bf16 contractions, which would round them in their loaders (same bits, half the bytes).  The conv output y
        # stays fp32 (the norm kernels got slower, not faster, reading 8-byte quads).
        r16 = int(bf16) == ops.PREC_BF16 and conv_block_acts16(Lc, N, Cout)
        adt = torch.bfloat16 if r16 else torch.float32
        side16 = r16
        packs = []
        # the three kernel sizes in ONE launch that stages the input rows once and writes whole rows of y
        # (csrc/conv_seq.hip): exact fp32 from the fp32 input, the bf16 mode's arithmetic from the bf16 copy of the input;
        # otherwise (bf16x3 / bf16x6 modes, odd shapes) three window GEMMs, one 64- / 128-column slice each
        seq_in = inp16 if (side16 and inp16 is not None) else (inp if int(bf16) == ops.PREC_FP32 else None)
        fwd_seq = seq_in is not None and ops.conv_fwd_seq_ok(Lc, Cout, ld_in, f32=seq_in.dtype == torch.float32)
        # bf16 mode: y itself is the bf16 tensor a bf16 Conv1d returns under autocast (train.py:68) -- written as such by the
        # sequence-tile kernel, read by the all-bf16 norm kernels forward and backward (half the bytes, three times over) --
        # wherever those kernels serve the sequence and the 1x1 conv's d-input GEMM returns the bf16 gradient they take
        y16 = r16 and fwd_seq and seq_in.dtype == torch.bfloat16 and conv_block_y16(Lc, N, Cout, ld_in)
        y = torch.empty(B, Lc, N, CT, device=inp.device, dtype=torch.bfloat16 if y16 else torch.float32)
        # bias | gamma | beta of the three branches as the 3*Cout vectors the kernels read: one launch, not three cats
        bgb = ops.pack_vectors([b3, b5, b7, g3, g5, g7, be3, be5, be7])
        bias3, gamma, beta = bgb[:CT], bgb[CT:2 * CT], bgb[2 * CT:]
        # ... and the kernel that produces the bf16 y also has every sequence whole in its registers: it hands over the
        # GroupNorm statistics, and the norm + GELU forward is an elementwise pass over the time steps act keeps
        stats = _empty(B * N, 3, 2, like=inp)
        st_given = y16 and ops.conv_fwd_stats_ok(Lc)
        if fwd_seq:
            ops.conv_fwd(seq_in.detach(), w3.detach(), w5.detach(), w7.detach(), bias3, y, B, Lc, N, Cout, cin, ld_in,
                         stats=stats if st_given else None)
        # the window-GEMM operands ([Cout][k*ld_in] forward, [k*Cout][ld_in] d-input) are only packed when a GEMM will read
        # them: the forward below, or a backward whose sequence-tile kernels do not serve this shape / precision
        dx_gemm = need_dinp and not ConvBlockFn._dx_seq(bf16, r16, Lc, Cout, ld_in)
        for j, (w, b) in enumerate(zip(ws, bs)):
            k = w.shape[2]
            if fwd_seq and not dx_gemm:
                packs.append(w.new_empty(0))
                continue
            wp = w if ld_in == cin else torch.nn.functional.pad(w, (0, 0, 0, ld_in - cin))
            fp, bp = ops.conv_weight_pack(wp.contiguous(), want_bwd=True)
            packs.append(bp)
            if fwd_seq:
                continue
            a_in = inp16 if (inp16 is not None and side16 and ld_in % 8 == 0) else inp
            gemm(M, Cout, k * ld_in, a_in, ld_in, fp, k * ld_in, y, CT, c_off=j * Cout,
                 a_win=win(N, Lc, Lc, 1, k, ld_in, (k - 1) // 2), bias=b, bf16=bf16)
        Lo = (Lc - 1) // stride + 1
        # the activation is only kept at the time steps the stride-s 1x1 conv reads (a COMPACT (B, Lo, N, CT) tensor: half
        # the bytes at stride 2, and that conv and its weight gradient become plain GEMMs, no window view) wherever the
        # register-resident norm kernels -- the ones that can skip rows -- serve the sequence
        compact = stride > 1 and ops.gn_reg_ok(Lc, N, Cout) and int(bf16) in (ops.PREC_FP32, ops.PREC_BF16)
        La = Lo if compact else Lc
        act = torch.empty(B, La, N, CT, device=inp.device, dtype=adt)
        ops.groupnorm_gelu_fwd(y, gamma, beta, act, stats, B, Lc, N, Cout, act_stride=stride if compact else 1,
                               stats_given=st_given)
        out = _empty(B, Lo, N, Cout, like=inp)
        wf2 = wf.view(Cout, CT)
        out16 = None
        awin = None if compact else win(N, Lc, Lo, stride, 1, CT, 0)
        # bf16 mode, compact bf16 activations: the 1x1 conv and its input gradient are plain contractions of bf16 tensors
        # once the weight is one too -- W and W^T rounded (one tiny launch; what the loaders would round W to), so that the
        # LDS-DMA GEMM serves them
        wfT16 = None
        if int(bf16) == ops.PREC_BF16 and compact and act.dtype == torch.bfloat16 and Cout % 8 == 0:
            wf2, wfT16 = ops.weight_bf16(wf2, same=True, transposed=True)
        ctx.wfT16 = wfT16
        if side16:
            # one epilogue, two forms of the same values: the fp32 tensor autograd sees (written through the epilogue's
            # pre-activation store; there is no activation here) and a bf16 copy for the window GEMMs that read it next
            out16 = torch.empty(B, Lo, N, Cout, device=inp.device, dtype=torch.bfloat16)
            gemm(B * Lo * N, Cout, CT, act, CT, wf2, CT, out16, Cout, a_win=awin, bias=bf, preact=(out, Cout), bf16=bf16)
            ctx.mark_non_differentiable(out16)
        else:
            gemm(B * Lo * N, Cout, CT, act, CT, wf2, CT, out, Cout, a_win=awin, bias=bf, bf16=bf16)
        ctx.save_for_backward(inp, y, act, stats, gamma, beta, wf, *packs)
        ctx.w357 = (w3.detach(), w5.detach(), w7.detach())       # raw (Cout, cin, k) weights: the fused d-inp kernel packs them
        ctx.inp16 = inp16 if (side16 and inp16 is not None and ld_in % 8 == 0) else None   # dW reads it instead of inp
        ctx.dims = (B, Lc, N, ld_in, cin, Cout, stride, Lo, need_dinp, bf16)
        ctx.compact = compact
        return out, out16

    @staticmethod
    def _dx_seq(bf16, dy16: bool, Lc: int, Cout: int, ld_in: int) -> bool:
        """Does the one-launch d-input kernel serve this block?  (dy16: the backward's dy is a bf16 tensor)"""
        return int(bf16) in (ops.PREC_FP32, ops.PREC_BF16) and (dy16 or int(bf16) == ops.PREC_FP32) \
            and ops.conv_dx_seq_ok(Lc, Cout, ld_in, f32=not dy16)

    @staticmethod
    def backward(ctx, dout, _dout16=None):
        inp, y, act, stats, gamma, beta, wf, bp3, bp5, bp7 = ctx.saved_tensors
        B, Lc, N, ld_in, cin, Cout, stride, Lo, need_dinp, bf16 = ctx.dims
        CT = 3 * Cout
        M = B * Lc * N
        Mo = B * Lo * N
        dout = dout.contiguous()
        wf2 = wf.view(Cout, CT)
        # final 1x1 strided conv
        # bf16 mode: both contractions read dout as a bf16 twin (their loaders would round it) -- written by the pass that sums
        # the fp32 values into the bias gradient
        if int(bf16) == ops.PREC_BF16 and act.dtype == torch.bfloat16 and ctx.compact and ops.tn_ok(Cout, CT, Mo) and Cout % 8 == 0:
            dout_g = torch.empty(Mo, Cout, device=dout.device, dtype=torch.bfloat16)
            dbf = colsum(dout, Cout, Mo, 1, 1, Cout, twin=dout_g)[0]
        else:
            dout_g = dout
            dbf = colsum(dout, Cout, Mo, 1, 1, Cout)[0]
        dwf = _empty(Cout, CT, like=inp)
        gemm(Cout, CT, Mo, dout_g, Cout, act, CT, dwf, CT, a_layout=A_KM, b_layout=B_KN,
             b_win=None if ctx.compact else win(N, Lc, Lo, stride, 1, CT, 0),
             split_k=pick_split_k(Cout, CT, Mo, prec=bf16), bf16=bf16)
        # the gradient at the 1x1 conv's input: a bf16 tensor in bf16 mode (what the backward of a bf16 Conv1d hands to the
        # fp32 GELU backward under autocast), read once by the GroupNorm + GELU backward
        d16 = act.dtype == torch.bfloat16 and ops.uses_bf16(CT, Cout, Cout, CT, b_layout=B_KN)
        dact = torch.empty(B, Lo, N, CT, device=inp.device, dtype=torch.bfloat16 if d16 else torch.float32)
        if ctx.wfT16 is not None and dout_g.dtype == torch.bfloat16 and d16:
            gemm(Mo, CT, Cout, dout_g, Cout, ctx.wfT16, Cout, dact, CT, bf16=bf16)       # [row][k] operands: the LDS-DMA kernel
        else:
            gemm(Mo, CT, Cout, dout_g, Cout, wf2, CT, dact, CT, b_layout=B_KN, bf16=bf16)
        # GroupNorm + GELU
        dy = torch.empty(B, Lc, N, CT, device=inp.device, dtype=act.dtype)
        dgamma, dbeta, dbconv = ops.groupnorm_gelu_bwd(dact, stride, y, gamma, beta, stats, dy, B, Lc, N, Cout)
        dinp = _empty(B, Lc, N, ld_in, like=inp) if need_dinp else None
        # d inp of the three kernel sizes in ONE launch that reads dy once (csrc/conv_seq.hip) instead of three
        # accumulating window GEMMs: exact fp32 from an fp32 dy, the bf16 mode's roundings from a bf16 dy (the bf16x3 /
        # bf16x6 modes keep the GEMM path)
        dx_seq = need_dinp and ConvBlockFn._dx_seq(bf16, dy.dtype == torch.bfloat16, Lc, Cout, ld_in)
        if dx_seq:
            ops.conv_dx(dy, ctx.w357[0], ctx.w357[1], ctx.w357[2], dinp, B, Lc, N, Cout, cin, ld_in)
        grads = []
        # d weights of the three kernel sizes in ONE persistent launch that reads inp16 and dy once (csrc/conv_dw_seq.hip)
        # instead of three split-K window GEMMs: exact fp32 from the fp32 tensors, the bf16 mode's arithmetic from the bf16
        # ones (the bf16x3 / bf16x6 modes keep the GEMM path)
        dw_in = ctx.inp16 if dy.dtype == torch.bfloat16 else (inp if int(bf16) == ops.PREC_FP32 else None)
        dw_seq = dw_in is not None and ops.conv_dw_seq_ok(Lc, Cout, ld_in)
        dws = ops.conv_dw(dw_in, dy, B, Lc, N, Cout, cin, ld_in) if dw_seq else None
        for j, (k, bp) in enumerate(((3, bp3), (5, bp5), (7, bp7))):
            K = k * ld_in
            db = dbconv[j * Cout:(j + 1) * Cout]
            if dw_seq:
                if need_dinp and not dx_seq:
                    gemm(M, ld_in, k * Cout, dy, CT, bp, ld_in, dinp, ld_in, b_layout=B_KN, a_off=j * Cout,
                         a_win=win(N, Lc, Lc, 1, k, Cout, (k - 1) // 2), accumulate=(j > 0), bf16=bf16)
                grads += [dws[j], db, dgamma[j * Cout:(j + 1) * Cout], dbeta[j * Cout:(j + 1) * Cout]]
                continue
            dpack = _empty(Cout, K, like=inp)
            b_src = ctx.inp16 if (ctx.inp16 is not None and dy.dtype == torch.bfloat16
                                  and os.environ.get("TECM_DW_B16", "1")[:1] != "0") else inp
            gemm(Cout, K, M, dy, CT, b_src, ld_in, dpack, K, a_layout=A_KM, b_layout=B_KN, a_off=j * Cout,
                 b_win=win(N, Lc, Lc, 1, k, ld_in, (k - 1) // 2), split_k=pick_split_k(Cout, K, M, prec=bf16), bf16=bf16)
            dw = ops.conv_weight_unpack(dpack, Cout, ld_in, k)
            if ld_in != cin:
                dw = dw[:, :cin, :].contiguous()
            if need_dinp and not dx_seq:
                gemm(M, ld_in, k * Cout, dy, CT, bp, ld_in, dinp, ld_in, b_layout=B_KN, a_off=j * Cout,
                     a_win=win(N, Lc, Lc, 1, k, Cout, (k - 1) // 2), accumulate=(j > 0), bf16=bf16)
            grads += [dw, db, dgamma[j * Cout:(j + 1) * Cout], dbeta[j * Cout:(j + 1) * Cout]]
        return (dinp, None, None, None, None, None, *grads, dwf.view_as(wf), dbf)


# ============================================================================ stage a-5 (+ wpe / embd dropout of a-6)
class PatchEmbedFn(torch.autograd.Function):
    """LatentPatchingProjection (modules.py:100-119) fused with GPT2Model's `+ wpe` and `drop`
    (modeling_gpt2.py:576-604).  conv (B, Lc, N, D) -> tokens (B, P, N, d_llm), P = Lc // patch_len."""

    @staticmethod
    def forward(ctx, conv, conv16, Wp, bp, wpe, patch_len: int, plan: DropPlan):
        """conv16: optional bf16 copy of conv (bf16 mode; the forward GEMM reads it instead)."""
        B, Lc, N, D = conv.shape
        P = Lc // patch_len
        d_llm = Wp.shape[0]
        M = B * P * N
        K = patch_len * D
        h0 = _empty(B, P, N, d_llm, like=conv)
        w = win(N, Lc, P, patch_len, patch_len, D, 0)
        dspec = plan.spec(SITE_EMBD, d_llm) if wpe is not None else None
        a_in = conv16 if (conv16 is not None and int(plan.bf16) == ops.PREC_BF16 and D % 8 == 0) else conv
        # bf16 operands + taps that are whole 64-wide K-tiles inside the sequence: the LDS-DMA kernel walks the window view
        # itself; it takes the weight as a bf16 tensor too (W and, for the backward, W^T rounded: one launch)
        ctx.WpT16 = None
        Wf = Wp
        if a_in.dtype == torch.bfloat16 and D % 64 == 0 and P * patch_len == Lc and M >= 256 and d_llm >= 128:
            Wf, ctx.WpT16 = ops.weight_bf16(Wp, same=True, transposed=True)
        gemm(M, d_llm, K, a_in, D, Wf, K, h0, d_llm, a_win=w, bias=bp,
             rowbias=(wpe, wpe.shape[1], N, P) if wpe is not None else None, out_drop=dspec, bf16=plan.bf16)
        # the weight gradient contracts the same tensor the forward read: the bf16 copy in bf16 mode (what its loader would
        # round the fp32 one to), at half the bytes and in the form the LDS-DMA kernel takes
        ctx.save_for_backward(a_in, Wp, wpe if wpe is not None else Wp)
        ctx.meta = (B, Lc, N, D, P, d_llm, patch_len, wpe is not None, dspec)
        ctx.plan = plan
        return h0

    @staticmethod
    def backward(ctx, dh0):
        conv, Wp, wpe = ctx.saved_tensors
        B, Lc, N, D, P, d_llm, patch_len, has_wpe, dspec = ctx.meta
        plan = ctx.plan
        M = B * P * N
        K = patch_len * D
        dh0 = dh0.contiguous()
        w = win(N, Lc, P, patch_len, patch_len, D, 0)
        # the embd-dropout mask of the forward epilogue, applied once: the masked gradient feeds two column sums and
        # two GEMMs
        # (bf16 mode: the two GEMMs read a bf16 twin -- what their loaders would round the fp32 values to -- so that the
        #  weight gradient runs on the natural-orientation LDS-DMA kernel; the column sums keep the fp32 values)
        t16 = int(plan.bf16) == ops.PREC_BF16 and conv.dtype == torch.bfloat16 and \
            ops.tn_ok(d_llm, K, M) and d_llm % 8 == 0
        dh16 = None
        cdrop = None                                       # mask the column sums apply to dh0 on the fly
        if t16:
            # ONE pass over dh0: the embd-dropout mask applied on the fly, the masked values summed in fp32 (wpe / bias
            # gradients) and written as the bf16 tensor the two GEMMs read; the fp32 masked gradient is never materialised
            dh16 = torch.empty(M, d_llm, device=dh0.device, dtype=torch.bfloat16)
            cdrop = dspec
        elif dspec is not None:
            dh0 = ops.dropout_apply(dh0, M, d_llm, dspec)
        dwpe = None
        if has_wpe:
            dwpe = torch.zeros_like(wpe)
            colsum(dh0, d_llm, B, N, P, d_llm, out=dwpe, in_drop=cdrop, twin=dh16)   # rows 0..P-1 of wpe
            dbp = colsum(dwpe, d_llm, P, 1, 1, d_llm)[0]                     # the bias gradient = the sum of those P rows:
        else:                                                                #   no second pass over the M x d_llm gradient
            dbp = colsum(dh0, d_llm, M, 1, 1, d_llm, in_drop=cdrop, twin=dh16)[0]
        dWp = _empty(d_llm, K, like=conv)
        dg = dh16 if dh16 is not None else dh0
        gemm(d_llm, K, M, dg, d_llm, conv, D, dWp, K, a_layout=A_KM, b_layout=B_KN, b_win=w,
             split_k=pick_split_k(d_llm, K, M, prec=plan.bf16), bf16=plan.bf16)
        dconv = _empty(B, Lc, N, D, like=conv)
        if P * patch_len != Lc:
            dconv.zero_()
        if dh16 is not None and K % 8 == 0:
            # W^T rounded to bf16 ([K][d_llm], one tiny launch): both operands [row][k] bf16 tensors -> the LDS-DMA kernel
            WpT16 = ctx.WpT16 if ctx.WpT16 is not None else ops.weight_bf16(Wp, same=False, transposed=True)[1]
            gemm(M, K, d_llm, dh16, d_llm, WpT16, d_llm, dconv, D, c_win=w, bf16=plan.bf16)
        else:
            gemm(M, K, d_llm, dg, d_llm, Wp, K, dconv, D, b_layout=B_KN, c_win=w, bf16=plan.bf16)
        return dconv, None, dWp, dbp, dwpe, None, None


# ============================================================================ stage a-6
_NK_CACHE: dict = {}
_KEXT_GEN: dict = {}           # data_ptr of a cached K-extended c_attn operand -> number of lora_fold rewrites (GPT2StackFn)


def _frozen_copy(W: torch.Tensor, kind: str) -> Optional[torch.Tensor]:
    """Derived copy of a FROZEN transformers-Conv1D weight W[K][N], cached per parameter and refreshed when the
    tensor is modified through torch (load_state_dict bumps _version):
      "nk32"  W^T as [N][K] fp32  -- the fp32 MFMA GEMM reads a [row][k] B tile with one ds_read_b128 per fragment,
              a [k][row] tile with four ds_read_b32: the forward GEMMs run ~5 % faster on the transposed copy (the
              backward already uses W as it is);
      "nk16"  W^T as [N][K] bf16, "kn16"  W as [K][N] bf16 -- bf16 mode: the weight operand of the forward / backward
              GEMMs is staged by a pure copy at half the bytes (TecmGemm::io_bf16), rounded once here instead of
              once per tile load.
    Trainable weights return None: their storage is updated by the fused optimizer behind torch's version counter.
    A write through `W.data` (which bumps no version counter) is NOT seen: call `invalidate_frozen_copies()` after
    editing frozen weights that way (load_state_dict, copy_ and every other torch op on W itself are picked up)."""
    if W.requires_grad:
        return None
    key = (id(W), kind)
    hit = _NK_CACHE.get(key)
    if hit is not None and hit[0]() is W and hit[1] == W._version and hit[2] == W.data_ptr():
        return hit[3]
    if len(_NK_CACHE) > 256:                                 # drop entries whose parameter is gone (ids get reused)
        for k in [k for k, v in _NK_CACHE.items() if v[0]() is None]:
            del _NK_CACHE[k]
    Wd = W.detach()
    if kind == "nk32":
        out = Wd.t().contiguous()
    elif kind == "nk16":
        out = Wd.t().contiguous().bfloat16()
    elif kind == "kn16":
        out = Wd.contiguous().bfloat16()
    elif kind in ("kext_kn32", "kext_kn16", "kext_nk32", "kext_nk16"):
        # K-extended c_attn operands with room for the LoRA slice (ops.lora_fold refreshes it every step):
        # "kext_kn*" = [ W ; 0 ] as [K + r][N] (backward), "kext_nk*" = [ W^T | 0 ] as [N][K + r] (forward)
        K, N = Wd.shape
        dt = torch.bfloat16 if kind.endswith("16") else torch.float32
        if "_kn" in kind:
            out = torch.zeros(K + LORA_R, N, device=W.device, dtype=dt)
            out[:K].copy_(Wd)
        else:
            out = torch.zeros(N, K + LORA_R, device=W.device, dtype=dt)
            out[:, :K].copy_(Wd.t())
    else:
        raise ValueError(kind)
    _NK_CACHE[key] = (weakref.ref(W), W._version, W.data_ptr(), out)
    return out


def invalidate_frozen_copies() -> None:
    """Drop every cached transposed / bf16 copy of the frozen GPT-2 weights (see _frozen_copy)."""
    _NK_CACHE.clear()


def _frozen_nk(W: torch.Tensor) -> Optional[torch.Tensor]:
    return _frozen_copy(W, "nk32")


def _fwd_weight(W: torch.Tensor, K: int, N: int, prec=0):
    """(tensor, ldb, b_layout) for  x[M,K] . W[K,N]  in the forward pass (bf16 mode: the cached bf16 copy)."""
    wt = _frozen_copy(W, "nk16" if int(prec) == ops.PREC_BF16 else "nk32")
    return (wt, K, B_NK) if wt is not None else (W, N, B_KN)


def _bwd_weight(W: torch.Tensor, prec=0) -> torch.Tensor:
    """B operand of  dY[M,N] . W[K,N]^T  (W read as [n = K][k = N]): bf16 mode uses the cached bf16 copy."""
    if int(prec) == ops.PREC_BF16:
        w16 = _frozen_copy(W, "kn16")
        if w16 is not None:
            return w16
    return W


class GPT2StackFn(torch.autograd.Function):
    """GPT2Block x n_layers + ln_f with LoRA(r=32) on c_attn (modeling_gpt2.py:262-310, :620;
    peft Linear: modules.py:177-186).  h0 (B, T, N, 768) already holds inputs_embeds + wpe (+ embd dropout).
    params per layer: ln1_w, ln1_b, Wqkv(768,2304), bqkv, loraA(32,768), loraB(2304,32), Wo, bo,
                      ln2_w, ln2_b, Wfc, bfc, Wproj, bproj      then lnf_w, lnf_b."""
    PER_LAYER = 14
    FROZEN_SLOTS = {2: "attn.c_attn.base_layer.weight", 3: "attn.c_attn.base_layer.bias", 6: "attn.c_proj.weight",
                    7: "attn.c_proj.bias", 10: "mlp.c_fc.weight", 11: "mlp.c_fc.bias", 12: "mlp.c_proj.weight",
                    13: "mlp.c_proj.bias"}

    @staticmethod
    def forward(ctx, h0, n_layers: int, plan: DropPlan, *params):
        # The backward below only forms the gradients the reference trains (freeze rule modules.py:195-203: lora_, ln_,
        # wpe).  A base weight switched to requires_grad=True would silently get no gradient: refuse instead.
        for i in range(n_layers):
            for slot, name in GPT2StackFn.FROZEN_SLOTS.items():
                if params[i * GPT2StackFn.PER_LAYER + slot].requires_grad:
                    raise _lib_error(f"h.{i}.{name} has requires_grad=True: the MI355X path computes gradients for the "
                                     "LoRA, LayerNorm and wpe parameters only (the reference's freeze rule, "
                                     "modules.py:195-203); freeze the GPT-2 base weights")
        B, T, N, D = h0.shape
        M = B * T * N
        KE = D + LORA_R
        h = h0.contiguous()
        saved: List[torch.Tensor] = []
        ctx_lAT: List[Optional[torch.Tensor]] = []           # per layer: bf16 lora_A^T (bf16 mode) for the backward's dz . A, or None
        ctx_gen: List[int] = []                              # per layer: generation of the shared K-extended operand (see wcat)
        for i in range(n_layers):
            (ln1w, ln1b, Wqkv, bqkv, lA, lB, Wo, bo, ln2w, ln2b, Wfc, bfc, Wpr,
             bpr) = params[i * GPT2StackFn.PER_LAYER:(i + 1) * GPT2StackFn.PER_LAYER]
            F3, F4 = Wqkv.shape[1], Wfc.shape[1]
            b16 = int(plan.bf16) == ops.PREC_BF16           # bf16 mode: weights and GEMM-only activations live in HBM as bf16
            wcatT = _frozen_copy(Wqkv, "kext_nk16" if b16 else "kext_nk32")   # [ W^T | (alpha/r) B ]: forward operand
            Wo_f, ldo_f, lay_o = _fwd_weight(Wo, D, D, plan.bf16)
            Wfc_f, ldfc_f, lay_fc = _fwd_weight(Wfc, D, F4, plan.bf16)
            Wpr_f, ldpr_f, lay_pr = _fwd_weight(Wpr, F4, D, plan.bf16)
            # Activations whose ONLY reader is a bf16 GEMM are written as bf16 by their producer (rounded once there
            # instead of in that GEMM's loader: bit-identical, half the bytes both ways): LN1's output for c_attn, the
            # attention context for attn.c_proj, LN2's output for c_fc, gelu(c_fc) for mlp.c_proj.
            a16 = b16 and all(w is not None and w.dtype == torch.bfloat16 for w in (wcatT, Wo_f, Wfc_f, Wpr_f))
            st1 = _empty(M, 2, like=h)
            lspec = plan.spec(site_lora(i), KE)
            if a16:
                # bf16 mode: [ LN1(h) | z ] only ever feeds bf16 contractions, so it exists as bf16 alone (u16); the LoRA
                # branch's input drop(LN1(h)) -- what autocast casts in front of lora_A -- is a second bf16 output of the
                # LayerNorm kernel (u16d), read by the LoRA-A GEMM here and by its weight gradient in the backward, and z
                # is written straight into u16's last 32 columns as bf16
                u = None
                u16 = torch.empty(M, KE, device=h.device, dtype=torch.bfloat16)
                u16d = torch.empty(M, D, device=h.device, dtype=torch.bfloat16) if lspec is not None else None
                ops.layernorm_fwd(h, D, ln1w, ln1b, None, KE, st1, M, D, y16=u16, ldy16=KE, y16d=u16d, ldy16d=D, drop16d=lspec)
                a_lora, ld_lora = (u16d, D) if u16d is not None else (u16, KE)
                # lora_A rounded (and transposed for its d-input contraction in the backward): both operands bf16 tensors
                lA16, lAT16 = ops.weight_bf16(lA, same=True, transposed=True)
                ctx_lAT.append(lAT16)
                gemm(M, LORA_R, D, a_lora, ld_lora, lA16, D, u16, KE, c_off=D, bf16=plan.bf16)
                u_s, ud_s = u16, (u16d if u16d is not None else h.new_empty(0))
            else:
                u = _empty(M, KE, like=h)                   # [ LN1(h) | z = drop(LN1(h)) A^T ]  (fp32: the LoRA gradients read it)
                u16 = None
                ops.layernorm_fwd(h, D, ln1w, ln1b, u, KE, st1, M, D)
                gemm(M, LORA_R, D, u, KE, lA, D, u, KE, c_off=D, a_drop=lspec, bf16=plan.bf16)
                u_s, ud_s = u, h.new_empty(0)
                ctx_lAT.append(None)
            # [ W ; (alpha/r) B^T ]  K-extended c_attn, backward operand ([KE][F3]) and forward operand ([F3][KE]): the
            # frozen 768 x 2304 part of both is cached per parameter version (_frozen_copy), ONE launch refreshes the 32
            # LoRA rows / columns of both from lora_B.  (The buffers are the cache's: they are rewritten by the next
            # forward of this layer, normally after the backward that reads `wcat` has run.  The rewrite goes through a raw
            # pointer, behind torch's version counter, so every fold bumps a generation number of the buffer and the backward
            # checks it: a second forward of the same frozen base weight with an outstanding backward -- two adapters over
            # one base, retain_graph across an optimizer step -- raises instead of differentiating against the wrong B.)
            wcat = _frozen_copy(Wqkv, "kext_kn16" if b16 else "kext_kn32")
            ops.lora_fold(lB.detach(), LORA_SCALE, wcat, wcatT, D)
            _KEXT_GEN[wcat.data_ptr()] = _KEXT_GEN.get(wcat.data_ptr(), 0) + 1
            ctx_gen.append(_KEXT_GEN[wcat.data_ptr()])
            # bf16 mode: qkv is written as bf16 by the c_attn GEMM (what a Linear's output is under autocast) and read as
            # such by the attention kernels, forward and backward: 644 -> 322 MB per layer, three times over
            qkv = torch.empty(M, F3, device=h.device, dtype=torch.bfloat16 if (a16 and QKV16) else torch.float32)
            gemm(M, F3, KE, u16 if a16 else u, KE, wcatT, KE, qkv, F3, b_layout=B_NK, bias=bqkv, bf16=plan.bf16)
            cx = torch.empty(M, D, device=h.device, dtype=torch.bfloat16 if a16 else torch.float32)
            aspec = plan.spec(site_attn(i), 1)
            ops.attention_fwd(qkv, cx, B, T, N, GPT_HEADS, D, aspec)
            h2 = _empty(M, D, like=h)
            gemm(M, D, D, cx, D, Wo_f, ldo_f, h2, D, b_layout=lay_o, bias=bo, out_drop=plan.spec(site_res1(i), D),
                 residual=(h, D), bf16=plan.bf16)
            st2 = _empty(M, 2, like=h)
            if a16:
                u2 = torch.empty(M, D, device=h.device, dtype=torch.bfloat16)
                ops.layernorm_fwd(h2, D, ln2w, ln2b, None, D, st2, M, D, y16=u2, ldy16=D)
            else:
                u2 = _empty(M, D, like=h)
                ops.layernorm_fwd(h2, D, ln2w, ln2b, u2, D, st2, M, D)
            # gelu(fc) is only ever read by the c_proj GEMM: in bf16 mode it is written as bf16.  The pre-activation
            # the backward differentiates GELU at is bf16 too (TECM_IO_PRE_BF16: rounded BEFORE the activation, as
            # autocast's bf16 Linear output is) -- a third less to write here, half as much to read back there.
            f16 = b16 and Wfc_f.dtype == torch.bfloat16 and Wpr_f.dtype == torch.bfloat16
            a = torch.empty(M, F4, device=h.device, dtype=torch.bfloat16 if f16 and PRE16 else torch.float32)
            f = torch.empty(M, F4, device=h.device, dtype=torch.bfloat16 if f16 else torch.float32)
            gemm(M, F4, D, u2, D, Wfc_f, ldfc_f, f, F4, b_layout=lay_fc, bias=bfc, preact=(a, F4), act=ACT_GELU_TANH,
                 bf16=plan.bf16)
            h3 = _empty(M, D, like=h)
            gemm(M, D, F4, f, F4, Wpr_f, ldpr_f, h3, D, b_layout=lay_pr, bias=bpr, out_drop=plan.spec(site_res2(i), D),
                 residual=(h2, D), bf16=plan.bf16)
            del cx, u2, f                                   # forward-only buffers: the backward needs none of them
            saved += [h, u_s, ud_s, st1, wcat, qkv, h2, st2, a]
            h = h3
        lnfw, lnfb = params[n_layers * GPT2StackFn.PER_LAYER:]
        stf = _empty(M, 2, like=h)
        # bf16 mode inside the whole model (plan.fuse_head): ln_f's only reader is F.dropout + the head's first Linear
        # (tec_mollm.py:115, modules.py:307), a bf16 contraction over view(B*N, T*D).  ln_f then writes exactly that operand:
        # dropout applied, rounded to bf16, rows in sequence-major order -- no fp32 ln_f output, no dropout pass, no window view
        # in the head GEMMs; the gradient comes back in the same form and ln_f's backward applies the mask (round 4).
        ctx.fused = bool(plan.fuse_head) and int(plan.bf16) == ops.PREC_BF16 and D % 8 == 0
        if ctx.fused:
            out = torch.empty(B, N, T * D, device=h.device, dtype=torch.bfloat16)
            pspec = plan.spec(SITE_POST, D)
            ops.layernorm_fwd(h, D, lnfw, lnfb, None, D, stf, M, D, y16d=out, ldy16d=D,
                              drop16d=pspec if pspec is not None else ops.NO_DROP, seq_major=(T, N))
        else:
            out = _empty(B, T, N, D, like=h)
            ops.layernorm_fwd(h, D, lnfw, lnfb, out, D, stf, M, D)
        ctx.save_for_backward(h, stf, *saved, *params)
        ctx.lAT16 = ctx_lAT
        ctx.kext_gen = ctx_gen
        ctx.meta = (B, T, N, D, n_layers, plan, len(saved))
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, N, D, n_layers, plan, nsaved = ctx.meta
        tens = ctx.saved_tensors
        h_last, stf = tens[0], tens[1]
        saved = tens[2:2 + nsaved]
        params = tens[2 + nsaved:]
        M = B * T * N
        KE = D + LORA_R
        dout = dout.contiguous()
        lnfw = params[n_layers * GPT2StackFn.PER_LAYER]
        dh = _empty(M, D, like=dout)
        # every LayerNorm backward also emits dropout(dx) for the GEMM that sits behind the next resid dropout
        b16 = int(plan.bf16) == ops.PREC_BF16

        def masked_buf(sp_, W):
            """dropout(dx) in front of the GEMM  . W^T : bf16 when that GEMM reads a bf16 copy of the frozen W."""
            if sp_ is None:
                return None
            w16 = b16 and _bwd_weight(W, plan.bf16).dtype == torch.bfloat16
            return torch.empty(M, D, device=dout.device, dtype=torch.bfloat16 if w16 else torch.float32)

        sp = plan.spec(site_res2(n_layers - 1), D)
        last_Wpr = params[(n_layers - 1) * GPT2StackFn.PER_LAYER + 12]
        dhm = masked_buf(sp, last_Wpr) if sp is not None else dh
        nig = ctx.needs_input_grad                        # (h0, n_layers, plan, *params): params start at index 3
        base_f = 3 + n_layers * GPT2StackFn.PER_LAYER
        dlnfw, dlnfb = ops.layernorm_bwd(dout, D, h_last, D, lnfw, stf, None, dh, M, D,
                                         dx_masked=dhm if sp is not None else None, mask_drop=sp,
                                         need_dgb=nig[base_f] or nig[base_f + 1],
                                         dy_seq_major=(T, N, plan.spec(SITE_POST, D)) if ctx.fused else None)
        pgrads: List[Optional[torch.Tensor]] = [None] * (n_layers * GPT2StackFn.PER_LAYER)
        for i in reversed(range(n_layers)):
            (ln1w, ln1b, Wqkv, bqkv, lA, lB, Wo, bo, ln2w, ln2b, Wfc, bfc, Wpr,
             bpr) = params[i * GPT2StackFn.PER_LAYER:(i + 1) * GPT2StackFn.PER_LAYER]
            h, u, ud, st1, wcat, qkv, h2, st2, a = saved[i * 9:(i + 1) * 9]
            if _KEXT_GEN.get(wcat.data_ptr()) != ctx.kext_gen[i]:
                raise _lib_error(f"GPT-2 block {i}: the K-extended c_attn operand [W ; 2 B^T] was rewritten by a later forward "
                                "of the same frozen base weight before this backward ran (shared cache buffer); run each "
                                "forward's backward before the next forward of that layer")
            F3, F4 = Wqkv.shape[1], Wfc.shape[1]
            # MLP:  h3 = h2 + drop(gelu(u2 Wfc + b) Wpr + b)
            Wpr_b, Wfc_b = _bwd_weight(Wpr, plan.bf16), _bwd_weight(Wfc, plan.bf16)
            # d gelu-input is only read by the next GEMM: bf16 in bf16 mode (see the forward's `f`)
            da16 = Wpr_b.dtype == torch.bfloat16 and Wfc_b.dtype == torch.bfloat16
            da = torch.empty(M, F4, device=dh.device, dtype=torch.bfloat16 if da16 else torch.float32)
            gemm(M, F4, D, dhm, D, Wpr_b, D, da, F4, act=ACT_GELU_TANH, dact_src=(a, F4), bf16=plan.bf16)
            # the gradients the frozen bf16 Linears hand back for their inputs (d LN2-out here, d ctx and d [LN1-out | z]
            # below) are bf16 tensors under autocast: written as such by their GEMMs, read once by fp32 kernels
            g16 = b16 and da16 and GRAD16
            du2 = torch.empty(M, D, device=dh.device, dtype=torch.bfloat16 if g16 else torch.float32)
            gemm(M, D, F4, da, F4, Wfc_b, F4, du2, D, bf16=plan.bf16)
            del da
            dh2 = _empty(M, D, like=dh)
            sp = plan.spec(site_res1(i), D)
            dh2m = masked_buf(sp, Wo) if sp is not None else dh2
            pb = 3 + i * GPT2StackFn.PER_LAYER
            dg2, db2 = ops.layernorm_bwd(du2, D, h2, D, ln2w, st2, dh, dh2, M, D,
                                         dx_masked=dh2m if sp is not None else None, mask_drop=sp,
                                         need_dgb=nig[pb + 8] or nig[pb + 9])
            # attention: h2 = h + drop(ctx Wo + b)
            Wo_b = _bwd_weight(Wo, plan.bf16)
            c16 = g16 and Wo_b.dtype == torch.bfloat16 and dh2m.dtype == torch.bfloat16 and qkv.dtype == torch.bfloat16
            dcx = torch.empty(M, D, device=dh.device, dtype=torch.bfloat16) if c16 else \
                (du2 if du2.dtype == torch.float32 else _empty(M, D, like=dh))            # fp32: reuse the buffer
            gemm(M, D, D, dh2m, D, Wo_b, D, dcx, D, bf16=plan.bf16)
            # dqkv is read by the c_attn dX GEMM ([row][k] A) and the LoRA-B dW GEMM ([k][m] A): bf16 when both run on the
            # bf16 matrix cores against a bf16 copy of [W ; (alpha/r) B^T]
            q16 = b16 and wcat.dtype == torch.bfloat16
            dqkv = torch.empty(M, F3, device=dh.device, dtype=torch.bfloat16 if q16 else torch.float32)
            ops.attention_bwd(qkv, dcx, dqkv, B, T, N, GPT_HEADS, D, plan.spec(site_attn(i), 1))
            u16g = g16 and q16 and u.dtype == torch.bfloat16
            du = torch.empty(M, KE, device=dh.device, dtype=torch.bfloat16 if u16g else torch.float32)   # [ d LN1-out | dz ]
            gemm(M, KE, F3, dqkv, F3, wcat, F3, du, KE, bf16=plan.bf16)
            lspec = plan.spec(site_lora(i), KE)
            dlB = _empty(F3, LORA_R, like=dh)
            gemm(F3, LORA_R, M, dqkv, F3, u, KE, dlB, LORA_R, a_layout=A_KM, b_layout=B_KN, b_off=D,
                 alpha=LORA_SCALE, split_k=pick_split_k(F3, LORA_R, M, prec=plan.bf16), bf16=plan.bf16)
            dlA = _empty(LORA_R, D, like=dh)
            if u.dtype == torch.bfloat16:                     # bf16 mode: drop(LN1(h)) was stored as bf16 by the forward
                b_in, ldb_in = (ud, D) if ud.numel() else (u, KE)
                gemm(LORA_R, D, M, du, KE, b_in, ldb_in, dlA, D, a_layout=A_KM, b_layout=B_KN, a_off=D,
                     split_k=pick_split_k(LORA_R, D, M, prec=plan.bf16), bf16=plan.bf16)
            else:
                gemm(LORA_R, D, M, du, KE, u, KE, dlA, D, a_layout=A_KM, b_layout=B_KN, a_off=D, b_drop=lspec,
                     split_k=pick_split_k(LORA_R, D, M, prec=plan.bf16), bf16=plan.bf16)
            # LoRA path back to LN1's output: the branch's input gradient dz A (lora_A's d-input GEMM, K = 32: a bf16 tensor in
            # bf16 mode, as under autocast) is a SECOND stream the LayerNorm backward adds through lora_dropout's mask -- it
            # used to be accumulated into du by that GEMM, a read-modify-write of the whole M x 768 gradient
            dzA = torch.empty(M, D, device=dh.device, dtype=torch.bfloat16 if (b16 and GRAD16) else torch.float32)
            if ctx.lAT16[i] is not None and dzA.dtype == torch.bfloat16 and du.dtype == torch.bfloat16:
                gemm(M, D, LORA_R, du, KE, ctx.lAT16[i], LORA_R, dzA, D, a_off=D, bf16=plan.bf16)     # [row][k] operands: LDS-DMA
            else:
                gemm(M, D, LORA_R, du, KE, lA, D, dzA, D, b_layout=B_KN, a_off=D, bf16=plan.bf16)
            dhn = _empty(M, D, like=dh)
            sp = plan.spec(site_res2(i - 1), D) if i > 0 else None
            dhm = masked_buf(sp, params[(i - 1) * GPT2StackFn.PER_LAYER + 12]) if sp is not None else dhn
            dg1, db1 = ops.layernorm_bwd(du, KE, h, D, ln1w, st1, dh2, dhn, M, D,
                                         dx_masked=dhm if sp is not None else None, mask_drop=sp,
                                         need_dgb=nig[pb + 0] or nig[pb + 1],
                                         add=(dzA, D, lspec))
            dh = dhn
            base = i * GPT2StackFn.PER_LAYER
            pgrads[base + 0], pgrads[base + 1] = dg1, db1
            pgrads[base + 4], pgrads[base + 5] = dlA, dlB
            pgrads[base + 8], pgrads[base + 9] = dg2, db2
        pg = [g if ctx.needs_input_grad[3 + j] else None for j, g in enumerate(pgrads)]
        return (dh.view(B, T, N, D), None, None, *pg, dlnfw, dlnfb)


# ============================================================================ stage a-7/a-8
class HeadFn(torch.autograd.Function):
    """F.dropout (tec_mollm.py:115) + PredictionHead (modules.py:295-313):
    hid (B, T, N, 768) -> (B, N, L_out); flattened feature index = t*768 + d."""

    @staticmethod
    def forward(ctx, hid, W1, b1, W2, b2, plan: DropPlan):
        ctx.fused = hid.dim() == 3 and hid.dtype == torch.bfloat16
        if ctx.fused:
            return HeadFn._forward_fused(ctx, hid, W1, b1, W2, b2, plan)
        B, T, N, D = hid.shape
        S = B * N
        Hd, K1 = W1.shape
        Lo = W2.shape[0]
        w = win(N, T, 1, T, T, D, 0)
        pspec = plan.spec(SITE_POST, D)
        hspec = plan.spec(SITE_HEAD, Hd)
        pre = _empty(S, Hd, like=hid)
        h1 = _empty(S, Hd, like=hid)
        hid = hid.contiguous()
        # F.dropout(hid) is read by the forward GEMM (once per 128-column tile) and again by the W1 gradient: mask it
        # once (one streaming pass) instead of hashing every element in both GEMM loaders
        # (bf16 mode: as the bf16 tensor autocast casts the dropped value to -- both readers are bf16 contractions)
        h16 = int(plan.bf16) == ops.PREC_BF16 and D % 8 == 0 and ops.uses_bf16(Hd, K1, D, K1) and \
            ops.uses_bf16(K1, S, Hd, D, a_layout=A_KM, b_layout=B_KN)
        hd = ops.dropout_apply(hid, B * T * N, D, pspec, out_bf16=h16) if pspec is not None else hid
        gemm(S, Hd, K1, hd, D, W1, K1, h1, Hd, a_win=w, bias=b1, preact=(pre, Hd), act=ACT_GELU_ERF,
             out_drop=hspec, bf16=plan.bf16)
        pred = _empty(B, N, Lo, like=hid)
        gemm(S, Lo, Hd, h1, Hd, W2, Hd, pred, Lo, bias=b2, bf16=plan.bf16)
        ctx.save_for_backward(hd, W1, W2, pre, h1)
        ctx.meta = (B, T, N, D, Hd, K1, Lo, w, pspec, hspec)
        ctx.plan = plan
        return pred

    @staticmethod
    def _forward_fused(ctx, hid, W1, b1, W2, b2, plan: DropPlan):
        """hid = bf16(F.dropout(ln_f(h))) as the plain (B, N, T*D) matrix GPT2StackFn wrote (plan.fuse_head): every head
        contraction is a plain one over bf16 tensors -- W1 and W1^T rounded once per step (tecm_weight_bf16)."""
        B, N, K1 = hid.shape
        S = B * N
        Hd = W1.shape[0]
        Lo = W2.shape[0]
        hspec = plan.spec(SITE_HEAD, Hd)
        W1_16, W1T16 = ops.weight_bf16(W1, same=True, transposed=True)
        pre = _empty(S, Hd, like=hid)
        h1 = _empty(S, Hd, like=hid)
        gemm(S, Hd, K1, hid, K1, W1_16, K1, h1, Hd, bias=b1, preact=(pre, Hd), act=ACT_GELU_ERF, out_drop=hspec, bf16=plan.bf16)
        pred = _empty(B, N, Lo, like=hid)
        gemm(S, Lo, Hd, h1, Hd, W2, Hd, pred, Lo, bias=b2, bf16=plan.bf16)
        ctx.save_for_backward(hid, W1, W2, pre, h1)
        ctx.W1T16 = W1T16
        ctx.meta = (B, N, K1, Hd, Lo, hspec)
        ctx.plan = plan
        return pred

    @staticmethod
    def _backward_fused(ctx, dpred):
        hid, W1, W2, pre, h1 = ctx.saved_tensors
        B, N, K1, Hd, Lo, hspec = ctx.meta
        plan = ctx.plan
        S = B * N
        dpred = dpred.contiguous()
        db2 = colsum(dpred, Lo, S, 1, 1, Lo)[0]
        dW2 = _empty(Lo, Hd, like=hid)
        gemm(Lo, Hd, S, dpred, Lo, h1, Hd, dW2, Hd, a_layout=A_KM, b_layout=B_KN, split_k=pick_split_k(Lo, Hd, S, prec=plan.bf16), bf16=plan.bf16)
        dpre = _empty(S, Hd, like=hid)
        gemm(S, Hd, Lo, dpred, Lo, W2, Hd, dpre, Hd, b_layout=B_KN, act=ACT_GELU_ERF, dact_src=(pre, Hd),
             out_drop=hspec, bf16=plan.bf16)
        dp = torch.empty(S, Hd, device=hid.device, dtype=torch.bfloat16)     # what both contractions round dpre to, written by
        db1 = colsum(dpre, Hd, S, 1, 1, Hd, twin=dp)[0]                      # the pass that sums the fp32 values into db1
        dW1 = _empty(Hd, K1, like=hid)
        gemm(Hd, K1, S, dp, Hd, hid, K1, dW1, K1, a_layout=A_KM, b_layout=B_KN,
             split_k=pick_split_k(Hd, K1, S, prec=plan.bf16), bf16=plan.bf16)
        # the gradient a bf16 Linear returns for its input is a bf16 tensor under autocast (train.py:68): stored as such, in
        # the operand's own (sequence-major) layout; ln_f's backward applies the dropout mask and the layout
        dhid = torch.empty(B, N, K1, device=hid.device, dtype=torch.bfloat16)
        gemm(S, K1, Hd, dp, Hd, ctx.W1T16, Hd, dhid, K1, bf16=plan.bf16)
        return dhid, dW1, db1, dW2, db2, None

    @staticmethod
    def backward(ctx, dpred):
        if ctx.fused:
            return HeadFn._backward_fused(ctx, dpred)
        hid, W1, W2, pre, h1 = ctx.saved_tensors
        B, T, N, D, Hd, K1, Lo, w, pspec, hspec = ctx.meta
        plan = ctx.plan
        S = B * N
        dpred = dpred.contiguous()
        db2 = colsum(dpred, Lo, S, 1, 1, Lo)[0]
        dW2 = _empty(Lo, Hd, like=hid)
        gemm(Lo, Hd, S, dpred, Lo, h1, Hd, dW2, Hd, a_layout=A_KM, b_layout=B_KN, split_k=pick_split_k(Lo, Hd, S, prec=plan.bf16), bf16=plan.bf16)
        dpre = _empty(S, Hd, like=hid)
        gemm(S, Hd, Lo, dpred, Lo, W2, Hd, dpre, Hd, b_layout=B_KN, act=ACT_GELU_ERF, dact_src=(pre, Hd),
             out_drop=hspec, bf16=plan.bf16)
        dW1 = _empty(Hd, K1, like=hid)
        # bf16 mode: both contractions read dpre as a bf16 twin (what their loaders round it to), written by the db1 pass
        if int(plan.bf16) == ops.PREC_BF16 and hid.dtype == torch.bfloat16 and ops.tn_ok(Hd, K1, S) and Hd % 8 == 0:
            dp = torch.empty(S, Hd, device=hid.device, dtype=torch.bfloat16)
            db1 = colsum(dpre, Hd, S, 1, 1, Hd, twin=dp)[0]
        else:
            dp = dpre
            db1 = colsum(dpre, Hd, S, 1, 1, Hd)[0]
        gemm(Hd, K1, S, dp, Hd, hid, D, dW1, K1, a_layout=A_KM, b_layout=B_KN, b_win=w,       # hid = dropout(hid) here
             split_k=pick_split_k(Hd, K1, S, prec=plan.bf16), bf16=plan.bf16)
        dhid = _empty(B, T, N, D, like=hid)
        if dp.dtype == torch.bfloat16 and K1 % 8 == 0:
            W1T16 = ops.weight_bf16(W1, same=False, transposed=True)[1]      # [K1][Hd] bf16: [row][k] operands for the LDS-DMA kernel
            gemm(S, K1, Hd, dp, Hd, W1T16, Hd, dhid, D, c_win=w, out_drop=pspec, bf16=plan.bf16)
        else:
            gemm(S, K1, Hd, dp, Hd, W1, K1, dhid, D, b_layout=B_KN, c_win=w, out_drop=pspec, bf16=plan.bf16)
        return dhid, dW1, db1, dW2, db2, None


# ============================================================================ loss
class HuberFn(torch.autograd.Function):
    """nn.HuberLoss(delta=1.0), mean reduction (train.py:372), loss and gradient in one pass."""

    @staticmethod
    def forward(ctx, pred, target, delta: float):
        if pred.dim() == 4 and pred.shape[3] == 1 and pred.shape == target.shape:
            # the model's permuted output view and the target, each in its own layout: no contiguous copies, and the
            # gradient comes back in the prediction's storage order (what the head's backward reads)
            loss, dpred = ops.huber_fwd_bwd_strided(pred, target, delta, 1.0, want_grad=True)
        else:
            loss, dpred = ops.huber_fwd_bwd(pred.contiguous(), target.contiguous(), delta, 1.0, want_grad=True)
            dpred = dpred.view(pred.shape)
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, gloss):
        (dpred,) = ctx.saved_tensors
        return dpred * gloss, None, None
