#!/usr/bin/env python3
"""bf16 mode, stage by stage: where does the HIP forward leave the bf16-emulating oracle (oracle/ref_cpu.BF16)?
For every stage the device output is compared (a) cumulatively and (b) with the ORACLE's input fed to the device
stage, which isolates that stage's own contribution.  Diagnostics only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from oracle import ref_cpu as R
from tests.parity import build_model
from tecmollm import functions as F_, graph as graph_

dev = torch.device("cuda")
grid = (int(os.environ.get("GH", 4)), int(os.environ.get("GW", 5)))
N = grid[0] * grid[1]
B, L = 2, 48
cfg = R.default_config(L_in=L, L_out=12, num_nodes=N)
p = R.init_params(cfg, seed=11)
x, tf, y = R.synthetic_batch(B, L, N, 6, 12, seed=111)
ei, _ = R.grid_graph(grid[0], grid[1], threshold_km=170.0)
q = R.BF16 if os.environ.get("Q", "bf16") == "bf16" else R.FP32
prec = "bf16" if q is R.BF16 else "fp32"


def err(a, b, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = (a - b).abs()
    rms = b.pow(2).mean().sqrt()
    print(f"{what:34s} max-norm {float(d.max() / b.abs().max()):.2e}  rms-rel {float(d.pow(2).mean().sqrt() / rms):.2e}  "
          f"worst elem / rms {float(d.max() / rms):.2e}")


def seq_to_tm(t, Bn, Nn):          # (S = B*N, T, D) -> (B, T, N, D)
    S, T, D = t.shape
    return t.view(Bn, Nn, T, D).permute(0, 2, 1, 3).contiguous()


with torch.no_grad():
    # ---------------- oracle intermediates
    h = R.embed(x, tf, p)
    xs = R.spatial(h, ei, p, 2, None)                                  # (L*B, N, C)
    xt = xs.view(L, B, N, 22).permute(1, 2, 0, 3).reshape(B * N, L, 22)
    c0 = R.conv_block(xt.permute(0, 2, 1), p, 0, 2, q)                 # (S, 64, 24)
    c1 = R.conv_block(c0, p, 1, 2, q)                                  # (S, 128, 12)
    tok = R.temporal_encoder(xt, p, cfg["temporal_strides"], cfg["patch_len"], q)
    hids = [R.gpt2_lora(tok, p, n, q) for n in (1, 2, 3)]              # each ends with ln_f
    pred = R.head(hids[-1], p, q)

    model = build_model(cfg, p, dev, "per_timestep", precision=prec).eval()
    plan = F_.DropPlan(False, 0.0, 0, 1 if prec == "bf16" else 0)
    tfd = tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, L, N, 4)
    meta = graph_.get(ei.to(dev), N, dev, 16)
    xs_d = F_.SpatialFn.apply(x.to(dev), tfd, *model.spatio_temporal_embedding.tables(),
                              *model.spatial_encoder.params(), meta, 2, B * L, plan)
    xs_o = xs.view(L, B, N, 22).permute(1, 0, 2, 3)                    # (B, L, N, 22)
    err(xs_d[..., :22], xs_o, "spatial (fp32 in both)")
    xs_o24 = torch.zeros(B, L, N, 24); xs_o24[..., :22] = xs_o
    emb = model.temporal_encoder.conv_embedder.embedder
    for name, inp_d in (("cumulative", xs_d), ("oracle input", xs_o24.to(dev))):
        b0, b0_16 = emb[0].forward_tm(inp_d, 22, True, plan.bf16, None)
        err(b0, seq_to_tm(c0.permute(0, 2, 1), B, N), f"conv block 0 [{name}]")
        b1, b1_16 = emb[1].forward_tm(b0, 64, True, plan.bf16, b0_16)
        err(b1, seq_to_tm(c1.permute(0, 2, 1), B, N), f"conv block 1 [{name}]")
    c0_tm = seq_to_tm(c0.permute(0, 2, 1), B, N).to(dev)
    b1, b1_16 = emb[1].forward_tm(c0_tm, 64, True, plan.bf16, c0_tm.bfloat16() if prec == "bf16" else None)
    err(b1, seq_to_tm(c1.permute(0, 2, 1), B, N), "conv block 1 [oracle block-0 out]")
    wpe = model.llm_backbone.trunk.wpe.weight
    h0 = model.temporal_encoder.forward_tm(xs_d, 22, wpe, plan)
    tok_o = seq_to_tm(tok + p[R.P_GPT + "wpe.weight"][:tok.shape[1]], B, N)
    err(h0, tok_o, "tokens + wpe [cumulative]")
    c1_tm = seq_to_tm(c1.permute(0, 2, 1), B, N).to(dev)
    h0b = model.temporal_encoder.patcher.forward_tm(c1_tm, wpe, plan, c1_tm.bfloat16() if prec == "bf16" else None)
    err(h0b, tok_o, "tokens + wpe [oracle conv out]")
    for n in (1, 2, 3):
        ps = model.llm_backbone.stack_params()
        sub = ps[:14 * n] + ps[-2:]
        hd = F_.GPT2StackFn.apply(h0, n, plan, *sub)
        err(hd, seq_to_tm(hids[n - 1], B, N), f"gpt2 x{n} + ln_f [cumulative]")
        hd = F_.GPT2StackFn.apply(tok_o.to(dev), n, plan, *sub)
        err(hd, seq_to_tm(hids[n - 1], B, N), f"gpt2 x{n} + ln_f [oracle tokens]")
    hid = model.llm_backbone.forward_tm(h0, plan)
    pr = model.prediction_head.forward_tm(hid, plan)
    err(pr.reshape(B * N, -1), pred, "head [cumulative]")
    pr = model.prediction_head.forward_tm(seq_to_tm(hids[-1], B, N).to(dev), plan)
    err(pr.reshape(B * N, -1), pred, "head [oracle hidden]")
