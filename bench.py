#!/usr/bin/env python3
"""Throughput of one TEC-MoLLM training step on N MI355X (BASELINE.json metric: train samples/sec on
synthetic (B,48,2911,10) -> (B,12,2911,1) batches).

    python bench.py [--gpus N --steps K --warmup W]          # N > 1: starts N ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                # or under torchrun: RANK / WORLD_SIZE come from the env

A step = forward + HuberLoss + backward + (RCCL all-reduce of the flat 12.3 MB gradient, the per-rank parameter
checksums riding in its tail) + clip(1.0) + AdamW + cosine-warm-restart scheduler, training mode (every dropout
site active), inputs resident in HBM.  Weak scaling: every rank processes `--batch` (default 8) samples;
value = all samples / max-over-ranks time.  Rank 0 prints ONE JSON line (< 4 KB: the driver parses the last stdout line)
carrying `roofline` (dominant kernel, measured with events on the launch stream) and, at N=1, `cpu_baseline` (the CPU
oracle's train step timed on the host cores of this box) and `configs_extra` (BASELINE configs[2] = bf16 and the
configs[4] per-GPU shape L_in=96 / L_out=24, each with its roofline summary).  N > 1: `config.dist` carries the
event-timed all-reduce (`allreduce_ms`).  Per-call-site GEMM tables, the non-GEMM block and per-rank lists go to
`bench_detail.json` (--detail-json) next to this script.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tec-mollm_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402   (importing torch does not touch the GPU; the launcher below relies on that)
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0


LINE_BUDGET = 4096          # bytes of the LAST stdout line; the driver keeps ~8 KB of stdout tail and parses that line


def _compact_roofline(r):
    """The scalar roofline summary the headline line carries; per-shape and non-GEMM tables go to the side file."""
    if not r:
        return r, None
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches", "avg_launch_ms", "share_of_step",
            "all_gemm_share_of_step")
    out = {k: r[k] for k in keep if k in r}
    t = r.get("traffic")
    out["traffic"] = t["hbm_bytes_per_launch"] if isinstance(t, dict) else t
    if isinstance(t, dict):
        out["traffic_source"] = t.get("source")
    st = r.get("step")
    if st:
        out["step"] = {k: st[k] for k in ("algorithmic_tflops", "peak", "frac_of_peak") if k in st}
    detail = {k: r[k] for k in ("shapes", "non_gemm", "timing", "traffic", "step") if r.get(k) is not None}
    return out, detail


def compact_line(line: dict):
    """(headline, detail): `headline` is the ONE JSON object printed as the last stdout line -- headline scalars, a scalar
    roofline summary, cpu_baseline, parity and each configs_extra leg as scalars + its roofline.{kernel, frac, step} --
    and always serialises to fewer than LINE_BUDGET bytes; `detail` carries everything that was cut (per-call-site GEMM
    tables, the non-GEMM block, per-rank lists, long sample descriptions) and is written next to the script."""
    head = dict(line)
    detail = {"headline_of": {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")}}
    head["roofline"], d = _compact_roofline(line.get("roofline"))
    if d:
        detail["roofline"] = d
    cb = line.get("cpu_baseline")
    if cb:
        head["cpu_baseline"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cb.items()
                                if k in ("value", "unit", "cores", "kind", "median_step_s", "sample")}
        head["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        detail["cpu_baseline"] = cb
    par = line.get("parity")
    if par:
        head["parity"] = {k: float(f"{par[k]:.4g}") for k in ("rmse_vs_ref", "max_rel_err", "r2_vs_ref", "ref_rms") if k in par}
        detail["parity"] = par
    cfg = dict(line.get("config") or {})
    di = cfg.get("dist")
    if di:
        detail["dist"] = di
        d2 = {k: v for k, v in di.items() if not isinstance(v, dict)}
        ar = di.get("allreduce_ms")
        if ar:
            d2["allreduce_ms"] = {k: ar[k] for k in ("mean_over_ranks", "max_over_ranks", "bytes", "launches") if k in ar}
        sr = di.get("step_ms_per_rank")
        if sr:
            d2["step_ms_per_rank"] = {k: sr[k] for k in ("min_over_ranks", "max_over_ranks", "slowest_single_step") if k in sr}
        cfg["dist"] = d2
    head["config"] = cfg
    for group in ("configs_extra", "other_precisions"):
        legs = line.get(group)
        if not legs:
            continue
        head[group], detail[group] = {}, {}
        for name, leg in legs.items():
            h = {k: v for k, v in leg.items() if k not in ("roofline", "workload")}
            if "workload" in leg:
                h["workload"] = str(leg["workload"])[:96]
            r, d = _compact_roofline(leg.get("roofline"))
            if r:
                h["roofline"] = {k: r[k] for k in ("kernel", "achieved", "peak", "frac", "avg_launch_ms", "share_of_step", "step") if k in r}
            head[group][name] = h
            detail[group][name] = dict(leg)
    text = json.dumps(head)
    if len(text) >= LINE_BUDGET:                           # last resort, never reached with today's fields: drop legs
        for group in ("other_precisions", "configs_extra"):
            if group in head and len(json.dumps(head)) >= LINE_BUDGET:
                head[group] = {k: {kk: vv for kk, vv in v.items() if kk in ("samples_per_s", "ms_per_step", "dtype")}
                               for k, v in head[group].items()}
    return head, detail


def emit(line: dict, detail_path=None):
    """Write the detail side file, then print the compact headline as the last stdout line."""
    head, detail = compact_line(line)
    path = detail_path or os.environ.get("TECM_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
    try:
        with open(path, "w") as f:
            json.dump(detail, f, indent=1)
        head["detail"] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:                                   # a read-only checkout must not cost the headline
        head["detail"] = f"not written: {e.__class__.__name__}"
    text = json.dumps(head)
    assert len(text) < LINE_BUDGET, f"bench line is {len(text)} bytes; the driver parses at most ~{LINE_BUDGET}"
    print(text, flush=True)
    return head


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (BASELINE configs[1]: B=8)")
    ap.add_argument("--L_in", type=int, default=48)
    ap.add_argument("--L_out", type=int, default=12)
    ap.add_argument("--c_in", type=int, default=10, help="raw feature width F (BASELINE: 10 -> d_emb 12)")
    ap.add_argument("--llm_layers", type=int, default=3, help="GPT-2 blocks (train.py:196; the reference's 4-GPU script uses 6)")
    ap.add_argument("--gat", choices=["per_timestep", "reference"], default="per_timestep")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16x3", "bf16x6"], default="fp32",
                    help="fp32 = BASELINE configs[1] (exact-f32 MFMA); bf16 = configs[2] (bf16 MFMA, fp32 accumulate)")
    ap.add_argument("--eval-mode", action="store_true", help="dropout off (diagnostics only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2, help="CPU baseline batch (BASELINE configs[0]: B=2)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU train steps after one warm-up step")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-precisions", action="store_true",
                    help="skip configs_extra (the bf16 and L_in=96 legs that follow the main run at N=1 with default flags)")
    ap.add_argument("--emulation-modes", action="store_true",
                    help="also time the opt-in bf16x6 / bf16x3 modes (context only; not part of the default run)")
    ap.add_argument("--loop", choices=["native", "reference"], default="native",
                    help="native: tecmollm.train.TrainStep (flat buffers, fused clip + AdamW).  reference: the statements of "
                         "the reference's own loop body (train.py:57-112) around the drop-in model -- autocast(bf16), "
                         "gradient_checkpointing_enable() per step, GradScaler, torch AdamW, clip_grad_norm_, loss.item()")
    ap.add_argument("--graph", action="store_true",
                    help="record the micro-batch once (hipGraph) and replay it: TrainStep.step_graphed; needs --warmup >= 2")
    ap.add_argument("--detail-json", default=None,
                    help="where the per-shape / non-GEMM tables go (default: bench_detail.json next to this script); "
                         "the last stdout line stays under 4 KB")
    ap.add_argument("--data", choices=["fixed", "window"], default="fixed",
                    help="fixed: one resident batch (default).  window: every step's batch is drawn by the device "
                         "window sampler (tecm_window_batch) from a resident synthetic series, as train.py:57-65 "
                         "draws it from the DataLoader")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ N-rank launcher
def visible_gpus():
    """Number of GPUs this process tree may use, WITHOUT touching the HIP runtime (the launcher parent must stay a
    process that never initialised the GPU): GPU nodes of the KFD topology in sysfs (CPU nodes have simd_count 0),
    narrowed by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  None when the topology
    is not readable (then nothing is checked here and a rank that finds no device fails on its own)."""
    root = os.environ.get("TECM_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted(os.listdir(root))
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes (one per GPU) of this same script and
    relay rank 0's JSON line.  Runs BEFORE anything in this process touches the GPU -- a process that has initialised
    HIP must never fork/exec GPU children -- and never replaces itself: the children are ordinary subprocesses, this
    process waits for them and exits with the first non-zero code (killing the exact PIDs it started)."""
    n = args.gpus
    backend = os.environ.get("TECM_DIST_BACKEND", "nccl")
    visible = visible_gpus()                       # sysfs only: this process never loads the HIP runtime
    if backend == "nccl" and visible is not None and visible < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs for RCCL (one rank per device), found {visible}",
              file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TECM_LAUNCHER="bench.py",
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
        out = None if r == 0 else subprocess.DEVNULL               # rank 0 prints the one JSON line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env, stdout=out))
    deadline = time.time() + float(os.environ.get("TECM_BENCH_TIMEOUT", 1500))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = code
        if time.time() > deadline:
            print("bench.py: ranks did not finish in time", file=sys.stderr)
            rc = 124
    for p in live:                                                 # a rank failed or timed out: stop the exact PIDs
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def make_config(args):
    conv_len = args.L_in // 4
    patch_len = 4
    if conv_len % patch_len != 0:
        patch_len = 2 if conv_len % 2 == 0 else 1
    return {
        "num_nodes": 2911, "d_emb": 22 - args.c_in, "spatial_in_channels_base": args.c_in,
        "spatial_out_channels": 11, "spatial_heads": 2, "temporal_channel_list": [64, 128],
        "temporal_strides": [2, 2], "patch_len": patch_len, "d_llm": 768, "llm_layers": getattr(args, "llm_layers", 3),
        "prediction_horizon": args.L_out, "temporal_seq_len": args.L_in, "num_years": 13,
    }


PRECISION_TEXT = {
    "fp32": "fp32",
    "bf16": "bf16 MFMA / fp32 accumulate",
    "bf16x6": "fp32 emulated on the bf16 matrix cores: plain GPT-2 GEMMs as 6 bf16 MFMAs per product of hi/mid/lo-split fp32 "
              "factors (terms below 2^-24 dropped), rest exact fp32",
    "bf16x3": "NOT exact fp32: plain GPT-2 GEMMs as 3 bf16 MFMAs per product of hi/lo-split fp32 factors (~1e-5), rest fp32",
}


def cpu_baseline(cfg, args):
    """SURVEY 8d protocol: the CPU oracle's full train step (forward + Huber + backward + clip + AdamW) in training
    mode (dropout masks drawn inside the step, p = 0.1 at every site, as the reference does), fp32, on the host cores
    this process may use, at B = --cpu-batch (BASELINE configs[0]: 2): ONE full warm-up step, then --cpu-steps (3)
    timed steps; value = B / median step time.  Also returns an eval forward of the same batch for the parity block."""
    from oracle import ref_cpu as R
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a 1-GPU box exposes every host core but grants a 16-core share; more threads only add contention
    cores = min(cores, int(os.environ.get("TECM_CPU_THREADS", 16)))
    torch.set_num_threads(cores)
    params = R.init_params(cfg, seed=0)
    p = {k: v.clone().requires_grad_(R.is_trainable(k)) for k, v in params.items()}
    train = [v for v in p.values() if v.requires_grad]
    opt = torch.optim.AdamW(train, lr=1e-4, weight_decay=1e-2)
    ei, _ = R.grid_graph()
    gwe = None if args.gat == "per_timestep" else 1
    B = args.cpu_batch
    x, tf, y = R.synthetic_batch(B, cfg["temporal_seq_len"], 2911, cfg["spatial_in_channels_base"],
                                 cfg["prediction_horizon"], seed=1234)
    with torch.no_grad():
        out_eval = R.forward(x, tf, ei, p, cfg, gwe)
    gen = torch.Generator().manual_seed(99)

    def step():
        t0 = time.perf_counter()
        masks = None if args.eval_mode else R.random_masks(cfg, B, ei, gwe, 0.1, gen)
        out = R.forward(x, tf, ei, p, cfg, gwe, masks=masks)
        loss = R.huber(out, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(train, 1.0)
        opt.step()
        return time.perf_counter() - t0

    warm = step()
    times = [step() for _ in range(max(1, args.cpu_steps))]
    med = sorted(times)[len(times) // 2]
    base = {"value": B / med, "unit": "samples/s", "cores": cores, "kind": "port",
            "step_s": [round(t, 2) for t in times], "warmup_step_s": round(warm, 2), "median_step_s": round(med, 2),
            "sample": f"{len(times)} timed full train steps (fwd+Huber+bwd+clip+AdamW) after 1 warm-up step of the fp32 "
                      f"PyTorch-CPU oracle at B={B} (BASELINE configs[0]), L_in={cfg['temporal_seq_len']}, N=2911, "
                      f"F={cfg['spatial_in_channels_base']}, gat={args.gat}, dropout "
                      f"{'off' if args.eval_mode else 'on (p=0.1, masks drawn inside the step)'}; median of the timed steps"}
    return base, (params, x, tf, ei, out_eval.detach())


def rmse_vs_ref(cfg, args, dev, ref):
    """BASELINE's "test RMSE vs ref": the HIP model (eval mode, same parameters, same batch) against the
    predictions of the CPU oracle; RMSE/MAE/R^2/Pearson per metrics.py:53-78 from the device metrics kernel, plus
    the max relative error of the 1e-3 parity bar."""
    from src.evaluation.metrics import HorizonMetrics
    from src.model.tec_mollm import TEC_MoLLM
    params, x, tf, ei, out_ref = ref
    mc = dict(cfg, gat_graphs=args.gat, include_wte=False, load_pretrained_gpt2=False, precision=args.precision)
    model = TEC_MoLLM(mc)
    model.load_state_dict(params, strict=True)
    model = model.to(dev).eval()
    B = x.shape[0]
    tfd = tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, x.shape[1], x.shape[2], 4)
    with torch.no_grad():
        out = model(x.to(dev), tfd, ei.to(dev))
    want = out_ref.to(dev)
    hm = HorizonMetrics(out.shape[1], None, device=dev)
    hm.update(out, want)
    m = hm.compute()
    rel = float((out - want).abs().max() / want.abs().max())
    return {"rmse_vs_ref": m["rmse_avg"], "mae_vs_ref": m["mae_avg"], "r2_vs_ref": m["r2_score_avg"],
            "pearson_vs_ref": m["pearson_r_avg"], "max_rel_err": rel, "ref_rms": float(want.pow(2).mean().sqrt()),
            "sample": f"eval forward at B={B} on the cpu_baseline batch, scaled units, {out.shape[1]} horizons"}


ALG_GFLOP_PER_SAMPLE = {48: 887.5, 96: 1822.6}    # fwd + bwd, dX-only for the frozen GPT-2 GEMMs, no recompute

# GPT-2 layer GEMMs (modeling_gpt2.py:262-310 + peft LoRA, modules.py:177-186) by (N, K, epilogue) of their M = B*T*N rows
_LAYER_SHAPES = [
    ("c_attn + LoRA-B fwd", lambda n, k, f: n == 2304 and k == 800),
    ("attn.c_proj fwd (+bias, dropout, residual)", lambda n, k, f: n == 768 and k == 768 and f["res"]),
    ("mlp.c_fc fwd (+bias, GELU, pre-activation)", lambda n, k, f: n == 3072 and k == 768 and f["pre"]),
    ("mlp.c_proj fwd (+bias, dropout, residual)", lambda n, k, f: n == 768 and k == 3072 and f["res"]),
    ("d mlp.c_proj (x GELU')", lambda n, k, f: n == 3072 and k == 768 and f["dact"]),
    ("d mlp.c_fc", lambda n, k, f: n == 768 and k == 3072 and not f["res"]),
    ("d attn.c_proj", lambda n, k, f: n == 768 and k == 768 and not f["res"]),
    ("d c_attn (+ d z)", lambda n, k, f: n == 800 and k == 2304),
]


def _other_site(M: int, n: int, k: int, batch: int, L_in: int) -> str:
    """Labels for the contractions outside the GPT-2 blocks (reference modules.py:36-41 the strided 1x1 conv, :114-116 the
    patch projection, :284-290 the head, :177-186 the LoRA matrices), by shape."""
    S, T = batch * 2911, L_in // 16
    Mt = S * T
    big = max(M, n, k)
    if big == S and M == S:
        return "head W1 fwd" if k == T * 768 else ("d head input" if n == T * 768 else "head (small)")
    if k == S:
        return "d head W1" if (M, n) == (576 * (T // 3 if T % 3 == 0 else 1), T * 768) or n == T * 768 else "d head W2"
    if M == Mt:
        return {(768, 512): "patch projection fwd", (512, 768): "d patch input", (32, 768): "lora_A fwd",
                (768, 32): "d lora_A input"}.get((n, k), "other (token rows)")
    if k == Mt:
        return {(768, 512): "d patch W", (2304, 32): "d lora_B", (32, 768): "d lora_A"}.get((M, n), "other weight gradient")
    if M in (Mt * 4, Mt * 8):
        return "1x1 conv fwd" if n < k else "d 1x1 conv input"
    if k in (Mt * 4, Mt * 8):
        return "d 1x1 conv W"
    return "other"


def _parse_detail(key: str):
    """'kernel M=.. N=.. K=.. win=000 drop=000 split=1 act=0 acc=0 dact=0 res=0 pre=0' -> (kernel, dict)."""
    parts = key.split(" ")
    kern = parts[0]
    f = {}
    for tok in parts[1:]:
        if "=" in tok:
            k, v = tok.split("=", 1)
            f[k] = v
    return kern, f


def latest_profile_json(precision: str, args):
    """The newest committed PMC traffic summary of this configuration (profiles/rNN_pmc_traffic_<prec>_B8.json)."""
    if args.batch != 8 or args.L_in != 48 or args.gat != "per_timestep" or precision not in ("fp32", "bf16"):
        return None, None
    for tag in ("r05", "r04", "r03", "r02", "r01"):
        rel = os.path.join("profiles", f"{tag}_pmc_traffic_{precision}_B8.json")
        try:
            with open(os.path.join(ROOT, rel)) as f:
                return json.load(f)["kernels"], rel
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_traffic(kernel: str, args, precision: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs of this same command, gfx950 corrections applied by tools/make_profiles.py); None when the
    committed passes do not cover this configuration."""
    kernels, rel = latest_profile_json(precision, args)
    k = (kernels or {}).get(kernel.split("<")[0] if kernel.startswith("gemm_bf16_dma6") else kernel)
    if k is None:
        return None
    return {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "fetch": k["fetch_bytes_per_launch"],
            "write": k["write_bytes_per_launch"], "source": rel}


def non_gemm_block(args, precision: str):
    """The largest kernels of the step that are NOT dense contractions -- HBM-bound by construction -- with the HBM-side
    bytes per launch and the rate from the committed PMC passes (counter traffic / average duration under the counters)."""
    kernels, rel = latest_profile_json(precision, args)
    if not kernels:
        return None
    rows = []
    for name, k in kernels.items():
        if "gemm" in name or "conv_fwd_seq" in name or "conv_dx_seq" in name or "conv_dw_seq" in name or k["avg_us_under_pmc"] <= 0:
            continue
        tot = k["avg_us_under_pmc"] * k["launches"]
        rate = k["hbm_bytes_per_launch"] / (k["avg_us_under_pmc"] * 1e-6) / 1e9
        rows.append((tot, {"kernel": name, "launches_in_trace": k["launches"], "avg_us": k["avg_us_under_pmc"],
                           "hbm_mb_per_launch": round(k["hbm_bytes_per_launch"] / 1e6, 1), "achieved_gbs": round(rate, 1),
                           "frac_of_hbm_peak": round(rate / HBM_PEAK_GBS, 3)}))
    rows.sort(key=lambda r: -r[0])
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "source": rel,
            "note": "counter traffic (FETCH_SIZE x2 + WRITE_SIZE) per launch / average launch duration in the PMC pass",
            "kernels": [r[1] for r in rows[:8]]}


def roofline_of(agg: dict, step_ms: float, steps: int, args, precision: str, batch: int, L_in: int):
    """agg: per-call-site GEMM records (ops.enable_gemm_timing(detail=True)) of `steps` steps taken in a SEPARATE short
    pass after the timed region (`value` carries no event bracketing).  Headline = the kernel with the largest total
    time: achieved = sum 2MNK / sum event time against the dense MFMA peak of its dtype.  `shapes` = every large GEMM
    call site with its own bound: max(flops / MFMA peak, algorithmic bytes / 8 TB/s)."""
    if not agg:
        return None
    by_kernel: dict = {}
    for key, a in agg.items():
        kern, _ = _parse_detail(key)
        t = by_kernel.setdefault(kern, {"ms": 0.0, "flops": 0.0, "n": 0})
        t["ms"] += a["ms"]; t["flops"] += a["flops"]; t["n"] += a["n"]
    name, a = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
    achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
    peak = BF16_MFMA_PEAK_TFLOPS if ("bf16" in name or "x3" in name) else F32_MFMA_PEAK_TFLOPS
    if "x3_kernel<2" in name:
        achieved *= 3.0                                # three bf16 MFMA products per fp32 product
    elif "x3_kernel<3" in name:
        achieved *= 6.0
    shapes = []
    for key, r in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        kern, f = _parse_detail(key)
        if "M" not in f or r["flops"] / r["n"] < 2e10 or not f.get("K", "").isdigit() or "seq" in kern:
            continue                                       # the sequence-tile conv kernels carry no operand byte count
        n_, k_ = int(f["N"]), int(f["K"])
        flags = {"res": f.get("res") == "1", "pre": f.get("pre") == "1", "dact": f.get("dact") == "1"}
        label = next((lab for lab, pred in _LAYER_SHAPES if int(f["M"]) == batch * 2911 * (L_in // 16) and pred(n_, k_, flags)), None)
        pk = BF16_MFMA_PEAK_TFLOPS if ("bf16" in kern or "x3" in kern) else F32_MFMA_PEAK_TFLOPS
        fl, by, us = r["flops"] / r["n"], r["bytes"] / r["n"], r["ms"] / r["n"] * 1e3
        t_mfma, t_hbm = fl / (pk * 1e12) * 1e6, by / (HBM_PEAK_GBS * 1e9) * 1e6
        shapes.append({"site": label or _other_site(int(f["M"]), n_, k_, batch, L_in), "kernel": kern, "M": int(f["M"]), "N": n_, "K": k_,
                       "launches_per_step": round(r["n"] / steps, 2), "avg_us": round(us, 1), "gflop": round(fl / 1e9, 1),
                       "algorithmic_mb": round(by / 1e6, 1), "t_mfma_us": round(t_mfma, 1), "t_hbm_us": round(t_hbm, 1),
                       "bound": "mfma" if t_mfma >= t_hbm else "hbm", "frac_of_bound": round(max(t_mfma, t_hbm) / us, 3),
                       "tflops": round(fl / us / 1e6, 1), "gbs": round(by / us / 1e3, 1)})
    gf = ALG_GFLOP_PER_SAMPLE.get(L_in) if getattr(args, "llm_layers", 3) == 3 else None
    step = None
    if gf is not None:
        tf = gf * batch / step_ms                      # GFLOP / ms = TFLOP/s
        step_peak = BF16_MFMA_PEAK_TFLOPS if precision in ("bf16", "bf16x3", "bf16x6") else F32_MFMA_PEAK_TFLOPS
        step = {"algorithmic_tflops": round(tf, 1), "peak": step_peak, "frac_of_peak": round(tf / step_peak, 4),
                "gflop_per_sample": gf, "note": "SURVEY 8d algorithmic flops (fwd + bwd, no recompute) x samples / step time"}
    all_ms = sum(v["ms"] for v in by_kernel.values()) / steps
    return {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": pmc_traffic(name, args, precision), "launches": a["n"],
            "avg_launch_ms": round(a["ms"] / a["n"], 4), "share_of_step": round(a["ms"] / steps / step_ms, 4),
            "all_gemm_share_of_step": round(all_ms / step_ms, 4),
            "timing": f"events on the launch stream around every GEMM in a separate pass of {steps} steps after the timed "
                      "region (the timed steps carry no event bracketing)",
            "step": step, "shapes": shapes[:24], "non_gemm": non_gemm_block(args, precision)}


def shape_pass(ts, batch_fn, ei, ew, steps: int = 3):
    """Per-call-site GEMM timing of `steps` extra steps (not part of `value`)."""
    from tecmollm import ops
    rec = ops.enable_gemm_timing(detail=True)
    for _ in range(steps):
        xb, tfb, yb = batch_fn()
        ts.step(xb, tfb, ei, ew, yb)
    agg = ops.summarize_gemm_timing(rec)
    ops.disable_gemm_timing()
    return agg, steps


def reference_loop(cfg, args, dev, ei, ew, B, steps=10, warmup=3, autocast=True, skip=()):
    """samples/s of the REFERENCE's training-loop body (train.py:57-112, :358-372, statement for statement) around the
    drop-in model: `torch.autocast(bf16)` (precision "auto" follows it), the per-step gradient_checkpointing_enable()
    call (a no-op here by design), GradScaler scale / unscale_ / step / update, clip_grad_norm_(model.parameters()),
    torch.optim.AdamW over the trainable parameters, CosineAnnealingWarmRestarts, loss.item() and empty_cache() every
    step, accumulation_steps = 1.  What a reference user gets WITHOUT adopting tecmollm.train.TrainStep.
    `skip` (tools/ref_loop_breakdown.py only; the bench runs the body whole) leaves out named statements -- "empty_cache",
    "item", "scaler", "clip" -- to price them."""
    from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts
    from src.model.tec_mollm import TEC_MoLLM
    from tecmollm.synthetic import synthetic_batch
    mc = dict(cfg, gat_graphs=args.gat, include_wte=False, load_pretrained_gpt2=False, precision="auto")
    torch.manual_seed(0)
    model = TEC_MoLLM(mc)
    with torch.no_grad():
        for blk in model.llm_backbone.trunk.h:
            blk.attn.c_attn.lora_B.default.weight.normal_(std=0.02)
    model = model.to(dev).train()
    optimizer = torch.optim.AdamW(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-4, weight_decay=1e-2)
    scheduler = CosineAnnealingWarmRestarts(optimizer, T_0=10, T_mult=2, eta_min=1e-7)
    loss_fn = torch.nn.HuberLoss(delta=1.0)
    scaler = torch.amp.GradScaler("cuda")
    accumulation_steps = 1
    L_in, L_out, H, W = cfg["temporal_seq_len"], cfg["prediction_horizon"], 41, 71
    x0, tf0, y0 = synthetic_batch(B, L_in, 2911, args.c_in, L_out, seed=1234)
    batch = {"x": x0.view(B, L_in, H, W, args.c_in).to(dev), "x_time_features": tf0[:, :, 0, :].contiguous().to(dev),
             "y": y0.view(B, L_out, H, W).permute(0, 2, 3, 1).contiguous().to(dev)}        # (B, H, W, L_out) as the dataset yields
    total_loss = 0.0

    def body(i):
        nonlocal total_loss
        x, y, time_features = batch["x"].to(dev), batch["y"].to(dev), batch["x_time_features"].to(dev)
        Bq, L, Hh, Ww, C = x.shape
        x = x.view(Bq, L, Hh * Ww, C)
        time_features = time_features.unsqueeze(-2).expand(Bq, L, Hh * Ww, -1)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=autocast):
            model.llm_backbone.model.gradient_checkpointing_enable()
            output = model(x, time_features, ei, ew)
            y_reshaped = y.permute(0, 3, 1, 2).reshape(Bq, -1, Hh * Ww, 1)
            loss = loss_fn(output, y_reshaped)
            loss = loss / accumulation_steps
        if "scaler" in skip:
            loss.backward()
        else:
            scaler.scale(loss).backward()
        del x, y, time_features, output, y_reshaped
        if "empty_cache" not in skip:
            torch.cuda.empty_cache()
        if (i + 1) % accumulation_steps == 0:
            if "scaler" not in skip:
                scaler.unscale_(optimizer)
            if "clip" not in skip:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            if "scaler" in skip:
                optimizer.step()
            else:
                scaler.step(optimizer)
                scaler.update()
            optimizer.zero_grad()
            scheduler.step()
        if "item" not in skip:
            total_loss += loss.item() * accumulation_steps

    optimizer.zero_grad()
    for i in range(warmup):
        body(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        body(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"samples_per_s": round(steps * B / dt, 2), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "batch": B,
           "dtype": "bf16" if autocast else "f32", "final_loss": round(total_loss / (steps + warmup), 5),
           "workload": "the reference's loop body (train.py:57-112) around the drop-in model: autocast bf16, GradScaler, "
                       "torch AdamW + clip_grad_norm_, loss.item() + empty_cache() per step"}
    del model, optimizer
    torch.cuda.empty_cache()
    return res


def extra_config(cfg, args, dev, mode, ei, ew, workload, L_in=None, L_out=None, with_roofline=True, steps=10, batch=None):
    """samples/s of one more BASELINE configuration, measured after the main run with its own roofline:
    mode "bf16" = autocast semantics (BASELINE configs[2]); L_in / L_out = the stress shape of configs[4] on one GPU;
    "bf16x6" / "bf16x3" (only with --emulation-modes) = fp32 emulated by six / three bf16 MFMAs per product."""
    from src.model.tec_mollm import TEC_MoLLM
    from tecmollm import ops
    from tecmollm.synthetic import synthetic_batch
    from tecmollm.train import TrainStep
    a2 = argparse.Namespace(**vars(args))
    a2.L_in, a2.L_out = L_in or args.L_in, L_out or args.L_out
    cfg2 = make_config(a2)
    mc = dict(cfg2, gat_graphs=args.gat, include_wte=False, load_pretrained_gpt2=False, precision=mode)
    torch.manual_seed(0)
    model = TEC_MoLLM(mc)
    with torch.no_grad():
        for blk in model.llm_backbone.trunk.h:
            blk.attn.c_attn.lora_B.default.weight.normal_(std=0.02)
    model = model.to(dev)
    model.train(not args.eval_mode)
    B = batch or args.batch
    x, tf, y = synthetic_batch(B, a2.L_in, 2911, args.c_in, a2.L_out, seed=1234)
    x, y = x.to(dev), y.to(dev)
    tf = tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, a2.L_in, 2911, 4)
    ts = TrainStep(model, world_size=1)
    for _ in range(3):
        ts.step(x, tf, ei, ew, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step(x, tf, ei, ew, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"samples_per_s": round(steps * B / dt, 2), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
           "dtype": {"fp32": "f32"}.get(mode, mode), "workload": workload}
    if with_roofline:
        a2.precision = mode
        agg, nsteps = shape_pass(ts, lambda: (x, tf, y), ei, ew)
        res["roofline"] = roofline_of(agg, dt / steps * 1e3, nsteps, a2, mode, B, a2.L_in)
    del ts, model
    torch.cuda.empty_cache()
    return res


class WindowFeed:
    """--data window: a synthetic series resident in HBM (T time steps of the 41 x 71 grid) behind the device mirror of
    the reference's SlidingWindowSamplerDataset (src/data/dataset.py:65-99); every step draws its batch of B windows
    with ONE launch of tecm_window_batch, as the DataLoader + collate + H2D + reshapes of train.py:57-65 would."""

    def __init__(self, args, dev, rank, steps_total):
        from src.data.dataset import SlidingWindowSamplerDataset
        T = args.L_in + args.L_out + 256
        g = torch.Generator().manual_seed(4321 + rank)
        X = torch.randn(T, 41, 71, args.c_in, generator=g)
        Y = torch.randn(T, 41, 71, args.L_out, generator=g)
        hours = torch.arange(T)
        tfeat = torch.stack([(hours % 24) // 2, (hours // 24) % 366, (hours // (24 * 366)) % 13,
                             ((hours // 24) % 366) // 92], -1).float()
        self.ds = SlidingWindowSamplerDataset.from_tensors(X, Y, tfeat, args.L_in, args.L_out, stride=1, device=dev)
        self.B = args.batch
        perm = torch.randperm(len(self.ds), generator=g)
        need = steps_total * self.B
        self.order = perm.repeat((need + len(perm) - 1) // len(perm))[:need].tolist()
        self.i = 0
        self.events = []

    def next(self, timed: bool):
        idx = self.order[self.i * self.B:(self.i + 1) * self.B]
        self.i += 1
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        x, tf, y = self.ds.batch(idx)
        if timed:
            e1.record()
            self.events.append((e0, e1))
        return x, tf, y

    def ms_per_step(self):
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.events) / max(len(self.events), 1)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))                   # nothing above this line has touched the GPU
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (the HIP path has no CPU fallback)")
    backend = os.environ.get("TECM_DIST_BACKEND", "nccl") if world > 1 else None
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPUs are visible (RCCL: one rank per device)")
    local = local % max(ndev, 1)                       # gloo rehearsal of the N-rank path on a 1-GPU box shares the card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_info = None
    if world > 1:
        # backend "nccl" IS RCCL on ROCm; TECM_DIST_BACKEND=gloo only exists to rehearse the multi-rank code path
        # on a single-GPU box (two ranks cannot share one device under RCCL)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                          # proof that the collective saw every rank
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "allreduce_of_ones": float(ones.item()),
                     "launcher": os.environ.get("TECM_LAUNCHER", "torchrun")}
        if dist_info["world_size"] != args.gpus or dist_info["allreduce_of_ones"] != float(args.gpus):
            raise SystemExit(f"process group is not {args.gpus} ranks wide: {dist_info}")

    from src.model.tec_mollm import TEC_MoLLM
    from tecmollm import check_device_errors, ops
    from tecmollm.train import TrainStep
    from tecmollm.synthetic import grid_graph, synthetic_batch

    cfg = make_config(args)
    if args.loop == "reference":                                    # diagnostics: the reference's loop body alone (N = 1)
        if world != 1:
            raise SystemExit("--loop reference is a single-GPU measurement")
        eig, ewg = grid_graph()
        r = reference_loop(cfg, args, dev, eig.to(dev), ewg.to(dev), args.batch, steps=args.steps, warmup=args.warmup,
                           autocast=args.precision != "fp32")
        print(json.dumps({"metric": "train samples/sec", "value": r["samples_per_s"], "unit": "samples/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": r["dtype"],
                          "data": "synthetic", "config": {"workload": r["workload"], "global_batch": args.batch,
                                                          "parallelism": "dp1", "loop": "reference"}}), flush=True)
        return
    mc = dict(cfg, gat_graphs=args.gat, include_wte=False, load_pretrained_gpt2=False, precision=args.precision)
    torch.manual_seed(0)                                           # identical weights on every rank ...
    model = TEC_MoLLM(mc)
    with torch.no_grad():                                          # exercise the LoRA path (peft inits B = 0)
        for blk in model.llm_backbone.trunk.h:
            blk.attn.c_attn.lora_B.default.weight.normal_(std=0.02)
    model = model.to(dev)
    model.train(not args.eval_mode)
    torch.manual_seed(1234 + rank)                                 # per-rank data and dropout streams
    B = args.batch
    x, tf, y = synthetic_batch(B, args.L_in, 2911, args.c_in, args.L_out, seed=1234 + rank)
    x, y = x.to(dev), y.to(dev)
    tf = tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, args.L_in, 2911, 4)   # train.py:65
    ei, ew = grid_graph()
    ei, ew = ei.to(dev), ew.to(dev)
    t_setup = time.perf_counter()
    from tecmollm import graph as graph_
    graph_.get(ei, 2911, dev, 22 - args.c_in)          # host-side CSR / tile windows of the graph (cached per process)
    host_graph_s = time.perf_counter() - t_setup
    ts = TrainStep(model, world_size=world, time_collective=True)   # rank 0's parameters broadcast (train.py:354)
    feed = WindowFeed(args, dev, rank, args.warmup + args.steps + 3) if args.data == "window" else None   # + the shape pass

    def batch(timed):
        return feed.next(timed) if feed is not None else (x, tf, y)

    def barrier():
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    step = ts.step_graphed if args.graph else ts.step
    if args.graph and args.warmup < 2:
        raise SystemExit("--graph needs --warmup >= 2 (one eager step, one that records)")
    for _ in range(args.warmup):
        xb, tfb, yb = batch(False)
        step(xb, tfb, ei, ew, yb)
    barrier()
    ts.reset_collective_timing()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        xb, tfb, yb = batch(True)
        loss = step(xb, tfb, ei, ew, yb)
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    check_device_errors(dev, sync=True)                # bad time indices / diverged ranks reported by any step
    # the roofline's per-GEMM events are collected in a SEPARATE short pass: `value` above carries no event bracketing
    shape_agg = None
    if not args.no_kernel_timing:                      # every rank runs it: the steps contain the collective
        shape_agg = shape_pass(ts, lambda: batch(False), ei, ew)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]      # this rank's device time per step
    if world > 1:
        coll = ts.collective_ms()
        mine = torch.tensor([sum(step_ms) / len(step_ms), min(step_ms), max(step_ms),
                             sum(coll) / max(len(coll), 1), max(coll) if coll else 0.0], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        dist_info["allreduce_ms"] = {"mean_over_ranks": round(float(allr[:, 3].mean()), 4),
                                     "max_over_ranks": round(float(allr[:, 4].max()), 4),
                                     "per_rank_mean": [round(float(v), 4) for v in allr[:, 3]],
                                     "bytes": int(ts.flat_grad_ext.numel() * 4), "launches": len(coll),
                                     "note": "events on the launch stream around the ONE all-reduce of the step (flat "
                                             "fp32 gradient + per-rank parameter checksums); includes waiting for the "
                                             "slowest rank to arrive"}
        dist_info["step_ms_per_rank"] = {"mean": [round(float(v), 3) for v in allr[:, 0]],
                                         "min_over_ranks": round(float(allr[:, 0].min()), 3),
                                         "max_over_ranks": round(float(allr[:, 0].max()), 3),
                                         "slowest_single_step": round(float(allr[:, 2].max()), 3)}
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        csum = ts._param_checksum().reshape(1)
        lo, hi = csum.clone(), csum.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist_info["param_checksum_min_eq_max"] = bool((lo == hi).item())
        if not dist_info["param_checksum_min_eq_max"]:
            raise SystemExit(f"ranks diverged: parameter checksum min {lo.item()!r} != max {hi.item()!r}")
    dt = float(tmax.item())

    if rank == 0:
        total = B * world * args.steps
        roof = roofline_of(shape_agg[0], dt / args.steps * 1e3, shape_agg[1], args, args.precision, B, args.L_in) \
            if shape_agg is not None else None
        line = {
            "metric": "train samples/sec", "value": round(total / dt, 3), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3", "bf16x6": "bf16x6"}[args.precision], "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{2 if args.precision == 'bf16' else 1}]: B={B}/GPU, "
                                   f"L_in={args.L_in}, L_out={args.L_out}, N=2911, "
                                   f"F={args.c_in} (d_emb={22 - args.c_in}), {args.llm_layers} GPT-2 blocks, full fwd+bwd+AdamW, "
                                   f"{PRECISION_TEXT[args.precision]}, "
                                   f"GATv2 {args.gat}, dropout {'off' if args.eval_mode else 'on (p=0.1)'}",
                       "global_batch": B * world, "parallelism": f"dp{world}", "final_loss": round(float(loss), 5),
                       "peak_hbm_gb_per_gpu": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                       "data_feed": args.data, "host_graph_build_s": round(host_graph_s, 3),
                       "launch": "hipGraph replay (TrainStep.step_graphed)" if args.graph else "eager",
                       "step_ms_min_max": [round(min(step_ms), 3), round(max(step_ms), 3)]},
            "roofline": roof,
        }
        if feed is not None:
            d_ms = feed.ms_per_step()
            line["config"]["data_feed_ms_per_step"] = round(d_ms, 4)
            line["config"]["data_feed_share_of_step"] = round(d_ms / (dt / args.steps * 1e3), 5)
        if dist_info is not None:
            line["config"]["dist"] = dist_info
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], ref = cpu_baseline(cfg, args)
            line["parity"] = rmse_vs_ref(cfg, args, dev, ref)
        default_shape = args.L_in == 48 and args.L_out == 12 and args.data == "fixed"
        if world == 1 and args.precision == "fp32" and default_shape and not args.no_other_precisions:
            # `value` above is the exact-fp32 number (BASELINE configs[1]).  Two more BASELINE configurations are
            # measured here, each with its own roofline: configs[2] (bf16 autocast semantics, same workload) and the
            # per-GPU shape of configs[4] (L_in = 96 / L_out = 24: twice the rows and tokens per sample), fp32.
            del ts, model
            torch.cuda.empty_cache()
            line["configs_extra"] = {
                "bf16": extra_config(cfg, args, dev, "bf16", ei, ew,
                                     "BASELINE configs[2]: the same step with every dense contraction the bf16 kernel "
                                     "serves on the bf16 matrix cores (operands rounded to bf16, fp32 accumulate; "
                                     "norms / softmax / GATv2 fp32)"),
                "L96": extra_config(cfg, args, dev, "fp32", ei, ew,
                                    f"BASELINE configs[4] per-GPU shape: B={B}, L_in=96, L_out=24 (6 tokens per "
                                    f"sequence, head 4608 -> 1152 -> 24), N=2911, F={args.c_in}, fp32, full "
                                    f"fwd+bwd+AdamW, GATv2 {args.gat}, dropout on", L_in=96, L_out=24)}
            # the reference's own loop body around the drop-in model (what train.py gets without adopting TrainStep),
            # at B = 8 and at the reference's per-GPU batch B = 2 (scripts/train_2gpu.sh:4-12), beside the native
            # TrainStep at the same batch: `ratio` = reference-loop rate / native rate
            ref_loop = {}
            for bq in (B, 2):
                r = reference_loop(cfg, args, dev, ei, ew, bq)
                nat = line["configs_extra"]["bf16"] if bq == B else \
                    extra_config(cfg, args, dev, "bf16", ei, ew, f"native TrainStep, bf16, B={bq}", with_roofline=False, batch=bq)
                # ... and the same body without its per-step torch.cuda.empty_cache() (train.py:81): that one statement returns
                # the step's activations to the driver and buys them back, 6-190 ms depending on the allocator's state
                r2 = reference_loop(cfg, args, dev, ei, ew, bq, skip=("empty_cache",))
                ref_loop[f"B{bq}"] = {"samples_per_s": r["samples_per_s"], "ms_per_step": r["ms_per_step"],
                                      "native_samples_per_s": nat["samples_per_s"], "native_ms_per_step": nat["ms_per_step"],
                                      "ratio": round(r["samples_per_s"] / nat["samples_per_s"], 3),
                                      "ms_per_step_without_empty_cache": r2["ms_per_step"],
                                      "ratio_without_empty_cache": round(r2["samples_per_s"] / nat["samples_per_s"], 3)}
            line["configs_extra"]["reference_loop_bf16"] = ref_loop
            if args.emulation_modes:
                line["other_precisions"] = {m: extra_config(cfg, args, dev, m, ei, ew, PRECISION_TEXT[m],
                                                            with_roofline=False, steps=5)
                                            for m in ("bf16x6", "bf16x3")}
        emit(line, args.detail_json)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
