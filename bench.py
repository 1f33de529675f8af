#!/usr/bin/env python3
"""Throughput of one TEC-MoLLM training step on N MI355X (BASELINE.json metric: train samples/sec on
synthetic (B,48,2911,10) -> (B,12,2911,1) batches).

    python bench.py [--gpus N --steps K --warmup W]          # N > 1: starts N ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                # or under torchrun: RANK / WORLD_SIZE come from the env

A step = forward + HuberLoss + backward + (RCCL all-reduce of the flat 12.3 MB gradient, the per-rank parameter
checksums riding in its tail) + clip(1.0) + AdamW + cosine-warm-restart scheduler, training mode (every dropout
site active), inputs resident in HBM.  Weak scaling: every rank processes `--batch` (default 8) samples;
value = all samples / max-over-ranks time.  Rank 0 prints ONE JSON line carrying `roofline` (dominant kernel,
measured with events on the launch stream inside the timed region) and, at N=1, `cpu_baseline` (the CPU oracle's
train step timed on the host cores of this box) and `configs_extra.bf16` (BASELINE configs[2] with its own roofline).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (BASELINE configs[1]: B=8)")
    ap.add_argument("--L_in", type=int, default=48)
    ap.add_argument("--L_out", type=int, default=12)
    ap.add_argument("--c_in", type=int, default=10, help="raw feature width F (BASELINE: 10 -> d_emb 12)")
    ap.add_argument("--gat", choices=["per_timestep", "reference"], default="per_timestep")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16x3", "bf16x6"], default="fp32",
                    help="fp32 = BASELINE configs[1] (exact-f32 MFMA); bf16 = configs[2] (bf16 MFMA, fp32 accumulate)")
    ap.add_argument("--eval-mode", action="store_true", help="dropout off (diagnostics only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2, help="CPU baseline batch (BASELINE configs[0]: B=2)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU train steps after one warm-up step")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-precisions", action="store_true",
                    help="skip the short runs of the other precision modes (N=1, default flags only)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ N-rank launcher
def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes (one per GPU) of this same script and
    relay rank 0's JSON line.  Runs BEFORE anything in this process touches the GPU -- a process that has initialised
    HIP must never fork/exec GPU children -- and never replaces itself: the children are ordinary subprocesses, this
    process waits for them and exits with the first non-zero code (killing the exact PIDs it started)."""
    n = args.gpus
    backend = os.environ.get("TECM_DIST_BACKEND", "nccl")
    visible = torch.cuda.device_count()            # counting devices does not initialise HIP on this image
    if backend == "nccl" and visible < n:
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
        "temporal_strides": [2, 2], "patch_len": patch_len, "d_llm": 768, "llm_layers": 3,
        "prediction_horizon": args.L_out, "temporal_seq_len": args.L_in, "num_years": 13,
    }


PRECISION_TEXT = {
    "fp32": "fp32",
    "bf16": "bf16 MFMA / fp32 accumulate",
    "bf16x6": "fp32 emulated on the bf16 matrix cores: plain GPT-2 GEMMs as 6 bf16 MFMAs per product of hi/mid/lo-split fp32 "
              "factors (terms below 2^-24 dropped), rest exact fp32",
    "bf16x3": "NOT exact fp32: plain GPT-2 GEMMs as 3 bf16 MFMAs per product of hi/lo-split fp32 factors (~1e-5), rest fp32",
}


def pmc_traffic(kernel: str, args, precision: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs of this same command, gfx950 corrections applied by tools/pmc_traffic.py); None when the
    committed passes do not cover this configuration."""
    if args.batch != 8 or args.L_in != 48 or args.gat != "per_timestep" or precision not in ("fp32", "bf16"):
        return None
    for tag in ("r02", "r01"):
        rel = os.path.join("profiles", f"{tag}_pmc_traffic_{precision}_B8.json")
        try:
            with open(os.path.join(ROOT, rel)) as f:
                k = json.load(f)["kernels"].get(kernel)
        except (OSError, KeyError, ValueError):
            continue
        if k is not None:
            return {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "fetch": k["fetch_bytes_per_launch"],
                    "write": k["write_bytes_per_launch"], "source": rel}
    return None


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


def roofline_of(agg: dict, dt_s: float, args, precision: str):
    """The GEMM variant with the largest total time inside the timed region: achieved = sum 2MNK / sum event time."""
    if not agg:
        return None
    name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
    achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
    peak = BF16_MFMA_PEAK_TFLOPS if ("bf16" in name or "x3" in name) else F32_MFMA_PEAK_TFLOPS
    if "x3_kernel<2" in name:
        achieved *= 3.0                                # three bf16 MFMA products per fp32 product
    elif "x3_kernel<3" in name:
        achieved *= 6.0
    return {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": pmc_traffic(name, args, precision), "launches": a["n"],
            "avg_launch_ms": round(a["ms"] / a["n"], 4), "share_of_step": round(a["ms"] / (dt_s * 1e3), 4),
            "all_gemm_share_of_step": round(sum(v["ms"] for v in agg.values()) / (dt_s * 1e3), 4)}


def other_precision(cfg, args, dev, mode, x, tf, ei, ew, y, with_roofline=False):
    """samples/s of the same step in another precision mode of the library:
    "bf16" = autocast semantics (BASELINE configs[2]; reported with its own roofline), "bf16x6" / "bf16x3" = fp32
    emulated by six / three bf16 MFMAs per product (opt-in modes, reported for context only)."""
    from src.model.tec_mollm import TEC_MoLLM
    from tecmollm import ops
    from tecmollm.train import TrainStep
    mc = dict(cfg, gat_graphs=args.gat, include_wte=False, load_pretrained_gpt2=False, precision=mode)
    torch.manual_seed(0)
    model = TEC_MoLLM(mc)
    with torch.no_grad():
        for blk in model.llm_backbone.trunk.h:
            blk.attn.c_attn.lora_B.default.weight.normal_(std=0.02)
    model = model.to(dev)
    model.train(not args.eval_mode)
    ts = TrainStep(model, world_size=1)
    n = 10 if with_roofline else 5
    for _ in range(3):
        ts.step(x, tf, ei, ew, y)
    torch.cuda.synchronize()
    prof = ops.enable_gemm_timing() if with_roofline else None
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(x, tf, ei, ew, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.disable_gemm_timing()
    res = {"samples_per_s": round(n * x.shape[0] / dt, 2), "ms_per_step": round(dt / n * 1e3, 2), "steps": n}
    if with_roofline:
        res["dtype"] = "bf16"
        res["workload"] = "BASELINE configs[2]: the same step with every dense contraction the bf16 kernel serves on the " \
                          "bf16 matrix cores (operands rounded to bf16, fp32 accumulate; norms / softmax / GATv2 fp32)"
        res["roofline"] = roofline_of(ops.summarize_gemm_timing(prof), dt, args, "bf16")
    return res


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
    ts = TrainStep(model, world_size=world)            # ... and rank 0's parameters broadcast anyway (train.py:354)

    def barrier():
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ts.step(x, tf, ei, ew, y)
    barrier()
    prof = None if args.no_kernel_timing else ops.enable_gemm_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = ts.step(x, tf, ei, ew, y)
    barrier()
    dt = time.perf_counter() - t0
    ops.disable_gemm_timing()
    check_device_errors(dev, sync=True)                # bad time indices / diverged ranks reported by any step
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
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
        roof = roofline_of(ops.summarize_gemm_timing(prof), dt, args, args.precision) if prof is not None else None
        line = {
            "metric": "train samples/sec", "value": round(total / dt, 3), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3", "bf16x6": "bf16x6"}[args.precision], "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{2 if args.precision == 'bf16' else 1}]: B={B}/GPU, "
                                   f"L_in={args.L_in}, L_out={args.L_out}, N=2911, "
                                   f"F={args.c_in} (d_emb={22 - args.c_in}), full fwd+bwd+AdamW, "
                                   f"{PRECISION_TEXT[args.precision]}, "
                                   f"GATv2 {args.gat}, dropout {'off' if args.eval_mode else 'on (p=0.1)'}",
                       "global_batch": B * world, "parallelism": f"dp{world}", "final_loss": round(float(loss), 5),
                       "peak_hbm_gb_per_gpu": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)},
            "roofline": roof,
        }
        if dist_info is not None:
            line["config"]["dist"] = dist_info
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], ref = cpu_baseline(cfg, args)
            line["parity"] = rmse_vs_ref(cfg, args, dev, ref)
        if world == 1 and args.precision == "fp32" and not args.no_other_precisions:
            # `value` above is the exact-fp32 number (BASELINE configs[1]).  configs[2] (bf16 autocast semantics) is
            # measured here on the same workload with its own roofline; the emulation modes are context only.
            line["configs_extra"] = {"bf16": other_precision(cfg, args, dev, "bf16", x, tf, ei, ew, y, True)}
            line["other_precisions"] = {m: other_precision(cfg, args, dev, m, x, tf, ei, ew, y)
                                        for m in ("bf16x6", "bf16x3")}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
