#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one profiling session (tools/capture_profiles.sh) into the committed summaries
under profiles/.

    python tools/make_profiles.py gpurun_out/prof_r02_fp32 r02 fp32

<dir> holds stats/ (--kernel-trace --stats of `bench.py --steps 3 --warmup 1 --no-cpu-baseline`) with stats.log,
fetch/ and write/ (--pmc FETCH_SIZE / WRITE_SIZE passes) and mfma/ (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE),
each collected in its own run as MI355X_MICROARCH.md prescribes; every pass is a rocpd SQLite database (p_results.db).
gfx950 corrections of the guide's HBM section: the counters are in KiB; FETCH_SIZE reports half of the bytes of wide
coalesced reads (x2); WRITE_SIZE is exact for 16-byte stores."""
import collections
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8.0e12


def short(name: str) -> str:
    m = re.search(r"(gemm_bf16_kernel|gemm_x3_kernel|gemm_kernel)<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(2).split(",")]
        if m.group(1) == "gemm_bf16_kernel":
            a = a[:4]
        return f"{m.group(1)}<{','.join(a)}>"
    if "gemm_bf16_dma_kernel" in name:
        return "gemm_bf16_dma_kernel"
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    return name.split("::")[-1].strip()[:70]


def kernels(db_path):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    agg = collections.OrderedDict()
    for n, s, e in db.execute(f"select {name_col}, start, end from kernels order by start"):
        a = agg.setdefault(short(n), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    return agg


def counters(db_path):
    """{kernel: {counter: [launches, sum, total_us]}}"""
    db = sqlite3.connect(db_path)
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    # one row per (dispatch, counter, hardware instance): sum the instances of a dispatch first
    for n, s, e, c, v in db.execute("select name, min(start), max(end), counter_name, sum(counter_value) from pmc_events "
                                    "group by dispatch_id, counter_name"):
        a = agg[short(n)][c]
        a[0] += 1
        a[1] += float(v)
        a[2] += (e - s) / 1e3
    return agg


def main():
    d, tag, prec = sys.argv[1], sys.argv[2], sys.argv[3]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    line = [l for l in open(f"{d}/stats.log") if l.startswith("{") and "samples/sec" in l][0]
    j = json.loads(line)
    steps = j["steps"] + j["warmup"]
    B = j["config"]["global_batch"]

    # ---- kernel statistics
    ks = kernels(f"{d}/stats/p_results.db")
    if "adamw_kernel" in ks:             # one optimizer launch per step: counts the steps of bench.py's per-shape GEMM timing pass
        steps = ks["adamw_kernel"][0]    # (3 more after the timed region) as well
    tot = sum(v[1] for v in ks.values())
    out = [f"# Round {tag[1:]} -- rocprofv3 --kernel-trace --stats of `python3 bench.py --precision {prec} --steps {j['steps']} "
           f"--warmup {j['warmup']} --no-cpu-baseline --no-other-precisions`\n",
           f"B={B}, {prec} (BASELINE configs[{2 if prec == 'bf16' else 1}]), GATv2 per_timestep, dropout on; {steps} steps in the "
           f"trace; bench line of the same run: {j['value']} samples/s, {j['ms_per_step']} ms/step, `roofline.kernel = "
           f"{j['roofline']['kernel']}`, {j['roofline']['achieved']} TFLOP/s from in-bench events over "
           f"{j['roofline']['launches']} launches (avg {j['roofline']['avg_launch_ms']} ms).\n",
           "| kernel | calls | ms/step | avg us | % |", "|---|---:|---:|---:|---:|"]
    with open(f"{prof}/{tag}_kernel_stats_{prec}_B8.csv", "w") as f:
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for k, (n, t) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{k}\",{n},{t:.1f},{t / n:.2f},{100 * t / tot:.3f}\n")
            if t / tot >= 0.0004:
                out.append(f"| `{k}` | {n} | {t / steps / 1e3:.2f} | {t / n:.1f} | {100 * t / tot:.1f} |")
    out.append(f"\nTotal GPU kernel time {tot / 1e3:.1f} ms over {steps} steps = {tot / steps / 1e3:.1f} ms/step.\n")
    dom = j["roofline"]["kernel"]
    if dom in ks:
        n, t = ks[dom]
        out.append(f"Roofline kernel `{dom}`: {n} launches, average {t / n:.1f} us per launch in this trace vs "
                   f"{j['roofline']['avg_launch_ms'] * 1e3:.1f} us from bench.py's events.\n")
    open(f"{prof}/{tag}_kernel_stats_{prec}_B8.md", "w").write("\n".join(out))
    print("\n".join(out[4:22]))

    # ---- HBM-side traffic per kernel
    fetch, write = counters(f"{d}/fetch/p_results.db"), counters(f"{d}/write/p_results.db")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        fc, wc = fetch[k].get("FETCH_SIZE", [0, 0.0, 0.0]), write[k].get("WRITE_SIZE", [0, 0.0, 0.0])
        n = fc[0] or wc[0]
        fb = 2.0 * 1024.0 * fc[1] / max(fc[0], 1)
        wb = 1024.0 * wc[1] / max(wc[0], 1)
        res[k] = {"launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                  "hbm_bytes_per_launch": round(fb + wb), "avg_us_under_pmc": round(fc[2] / max(fc[0], 1), 1)}
    json.dump({"note": "FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes; separate --pmc passes of "
                       f"`bench.py --precision {prec} --steps 1 --warmup 1 --no-cpu-baseline --no-other-precisions "
                       "--no-kernel-timing`", "kernels": res},
              open(f"{prof}/{tag}_pmc_traffic_{prec}_B8.json", "w"), indent=1)

    # algorithmic bytes of the two HBM-bound kernels north_star singles out: 4*L*N*(Cin + C) per sample
    L, N, C, Cin = 48, 2911, 22, 10
    alg = {"spatial_fwd_kernel": 4 * L * N * (Cin + C) * B, "spatial_bwd_kernel<1>": 4 * L * N * (C + Cin) * B,
           "spatial_bwd_kernel<2>": 4 * L * N * (C + Cin) * B}
    for cin in (6, 10):                                  # round 5: the second formulations (template argument = C_in)
        alg[f"spatial_fwd2_kernel<{cin}>"] = alg[f"spatial_bwd2_kernel<{cin}>"] = 4 * L * N * (Cin + C) * B
    md = [f"# Round {tag[1:]} -- L2-miss (HBM-side) traffic and rate per kernel, {prec}\n",
          "From the `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes (separate runs; KiB -> bytes, FETCH x2 on gfx950) of "
          f"`python3 bench.py --precision {prec} --steps 1 --warmup 1 --no-cpu-baseline --no-other-precisions "
          "--no-kernel-timing` (B=8; 2 steps in each trace).  Rate = (fetch + write) per launch / average kernel duration in "
          "the FETCH pass; fraction of the 8 TB/s HBM3E spec (the guide measures ~6.3 TB/s achievable).  FETCH_SIZE counts "
          "L2 misses, whether the Infinity Cache or HBM serves them.  `alg MB` = algorithmic bytes per launch (SURVEY 8d) "
          "for the two kernels whose bound is HBM by byte count; `alg frac` = alg bytes / duration / 8 TB/s.\n",
          "| kernel | launches | avg us | fetch MB | write MB | TB/s | of 8 TB/s | alg MB | alg frac |",
          "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    rows = sorted(res.items(), key=lambda kv: -kv[1]["avg_us_under_pmc"] * kv[1]["launches"])
    for k, v in rows[:28]:
        us = v["avg_us_under_pmc"]
        if us <= 0:
            continue
        rate = v["hbm_bytes_per_launch"] / (us * 1e-6)
        a = alg.get(k)
        md.append(f"| `{k}` | {v['launches']} | {us:.1f} | {v['fetch_bytes_per_launch'] / 1e6:.1f} | "
                  f"{v['write_bytes_per_launch'] / 1e6:.1f} | {rate / 1e12:.2f} | {rate / HBM_PEAK:.2f} | "
                  f"{'' if a is None else f'{a / 1e6:.1f}'} | {'' if a is None else f'{a / (us * 1e-6) / HBM_PEAK:.3f}'} |")
    open(f"{prof}/{tag}_hbm_rates_{prec}_B8.md", "w").write("\n".join(md) + "\n")

    # ---- MFMA pipe occupancy and shader clock
    mf = counters(f"{d}/mfma/p_results.db")
    md = [f"# Round {tag[1:]} -- MFMA pipe occupancy and shader clock per kernel, {prec}\n",
          "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` (its own pass, no other trace domains) of",
          f"`python3 bench.py --precision {prec} --steps 1 --warmup 1 --no-cpu-baseline --no-other-precisions "
          "--no-kernel-timing` (B=8).  GRBM_GUI_ACTIVE is summed over the 8 XCDs:",
          "clock = GRBM_GUI_ACTIVE / 8 / kernel time.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs:",
          "MFMA-busy fraction = busy / (GRBM_GUI_ACTIVE / 8) / 1024.\n",
          "| kernel | launches | time ms | shader clock GHz | MFMA pipe busy |", "|---|---:|---:|---:|---:|"]
    rows = sorted(mf.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 0, 0])[2])
    for k, v in rows[:16]:
        g = v.get("GRBM_GUI_ACTIVE")
        b = v.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if not g or not b or ("gemm" not in k and "spatial" not in k):
            continue
        gui = g[1] / 8
        md.append(f"| `{k}` | {g[0]} | {g[2] / 1e3:.2f} | {gui / max(g[2] * 1e3, 1):.2f} | {b[1] / max(gui, 1) / 1024:.2f} |")
    open(f"{prof}/{tag}_pmc_mfma_{prec}_B8.md", "w").write("\n".join(md) + "\n")
    print("\n".join(md[6:16]))


if __name__ == "__main__":
    main()
