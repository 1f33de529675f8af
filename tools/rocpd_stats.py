#!/usr/bin/env python3
"""Per-kernel statistics out of a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`
writes NAME_results.db on ROCm 7.2): calls, total / average / min / max duration, share of the GPU time.

    python tools/rocpd_stats.py <results.db> [--md] [--top N] [--skip-first K]
"""
import argparse
import re
import sqlite3


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    return name.split("::")[-1] if "<" not in name.split("::")[-1] or True else name


def load(path, skip_first=0):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    agg = {}
    seen = {}
    for n, s, e in rows:
        k = short(n)
        seen[k] = seen.get(k, 0) + 1
        if seen[k] <= skip_first:
            continue
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--md", action="store_true")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--skip-first", type=int, default=0, help="drop the first K launches of every kernel (warm-up)")
    a = ap.parse_args()
    agg = load(a.db, a.skip_first)
    tot = sum(v[1] for v in agg.values())
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]
    if a.md:
        print("| kernel | calls | total us | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
        for k, (n, t, mn, mx) in items:
            print(f"| `{k}` | {n} | {t:.0f} | {t / n:.1f} | {mn:.1f} | {mx:.1f} | {100 * t / tot:.1f} |")
    else:
        for k, (n, t, mn, mx) in items:
            print(f"{k[:70]:70s} n={n:5d} total={t:10.0f}us avg={t / n:9.1f} min={mn:9.1f} max={mx:9.1f} {100 * t / tot:5.1f}%")
    print(f"\ntotal kernel time {tot / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
