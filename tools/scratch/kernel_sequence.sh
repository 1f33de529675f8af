#!/bin/bash
# order of the kernels of the last traced bf16 step, to see what the tiny runtime copy kernels sit between
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/kseq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o kt -- python3 $R/bench.py --precision bf16 --no-cpu-baseline --no-other-precisions --no-kernel-timing --steps 2 --warmup 1 > $OUT/bench.log 2>&1
cd $R
python3 - "$OUT/kt_results.db" <<'PY' > gpurun_out/kernel_sequence.txt
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
names = [re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:60] for n, *_ in rows]
# last step = from the last spatial_fwd2 launch on
last = max(i for i, n in enumerate(names) if n.startswith("spatial_fwd2") or "spatial_prep" in n and False)
for i in range(last - 3, len(names)):
    n, s, e, gx, wx = rows[i]
    gap = (s - rows[i - 1][2]) / 1000.0 if i else 0.0
    print(f"{i - last:4d} {names[i]:60s} {(e - s) / 1000.0:8.2f} us  gap {gap:6.2f} us  grid {gx // max(wx, 1)}")
PY
rm -rf $OUT
