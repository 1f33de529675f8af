#!/bin/bash
# On the GPU box: one short rocprofv3 kernel trace of bench.py and a per-kernel / per-grid table of average durations
# for kernels whose name matches the pattern (read straight from the rocpd database, no --stats post-processing).
#   bash tools/kernel_times.sh bf16 'colsum|splitk_reduce'
PREC=${1:-fp32}; PAT=${2:-.}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/kt; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# BATCH=2 bash tools/kernel_times.sh bf16 .   : the same at another batch size (the reference's default is 2 per GPU)
STEPS=${STEPS:-3}
rocprofv3 --kernel-trace -d $OUT -o kt -- python3 $R/bench.py --precision $PREC --no-cpu-baseline --no-other-precisions --no-kernel-timing --steps $STEPS --warmup 1 ${BATCH:+--batch $BATCH} > $OUT/bench.log 2>&1
cd $R
python3 - "$OUT/kt_results.db" "$PAT" "$OUT/bench.log" "$STEPS" <<'PY' | tee gpurun_out/kernel_times_${PREC}${BATCH:+_B$BATCH}.txt
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1]); pat = re.compile(sys.argv[2])
rows = c.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 "
                 "from kernels group by name, grid_x, grid_y order by name").fetchall()
for n, gx, gy, wx, cnt, avg, tot in rows:
    short = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
    if pat.search(short):
        print(f"{short[:48]:48s} grid {gx // wx:6d} x {gy:3d}  wg {wx:4d}  n {cnt:4d}  avg {avg:8.2f} us  total {tot / 1000:7.3f} ms")
# kernel-time sum vs wall time per step: is the step launch-bound?
import json
steps = int(sys.argv[4]) + 1
# (the trace also holds the set-up kernels -- parameter initialisation, graph upload, the first allocation fills --, so the
#  "idle share" below is an UPPER bound on the step's own: kernel time is over-counted, never under-counted)
ksum = c.execute("select sum(end-start)/1e6, count(*), (max(end)-min(start))/1e6 from kernels").fetchone()
line = [l for l in open(sys.argv[3]) if l.startswith("{")]
wall = json.loads(line[-1])["ms_per_step"] if line else float("nan")
print(f"SUMMARY kernel-time sum {ksum[0] / steps:.3f} ms/step over {ksum[1] / steps:.0f} launches/step; bench wall {wall:.3f} ms/step "
      f"(under the profiler); GPU idle share <= {1 - ksum[0] / steps / wall:.1%} (set-up kernels included in the sum)")
PY
cp $OUT/bench.log gpurun_out/kernel_times_bench.log; rm -rf $OUT
