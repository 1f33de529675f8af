#!/bin/bash
# On the GPU box: one short rocprofv3 kernel trace of bench.py and a per-kernel / per-grid table of average durations
# for kernels whose name matches the pattern (read straight from the rocpd database, no --stats post-processing).
#   bash tools/kernel_times.sh bf16 'colsum|splitk_reduce'
PREC=${1:-fp32}; PAT=${2:-.}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/kt; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o kt -- python3 $R/bench.py --precision $PREC --no-cpu-baseline --no-other-precisions --steps 3 --warmup 1 > $OUT/bench.log 2>&1
cd $R
python3 - "$OUT/kt_results.db" "$PAT" <<'PY' | tee gpurun_out/kernel_times_$PREC.txt
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1]); pat = re.compile(sys.argv[2])
rows = c.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 "
                 "from kernels group by name, grid_x, grid_y order by name").fetchall()
for n, gx, gy, wx, cnt, avg, tot in rows:
    short = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
    if pat.search(short):
        print(f"{short[:48]:48s} grid {gx // wx:6d} x {gy:3d}  wg {wx:4d}  n {cnt:4d}  avg {avg:8.2f} us  total {tot / 1000:7.3f} ms")
PY
cp $OUT/bench.log gpurun_out/kernel_times_bench.log; rm -rf $OUT
