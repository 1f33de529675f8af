#!/bin/bash
# tile order of the eight-phase GEMM: GROUP_M (m-tiles per L2 super-tile) against time and L2-miss traffic, step's shapes
R=$GRAFT_REPO_ROOT
export BF16=1
S="69864,3072,768,nk;69864,2304,800,nk;69864,800,2304,nk;69864,768,3072,nk;69864,768,768,nk"
cd /tmp && export TMPDIR=/tmp
for gm in 1 2 4 8 16 64; do
  export GROUP_M=$gm
  echo "#### GROUP_M $gm  (bf16 C, no epilogue stream | c_fc form on the first shape)"
  RES16=abc EPI="" SHAPES="$S" python3 $R/tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  RES16=abcp EPI="bias,gelu,preact" SHAPES="69864,3072,768,nk" python3 $R/tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  OUT=$R/gpurun_out/gmp_$gm; rm -rf $OUT
  RES16=abc EPI="" SHAPES="$S" rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o p -- python3 $R/tools/gemm_shape.py > /dev/null 2>&1
  python3 - $OUT/p_results.db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, grid_x, sum(counter_value), count(distinct dispatch_id) from pmc_events where counter_name='FETCH_SIZE' group by name, grid_x").fetchall() if False else None
try:
    q = ("select k.name, k.grid_x, sum(p.counter_value), count(distinct p.dispatch_id) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
         "where p.counter_name='FETCH_SIZE' group by k.name, k.grid_x")
    rows = c.execute(q).fetchall()
except Exception as e:
    rows = c.execute("select name, 0, sum(counter_value), count(distinct dispatch_id) from pmc_events where counter_name='FETCH_SIZE' group by name").fetchall()
for n, gx, v, k in rows:
    if "p8" in n: print(f"   fetch (x2 corrected) {n.split('(')[0][-28:]} grid {gx:7d}: {2 * v * 1024 / k / 1e6:8.1f} MB/launch over {k} launches")
PY
  rm -rf $OUT
done
