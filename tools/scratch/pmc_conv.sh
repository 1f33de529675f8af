#!/bin/bash
# SQ counters of conv_fwd_seq (bf16 y) at the two conv blocks: where do its waves spend their cycles?
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export Y16=1
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag -o p -- python3 $R/tools/conv_fwd_bench.py > $OUT/$tag.log 2>&1
done
cd $R
python3 - $OUT <<'PY' | tee gpurun_out/pmc_conv.txt
import glob, sqlite3, sys, collections
res = collections.defaultdict(dict)
for db in glob.glob(sys.argv[1] + "/*/p_results.db"):
    c = sqlite3.connect(db)
    try:
        q = ("select k.name, k.grid_x, p.counter_name, sum(p.counter_value), count(distinct p.dispatch_id) from pmc_events p join kernels k "
             "on p.dispatch_id = k.dispatch_id group by k.name, k.grid_x, p.counter_name")
        rows = c.execute(q).fetchall()
    except Exception:
        rows = [(n, 0, cn, v, k) for n, cn, v, k in c.execute("select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name")]
    for n, gx, cn, v, k in rows:
        if "conv_fwd_seq_kernel" in n: res[(n.split("(")[0][-32:], gx)][cn] = v / max(k, 1)
for n, d in res.items():
    print(n)
    for k, v in sorted(d.items()):
        print(f"   {k:32s} {v:18.0f}")
PY
rm -rf $OUT
