#!/bin/bash
# On the GPU box: LDS / MFMA / wait counters of ONE bf16 GEMM shape through tools/gemm_shape.py (its own --pmc passes).
#   SHAPES="8192,8192,8192,nk" TECM_BF16_DMA=1 bash tools/pmc_gemm.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcg; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BF16=1 RES16=${RES16:-ab}
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag -o p -- python3 $R/tools/gemm_shape.py > $OUT/$tag.log 2>&1
done
cd $R
python3 - $OUT <<'PY' | tee gpurun_out/pmc_gemm_${TECM_BF16_DMA:-d}.txt
import glob, sqlite3, sys, collections
res = collections.defaultdict(dict)
for db in glob.glob(sys.argv[1] + "/*/p_results.db"):
    c = sqlite3.connect(db)
    for n, cn, v, k in c.execute("select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name"):
        if "gemm" in n:
            res[n.split("(")[0][-40:]][cn] = v / max(k, 1)
for n, d in res.items():
    print(n)
    for k, v in sorted(d.items()):
        print(f"   {k:32s} {v:16.0f}")
PY
rm -rf $OUT
