#!/bin/bash
# On the GPU box: the rocprofv3 passes behind profiles/<tag>_*_<precision>_B8.*  -- each counter set in ITS OWN run
# (--pmc never together with other trace domains), as MI355X_MICROARCH.md prescribes.
#   bash tools/capture_profiles.sh fp32 r02        (then, anywhere:  python tools/make_profiles.py gpurun_out/prof_r02_fp32 r02 fp32)
set -e
PREC=${1:-fp32}; TAG=${2:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${TAG}_${PREC}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --precision $PREC --no-cpu-baseline --no-other-precisions"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o p -- $BENCH --steps 3 --warmup 1 > $OUT/stats.log 2>&1
echo "stats done"; grep -o '"value": [0-9.]*' $OUT/stats.log | head -1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p -- $BENCH --steps 1 --warmup 1 --no-kernel-timing > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p -- $BENCH --steps 1 --warmup 1 --no-kernel-timing > $OUT/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o p -- $BENCH --steps 1 --warmup 1 --no-kernel-timing > $OUT/mfma.log 2>&1
echo "mfma done"
ls -la $OUT/*/
