#!/bin/bash
# On the GPU box: capture_profiles.sh for one precision, turn the rocpd databases into the committed summaries there
# (they are ~45 MB per precision, gpurun brings back at most 64 MiB) and leave only the summaries under gpurun_out/.
#   bash tools/capture_and_summarise.sh fp32 r03
set -e
PREC=${1:-fp32}; TAG=${2:-r03}
R=$GRAFT_REPO_ROOT
bash $R/tools/capture_profiles.sh $PREC $TAG > $R/gpurun_out/cap_${TAG}_${PREC}.log 2>&1
cd $R
python3 tools/make_profiles.py gpurun_out/prof_${TAG}_${PREC} $TAG $PREC
mkdir -p gpurun_out/${TAG}_profiles
cp profiles/${TAG}_*_${PREC}_B8.* gpurun_out/${TAG}_profiles/
cp gpurun_out/prof_${TAG}_${PREC}/stats.log gpurun_out/${TAG}_profiles/stats_${PREC}.log
rm -rf gpurun_out/prof_${TAG}_${PREC}
ls -la gpurun_out/${TAG}_profiles
