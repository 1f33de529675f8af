#!/bin/bash
# chunks of graphs per tile in spatial_bwd2 (TECM_SPB2_NCH) at B = 2 and B = 8: per-block flush cost against blocks in flight
for B in 2 8; do
  for n in 0 4 6 8 10 13 16 19 26 38; do
    if [ $n = 0 ]; then unset TECM_SPB2_NCH; else export TECM_SPB2_NCH=$n; fi
    echo -n "B=$B nch=${n}: "; BATCH=$B python tools/spatial_bench.py 2>&1 | grep "spatial fwd"
  done
done
