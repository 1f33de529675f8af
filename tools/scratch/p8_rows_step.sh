#!/bin/bash
# step-level A/B of the short tiles: bf16 step at B = 8 and B = 2, launcher's choice vs 256-row tiles pinned
b() { python bench.py --precision bf16 --no-cpu-baseline --no-other-precisions --no-kernel-timing --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], "ms", j["value"], "samples/s")'; }
for rep in 1 2; do
  for B in 8 2; do
    unset TECM_P8_ROWS; echo -n "B=$B auto: "; b --batch $B
    export TECM_P8_ROWS=128; echo -n "B=$B 128 : "; b --batch $B
  done
done
