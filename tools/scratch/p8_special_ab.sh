#!/bin/bash
# same-box A/B of the branch-free epilogue forms of the eight-phase GEMM: bf16 step, shipped vs -DP8_SPECIAL=false
b() { python bench.py --precision bf16 --no-cpu-baseline --no-other-precisions --no-kernel-timing --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], "ms", j["value"], "samples/s")'; }
for rep in 1 2 3; do
  unset TECM_LIB; echo -n "shipped   : "; b
  export TECM_LIB=$PWD/tec-mollm_amd/tecmollm/variants/libtecmollm_hip_nospecial.so; echo -n "nospecial : "; b
done
unset TECM_LIB; echo -n "shipped B=2: "; b --batch 2
export TECM_LIB=$PWD/tec-mollm_amd/tecmollm/variants/libtecmollm_hip_nospecial.so; echo -n "nospecial B=2: "; b --batch 2
