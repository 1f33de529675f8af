#!/bin/bash
# eight-phase GEMM: non-temporal (shipped) vs plain write-back stores in the epilogue, step's shapes + the bf16 step
export BF16=1
S1="69864,3072,768,nk"; S2="69864,768,768,nk"; S3="69864,768,3072,nk"; S4="69864,2304,800,nk"
for lib in "" tec-mollm_amd/tecmollm/variants/libtecmollm_hip_wb2.so "" tec-mollm_amd/tecmollm/variants/libtecmollm_hip_wb2.so; do
  if [ -n "$lib" ]; then export TECM_LIB=$PWD/$lib; else unset TECM_LIB; fi
  echo "#### ${lib:-shipped (nt stores)}"
  RES16=abc EPI="" SHAPES="$S1;$S2;$S3;$S4" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  RES16=abcp EPI="bias,gelu,preact" SHAPES="$S1" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  RES16=abcp EPI="dact,gelu" SHAPES="$S1" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  RES16=ab EPI="bias,resid,drop" SHAPES="$S2;$S3" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  python bench.py --precision bf16 --no-cpu-baseline --no-other-precisions --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("step", j["ms_per_step"], "ms")'
done
