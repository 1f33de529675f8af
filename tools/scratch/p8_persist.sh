#!/bin/bash
# does de-phasing the CUs (first occupants start late) shorten the p8 launches?  TECM_P8_PERSIST = grid size of the persistent launch
export BF16=1
S1="69864,3072,768,nk"; S2="69864,768,768,nk"; S3="69864,768,3072,nk"
for st in 0 256 512; do
  export TECM_P8_PERSIST=$st
  echo "#### stagger $st"
  RES16=abc EPI="" SHAPES="$S1;$S2" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  RES16=abcp EPI="bias,gelu,preact" SHAPES="$S1" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  RES16=abcp EPI="dact,gelu" SHAPES="$S1" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  RES16=ab EPI="bias,resid,drop" SHAPES="$S2;$S3" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
done
