#!/bin/bash
# tile height of the eight-phase GEMM (TECM_P8_ROWS = 128 | 112 | 96, unset = the launcher's choice) at the step's shapes
export BF16=1
S1="69864,3072,768,nk"; S2="69864,768,768,nk"; S3="69864,768,3072,nk"; S4="69864,2304,800,nk"; S5="69864,800,2304,nk"; S6="23288,2304,576,nk"
for r in 128 112 96 auto; do
  if [ $r = auto ]; then unset TECM_P8_ROWS; else export TECM_P8_ROWS=$r; fi
  echo "#### rows $r"
  RES16=abc EPI="" SHAPES="$S1;$S2;$S3;$S4;$S5;$S6" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
  RES16=ab EPI="bias,resid,drop" SHAPES="$S2;$S3" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids
done
