#!/bin/bash
# per-epilogue-feature timing of the p8 geometry at the step's shapes (on the GPU box)
export BF16=1
S1="69864,3072,768,nk"; S2="69864,768,768,nk"; S3="69864,768,3072,nk"
run() { echo "== RES16=$1 EPI=$2"; RES16=$1 EPI=$2 SHAPES="$3" python tools/gemm_shape.py 2>&1 | grep -v amdgpu.ids; }
run ab "" "$S1;$S2"
run abc "" "$S1;$S2"
run abcp "bias,gelu,preact" "$S1"
run abcp "dact,gelu" "$S1"
run ab "bias,resid,drop" "$S2;$S3"
