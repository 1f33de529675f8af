#!/bin/bash
# A/B of the bf16 LDS-DMA GEMM geometries on the step's own shapes and epilogues (TECM_BF16_DMA = 1: 256x256, one block
# per CU; 2: 256x128x32, two blocks of 8 waves per CU).  Run on the GPU box.
cd $GRAFT_REPO_ROOT
for sel in ${SELS:-1 2}; do
  echo "== TECM_BF16_DMA=$sel"
  export TECM_BF16_DMA=$sel BF16=1
  RES16=abc EPI=bias,gelu,preact SHAPES="69864,3072,768,nk" python3 tools/gemm_shape.py
  RES16=abc EPI=dact SHAPES="69864,3072,768,nk" python3 tools/gemm_shape.py
  RES16=ab EPI=bias,drop,resid SHAPES="69864,768,3072,nk;69864,768,768,nk" python3 tools/gemm_shape.py
  RES16=ab EPI=bias SHAPES="69864,2304,800,nk" python3 tools/gemm_shape.py
  RES16=ab SHAPES="69864,800,2304,nk;69864,768,3072,nk;69864,768,768,nk;8192,8192,8192,nk" python3 tools/gemm_shape.py
done 2>&1 | grep -v amdgpu.ids
