#!/bin/bash
# A/B of bf16 LDS-DMA GEMM geometries with the step's own residencies (bf16 pre-activation as the step stores it).
# SELS="5 6 7" GMS="0 2 4" bash tools/dma_ab2.sh        (run on the GPU box; GMS = GROUP_M values, GROUPS is a bash special)
cd $GRAFT_REPO_ROOT
for sel in ${SELS:-5 6}; do
 for gm in ${GMS:-0}; do
  echo "== TECM_BF16_DMA=$sel GROUP_M=$gm"
  export TECM_BF16_DMA=$sel BF16=1 GROUP_M=$gm
  RES16=abcp EPI=bias,gelu,preact SHAPES="69864,3072,768,nk" python3 tools/gemm_shape.py
  RES16=abcp EPI=dact SHAPES="69864,3072,768,nk" python3 tools/gemm_shape.py
  RES16=ab EPI=bias,drop,resid SHAPES="69864,768,3072,nk;69864,768,768,nk" python3 tools/gemm_shape.py
  RES16=abc EPI=bias SHAPES="69864,2304,800,nk" python3 tools/gemm_shape.py
  RES16=ab SHAPES="69864,768,3072,nk;69864,768,768,nk;8192,8192,8192,nk" python3 tools/gemm_shape.py
 done
done 2>&1 | grep -v amdgpu.ids
