#!/bin/bash
# the patch projection's two generic-epilogue launches after the fast forms 5 / 6: kernel times from a traced bf16 / fp32 step
for p in bf16 fp32; do
  STEPS=3 bash tools/kernel_times.sh $p 'gemm_bf16_dma_kernel|gemm_kernel<0,0,4,4,128,true|gemm_kernel<0,1,4,4,128,false,false,128>|SUMMARY' | cut -c1-150
done
