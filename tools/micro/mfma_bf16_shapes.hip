// Bare bf16 MFMA loops on random operands held in registers: v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 at
// equal FLOP per wave, one and two waves per SIMD, 8 independent 32x32 (32 independent 16x16) accumulator tiles per wave
// (the accumulator footprint of the LDS-DMA GEMM's wave tile).  Question: what does the chip sustain on the matrix pipe
// alone -- the ceiling the GEMM's K loop (1.05-1.08 PFLOP/s at 8192^3) should be priced against on random data.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16 tools/micro/mfma_bf16_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512) void loop(const float* in, float* out, int iters) {
  bf16x8 a[6], b[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[i][e] = (__bf16)in[(threadIdx.x * 48 + i * 8 + e) & 4095];
      b[i][e] = (__bf16)in[(threadIdx.x * 48 + i * 8 + e + 1777) & 4095];
    }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 8; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + (t >> 1)) % 6], b[(k + (t & 1)) % 6], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[t][e];
  } else {
    f32x4 acc[32];
#pragma unroll
    for (int t = 0; t < 32; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int t = 0; t < 32; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(k + (t >> 2)) % 6], b[(k + (t & 3)) % 6], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 32; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += acc[t][e];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<float> h(4096);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 32768.0f - 1.0f; }
  float *din, *dout;
  hipMalloc(&din, 4096 * 4); hipMalloc(&dout, 1024 * 512 * 4);
  for (int zero : {0, 1}) {
    if (zero) for (auto& v : h) v = 0.f;
    hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int waves : {4, 8}) {
      for (int shape : {32, 16}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = 256;
        hipEventRecord(e0);
        for (int rep = 0; rep < 20; ++rep) {
          if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(64 * waves), 0, 0, din, dout, iters);
          else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(64 * waves), 0, 0, din, dout, iters);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per iteration per wave: 32 MFMAs of 32x32x16 (32768 flop) = 64 MFMAs of 16x16x32 (16384 flop) = 1048576 flop
        const double flop = 20.0 * blocks * waves * (double)iters * 1048576.0;
        printf("%s operands  waves/CU %d  mfma %s : %8.2f ms  %7.1f TFLOP/s\n", zero ? "zero  " : "random", waves,
               shape == 32 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9);
      }
    }
  }
  return 0;
}
