// Bare f32 MFMA loops on random operands held in registers: v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32 at
// equal FLOP per wave, one and two waves per SIMD.  Question (MI355X_MICROARCH.md, DVFS item 7, measured there for bf16
// only): does the 16x16 shape hold a higher clock under sustained load?   hipcc --offload-arch=gfx950 -O3 -o mfma_f32 ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void loop(const float* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(threadIdx.x * 8 + i) & 4095];
    b[i] = in[(threadIdx.x * 8 + i + 1777) & 4095];
  }
  if constexpr (SHAPE == 32) {
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + t) & 7], b[k], acc[t], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + t) & 7], b[(k + (t >> 2)) & 7], acc[t], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += acc[t][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
  (void)lane;
}

int main() {
  std::vector<float> h(4096);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 32768.0f - 1.0f; }
  float *din, *dout;
  hipMalloc(&din, 4096 * 4); hipMalloc(&dout, 1024 * 512 * 4);
  hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const int iters = 40000;
  for (int waves : {4, 8}) {
    for (int shape : {32, 16, 32, 16}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int blocks = 256;
      hipEventRecord(e0);
      for (int rep = 0; rep < 20; ++rep) {
        if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(64 * waves), 0, 0, din, dout, iters);
        else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(64 * waves), 0, 0, din, dout, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per iteration per wave: 32 MFMAs of 32x32x2 (4096 flop) = 64 MFMAs of 16x16x4 (2048 flop) = 131072 flop
      const double flop = 20.0 * blocks * waves * (double)iters * 131072.0;
      printf("waves/CU %d  mfma %s : %8.2f ms  %7.1f TFLOP/s\n", waves, shape == 32 ? "32x32x2 " : "16x16x4 ", ms, flop / ms / 1e9);
    }
  }
  return 0;
}
