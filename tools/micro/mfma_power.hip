// Microbenchmark: sustained fp16 MFMA rate of v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 on
// random vs zero register operands (data-dependent power).  8 waves/CU, 128 accumulator registers each,
// no memory traffic.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  half8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + 512 * i) % 4096];
  for (int i = 0; i < 2; ++i) b[i] = src[(threadIdx.x + 512 * (i + 4)) % 4096];
  float sum = 0.f;
  if constexpr (KIND == 0) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[i & 1], a[i >> 1], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) sum += acc[i][r];
  } else {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i)
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i & 1], a[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i)
      for (int r = 0; r < 4; ++r) sum += acc[i][r];
  }
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main() {
  half8* src;
  float* out;
  hipMalloc(&src, 4096 * sizeof(half8));
  hipMalloc(&out, 256 * 8 * 512 * 4);
  _Float16* h = (_Float16*)malloc(4096 * 16);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 4096 * 8; ++i) h[i] = mode ? (_Float16)((rand() / (float)RAND_MAX) * 2 - 1) : (_Float16)0.f;
    hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int kind = 0; kind < 2; ++kind) {
      const int iters = kind == 0 ? 20000 : 5000;  // 8 x 32K-flop vs 32 x 16K-flop MFMAs per iteration
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (kind == 0)
          hipLaunchKernelGGL(k<0>, dim3(256 * 4), dim3(512), 0, 0, src, out, iters);
        else
          hipLaunchKernelGGL(k<1>, dim3(256 * 4), dim3(512), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 256.0 * 4 * 8 * (double)iters * (kind == 0 ? 8 * 32768.0 * 2 : 32 * 16384.0 * 2) / 2;
        if (rep == 2)
          printf("%s operands, %s: %.2f ms, %.0f TFLOP/s\n", mode ? "random" : "zero  ",
                 kind == 0 ? "32x32x16" : "16x16x32", ms, flops * 2 / ms / 1e9 / 2);
      }
    }
  }
  return 0;
}
