// Microbenchmark: does the operand-reuse pattern of back-to-back v_mfma_f32_16x16x32_f16 change the
// sustained (power-limited) rate on random data?  32 independent accumulators, 4 A and 8 B fragments:
//   order 0: A outer (A reused by 8 consecutive MFMAs), order 1: B outer (B reused by 4), order 2: interleaved.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ORDER>
__global__ __launch_bounds__(512) void k(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  half8 a[4], b[8];
  for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + 512 * i) % 8192];
  for (int i = 0; i < 8; ++i) b[i] = src[(threadIdx.x + 512 * (i + 4)) % 8192];
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if constexpr (ORDER == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else if constexpr (ORDER == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        const int i = t & 3, j = (t * 5) & 7;
        acc[i][(j + (t >> 3)) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[(j + (t >> 3)) & 7], acc[i][(j + (t >> 3)) & 7], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float sum = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j)
      for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main() {
  half8* src;
  float* out;
  hipMalloc(&src, 8192 * sizeof(half8));
  hipMalloc(&out, 1024 * 512 * 4);
  _Float16* h = (_Float16*)malloc(8192 * 16);
  for (int i = 0; i < 8192 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1);
  hipMemcpy(src, h, 8192 * 16, hipMemcpyHostToDevice);
  const int iters = 8000;
  for (int order = 0; order < 3; ++order) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (order == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(512), 0, 0, src, out, iters);
      if (order == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(512), 0, 0, src, out, iters);
      if (order == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(512), 0, 0, src, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = 1024.0 * 8 * iters * 32 * 16384.0;
    printf("order %d (%s): %.2f ms, %.0f TFLOP/s\n", order,
           order == 0 ? "A outer, reused 8x" : (order == 1 ? "B outer, reused 4x" : "interleaved"), ms, flops / ms / 1e9);
  }
  return 0;
}
