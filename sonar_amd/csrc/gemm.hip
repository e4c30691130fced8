// Dense projection GEMMs of the SONAR text encoder/decoder layers
// (q/k/v/out projections and the FFN: reference wiring at
// sonar/models/sonar_text/factory.py:130-153), fp16 in, fp32 accumulate on
// v_mfma_f32_32x32x16_f16, with the bias / ReLU / residual epilogues fused.
#include "gemm_tile.hpp"
#include "kernels.hpp"

namespace smi {

// EPI_BIAS_F16 : out_h[m][n]  = f16(acc + bias[n])
// EPI_RELU_F16 : out_h[m][n]  = f16(max(acc + bias[n], 0))
// EPI_RESID_F32: resid[m][n] += acc + bias[n]          (fp32 residual stream)
template <int EPI>
__global__ __launch_bounds__(GT_THREADS, 2) void gemm_tn_kernel(const f16* __restrict__ X,
                                                                const f16* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                void* __restrict__ out, int M, int N,
                                                                int K, int ldo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile_m, tile_n;
  gt_tile_coords(M / GT_BM, N / GT_BN, tile_m, tile_n);
  const int m0 = tile_m * GT_BM, n0 = tile_n * GT_BN;

  GemmTileAcc acc;
  gt_mainloop(acc, X, W, K, m0, n0, smem);

#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = gt_col(n0, ni, q);
      const f32x4 b = *(const f32x4*)(bias + n);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int m = gt_row(m0, mi);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc.v[ni][mi][q * 4 + e] + b[e];
        if constexpr (EPI == EPI_RESID_F32) {
          float* p = (float*)out + (size_t)m * ldo + n;
          f32x4 o = *(f32x4*)p;
          *(f32x4*)p = o + v;
        } else {
          if constexpr (EPI == EPI_RELU_F16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          half4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (f16)v[e];
          *(half4*)((f16*)out + (size_t)m * ldo + n) = h;
        }
      }
    }
  }
}

template <int EPI>
static hipError_t launch_one(const f16* X, const f16* W, const float* bias, void* out, int M, int N,
                             int K, int ldo, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int grid = (M / GT_BM) * (N / GT_BN);
  hipLaunchKernelGGL(gemm_tn_kernel<EPI>, dim3(grid), dim3(GT_THREADS), GT_LDS_BYTES, stream, X, W,
                     bias, out, M, N, K, ldo);
  return hipGetLastError();
}

hipError_t launch_gemm_tn(int epi, const f16* X, const f16* W, const float* bias, void* out, int M,
                          int N, int K, int ldo, hipStream_t stream) {
  if (M % GT_BM || N % GT_BN || K % GT_BK || M <= 0) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_BIAS_F16: return launch_one<EPI_BIAS_F16>(X, W, bias, out, M, N, K, ldo, stream);
    case EPI_RELU_F16: return launch_one<EPI_RELU_F16>(X, W, bias, out, M, N, K, ldo, stream);
    case EPI_RESID_F32: return launch_one<EPI_RESID_F32>(X, W, bias, out, M, N, K, ldo, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace smi
