// Dense projection GEMMs of the SONAR text encoder/decoder and conformer layers
// (q/k/v/out projections, FFNs, pointwise convolutions: reference wiring at
// sonar/models/sonar_text/factory.py:130-153 and, for the conformer, fairseq2's
// ConformerBlock built by sonar/models/sonar_speech/factory.py:64-71), fp16 in, fp32
// accumulate on v_mfma_f32_32x32x16_f16, with the epilogues fused.
#include "gemm_tile.hpp"
#include "gemm_tile256.hpp"
#include "kernels.hpp"

namespace smi {

// EPI_BIAS_F16      : out_h[m][n]  = f16(acc + bias[n])
// EPI_RELU_F16      : out_h[m][n]  = f16(max(acc + bias[n], 0))
// EPI_RESID_F32     : resid[m][n] += acc + bias[n]            (fp32 residual stream)
// EPI_STORE_F32     : out_f[m][n]  = acc + bias[n]            (fp32 logits / split-K slabs)
// EPI_RESID_HALF_F32: resid[m][n] += 0.5 * (acc + bias[n])    (macaron half-step FFN)
// EPI_SILU_F16      : out_h[m][n]  = f16(silu(acc + bias[n]))
// EPI_GLU_F16       : out_h[m][g*32+c] = f16(a * sigmoid(b)), a/b = columns g*64+c / g*64+32+c
//                     (W rows interleaved in 32-channel groups at pack time), out width N/2
// bias may be null for every epilogue.
__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + __expf(-v)); }

template <int EPI>
__device__ __forceinline__ f32x4 epi_act(f32x4 v) {
  if constexpr (EPI == EPI_RELU_F16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if constexpr (EPI == EPI_SILU_F16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
  }
  return v;
}

// LAYOUT: 0 = row-major operands and output; 1 = tile-major X and W (common.hpp), row-major
// output; 2 = tile-major X, W and fp16 output (the output is the next GEMM's X, its K = N).
template <int EPI, int LAYOUT = 0>
__global__ __launch_bounds__(GT_THREADS, 2) void gemm_tn_kernel(const f16* __restrict__ X,
                                                                const f16* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                void* __restrict__ out, int M, int N,
                                                                int K, int ldo, int ksplit,
                                                                size_t part_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile_m, tile_n;
  gt_tile_coords(M / GT_BM, N / GT_BN, tile_m, tile_n);
  const int m0 = tile_m * GT_BM, n0 = tile_n * GT_BN;

  GemmTileAcc acc;
  // split-K (EPI_STORE_F32 only): blockIdx.y owns K/ksplit columns and its own output slab;
  // the consumer sums the slabs (decode-time GEMMs have too few tiles to fill 256 CUs otherwise)
  const int kz = blockIdx.y;
  const int klen = K / ksplit;
  gt_mainloop<(LAYOUT > 0)>(acc, X, W, K, m0, n0, smem, kz * klen, klen);
  if (kz > 0) {
    bias = nullptr;
    out = (char*)out + (size_t)kz * part_stride;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, wn = wave & 1;

  if constexpr (EPI == EPI_GLU_F16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 ba = {0.f, 0.f, 0.f, 0.f}, bg = ba;
      if (bias) {
        ba = *(const f32x4*)(bias + gt_col(n0, 0, q));
        bg = *(const f32x4*)(bias + gt_col(n0, 1, q));
      }
      const int oc = n0 / 2 + wn * 32 + 8 * q + 4 * hi;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int m = gt_row(m0, mi);
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          h[e] = (f16)((acc.v[0][mi][q * 4 + e] + ba[e]) * sigmoid_f(acc.v[1][mi][q * 4 + e] + bg[e]));
        *(half4*)((f16*)out + (size_t)m * ldo + oc) = h;
      }
    }
  } else {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = gt_col(n0, ni, q);
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (bias) b = *(const f32x4*)(bias + n);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int m = gt_row(m0, mi);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc.v[ni][mi][q * 4 + e] + b[e];
          if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_RESID_HALF_F32) {
            float* p = (float*)out + (size_t)m * ldo + n;
            f32x4 o = *(f32x4*)p;
            if constexpr (EPI == EPI_RESID_HALF_F32) v = v * 0.5f;
            *(f32x4*)p = o + v;
          } else if constexpr (EPI == EPI_STORE_F32) {
            *(f32x4*)((float*)out + (size_t)m * ldo + n) = v;
          } else {
            v = epi_act<EPI>(v);
            half4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (f16)v[e];
            if constexpr (LAYOUT == 2)
              *(half4*)((f16*)out + tm_offset(m, n, N)) = h;
            else
              *(half4*)((f16*)out + (size_t)m * ldo + n) = h;
          }
        }
      }
    }
  }
}

// Same epilogues on the 256x256 ping-pong tile engine (gemm_tile256.hpp).
template <int EPI, int LAYOUT = 0>
__global__ __launch_bounds__(G2_THREADS) void gemm_tn256_kernel(const f16* __restrict__ X,
                                                                const f16* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                void* __restrict__ out, int M, int N,
                                                                int K, int ldo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile_m, tile_n;
  g2_tile_coords(M / G2_BM, N / G2_BN, tile_m, tile_n);
  const int m0 = tile_m * G2_BM, n0 = tile_n * G2_BN;

  GemmTile256Acc acc;
  g2_mainloop<(LAYOUT > 0), (LAYOUT > 0)>(acc, X, W, K, m0, n0, smem);

  // ---- epilogue: stage the C tile through LDS (free after the main loop) so the
  // global stores are whole row segments instead of 8-B pieces 32 rows apart.
  // Row stride 528 B: 16-B aligned and 2-way-or-better on the LDS banks.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5, wr = wave >> 2, wc = wave & 3;
  constexpr int CS = G2_CSTRIDE;
  if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32 || EPI == EPI_RESID_HALF_F32) {
    // fp32 outputs, two passes of 128 columns (pass p = the waves' ni block)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (p) __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (bias) b = *(const f32x4*)(bias + g2_col(n0, p, q));
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc.v[p][mi][q * 4 + e] + b[e];
          if constexpr (EPI == EPI_RESID_HALF_F32) v = v * 0.5f;
          *(f32x4*)(smem + (wr * 128 + mi * 32 + l31) * CS + (wc * 32 + 8 * q + 4 * hi) * 4) = v;
        }
      }
      __syncthreads();
      const int c = lane & 31;
      const int gcol = n0 + (c >> 3) * 64 + p * 32 + (c & 7) * 4;
      f32x4 old[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = wave * 32 + it * 2 + hi;
        if constexpr (EPI == EPI_STORE_F32)
          old[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        else
          old[it] = *(const f32x4*)((const float*)out + (size_t)(m0 + row) * ldo + gcol);
      }
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = wave * 32 + it * 2 + hi;
        const f32x4 v = *(const f32x4*)(smem + row * CS + c * 16);
        *(f32x4*)((float*)out + (size_t)(m0 + row) * ldo + gcol) = old[it] + v;
      }
    }
  } else if constexpr (EPI == EPI_GLU_F16) {
    // 128 output channels per tile: channel wc*32 + 8q + 4hi + e from the wave's (a, gate) blocks
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 ba = {0.f, 0.f, 0.f, 0.f}, bg = ba;
      if (bias) {
        ba = *(const f32x4*)(bias + g2_col(n0, 0, q));
        bg = *(const f32x4*)(bias + g2_col(n0, 1, q));
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          h[e] = (f16)((acc.v[0][mi][q * 4 + e] + ba[e]) * sigmoid_f(acc.v[1][mi][q * 4 + e] + bg[e]));
        *(half4*)(smem + (wr * 128 + mi * 32 + l31) * CS + (wc * 32 + 8 * q + 4 * hi) * 2) = h;
      }
    }
    __syncthreads();
    const int c = lane & 15;  // 16 lanes x 16 B = one 256-B output row
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = wave * 32 + it * 4 + (lane >> 4);
      const f32x4 v = *(const f32x4*)(smem + row * CS + c * 16);
      *(f32x4*)((f16*)out + (size_t)(m0 + row) * ldo + n0 / 2 + c * 8) = v;
    }
  } else {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (bias) b = *(const f32x4*)(bias + g2_col(n0, ni, q));
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc.v[ni][mi][q * 4 + e] + b[e];
          v = epi_act<EPI>(v);
          half4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (f16)v[e];
          *(half4*)(smem + (wr * 128 + mi * 32 + l31) * CS + (wc * 64 + ni * 32 + 8 * q + 4 * hi) * 2) = h;
        }
      }
    }
    __syncthreads();
    if constexpr (LAYOUT == 2) {
      // the tile is 8 tile-major blocks (k-blocks n0/32 .. n0/32+7 of row block m0/256) of 16 KiB;
      // every wave instruction stores 1 KiB (16 rows x 64 B) linearly
      f16* blk0 = (f16*)out + ((size_t)(m0 >> 8) * (N >> 5) + (n0 >> 5)) * TM_BLOCK;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int piece = wave * 16 + it;
        const int j = piece >> 4, i = piece & 15;
        const int row = i * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        const f32x4 v = *(const f32x4*)(smem + row * CS + (j * 32 + chunk * 8) * 2);
        *(f32x4*)(blk0 + (size_t)j * TM_BLOCK + i * 512 + lane * 8) = v;
      }
    } else {
      const int c = lane & 31;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = wave * 32 + it * 2 + hi;
        const f32x4 v = *(const f32x4*)(smem + row * CS + c * 16);
        *(f32x4*)((f16*)out + (size_t)(m0 + row) * ldo + n0 + c * 8) = v;
      }
    }
  }
}

template <int EPI, int LAYOUT>
static hipError_t launch_one256(const f16* X, const f16* W, const float* bias, void* out, int M,
                                int N, int K, int ldo, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256_kernel<EPI, LAYOUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G2_KERNEL_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int grid = (M / G2_BM) * (N / G2_BN);
  hipLaunchKernelGGL((gemm_tn256_kernel<EPI, LAYOUT>), dim3(grid), dim3(G2_THREADS), G2_KERNEL_LDS_BYTES,
                     stream, X, W, bias, out, M, N, K, ldo);
  return hipGetLastError();
}

template <int EPI, int LAYOUT = 0>
static hipError_t launch_one(const f16* X, const f16* W, const float* bias, void* out, int M, int N,
                             int K, int ldo, hipStream_t stream, int ksplit = 1, size_t part_stride = 0) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<EPI, LAYOUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int grid = (M / GT_BM) * (N / GT_BN);
  hipLaunchKernelGGL((gemm_tn_kernel<EPI, LAYOUT>), dim3(grid, ksplit), dim3(GT_THREADS), GT_LDS_BYTES,
                     stream, X, W, bias, out, M, N, K, ldo, ksplit, part_stride);
  return hipGetLastError();
}

hipError_t launch_gemm_tn(int epi_sel, const f16* X, const f16* W, const float* bias, void* out,
                          int M, int N, int K, int ldo, hipStream_t stream) {
  // epi_sel = epilogue | (engine << 8) | layout flags: engine 0 auto, 1 force 128x128, 2 force
  // 256x256; GEMM_IN_TM = X and W tile-major, GEMM_OUT_TM = fp16 output tile-major (needs IN_TM)
  const int epi = epi_sel & 0xff, sel = (epi_sel >> 8) & 0xf;
  const bool in_tm = epi_sel & GEMM_IN_TM, out_tm = epi_sel & GEMM_OUT_TM;
  if (M % GT_BM || N % GT_BN || K % GT_BK || M <= 0) return hipErrorInvalidValue;
  if (in_tm && (M % TM_ROWS || N % TM_ROWS)) return hipErrorInvalidValue;
  if (out_tm && (!in_tm || ldo != N)) return hipErrorInvalidValue;
  const bool can256 = M % G2_BM == 0 && N % G2_BN == 0;
  if (sel == 2 && !can256) return hipErrorInvalidValue;
  // the 256x256 engine runs one workgroup per CU: it needs a grid that fills the 256 CUs,
  // otherwise the 128x128 engine (4x the workgroups) wins (decode-time GEMMs, M ~ 1k rows)
  const bool use256 = sel == 2 || (sel == 0 && can256 && (int64_t)(M / G2_BM) * (N / G2_BN) >= 192);
#define SMI_EPI_CASE(E, L)                                                     \
  case E:                                                                      \
    return use256 ? launch_one256<E, L>(X, W, bias, out, M, N, K, ldo, stream) \
                  : launch_one<E, L>(X, W, bias, out, M, N, K, ldo, stream);
  if (out_tm) {  // fp16 outputs that feed the next GEMM
    switch (epi) {
      SMI_EPI_CASE(EPI_BIAS_F16, 2)
      SMI_EPI_CASE(EPI_RELU_F16, 2)
    }
    return hipErrorInvalidValue;
  }
  if (in_tm) {
    switch (epi) {
      SMI_EPI_CASE(EPI_BIAS_F16, 1)
      SMI_EPI_CASE(EPI_RESID_F32, 1)
      SMI_EPI_CASE(EPI_STORE_F32, 1)
    }
    return hipErrorInvalidValue;
  }
  switch (epi) {
    SMI_EPI_CASE(EPI_BIAS_F16, 0)
    SMI_EPI_CASE(EPI_RELU_F16, 0)
    SMI_EPI_CASE(EPI_RESID_F32, 0)
    SMI_EPI_CASE(EPI_STORE_F32, 0)
    SMI_EPI_CASE(EPI_RESID_HALF_F32, 0)
    SMI_EPI_CASE(EPI_SILU_F16, 0)
    SMI_EPI_CASE(EPI_GLU_F16, 0)
  }
#undef SMI_EPI_CASE
  return hipErrorInvalidValue;
}

// Split-K GEMM into `ksplit` fp32 slabs: parts[z][m][n] = X[:, Kz] . W[:, Kz]^T (+ bias for z = 0);
// the consumer (launch_sum_layernorm) adds the slabs to the residual stream.
hipError_t launch_gemm_tn_splitk(const f16* X, const f16* W, const float* bias, float* parts, int M,
                                 int N, int K, int ksplit, hipStream_t stream) {
  if (M % GT_BM || N % GT_BN || ksplit < 1 || K % (GT_BK * ksplit) || M <= 0) return hipErrorInvalidValue;
  return launch_one<EPI_STORE_F32>(X, W, bias, parts, M, N, K, N, stream, ksplit, (size_t)M * N * 4);
}

}  // namespace smi
