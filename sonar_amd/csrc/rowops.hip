// HBM-bound row kernels of the text encoder: embedding gather + sinusoidal
// positions, LayerNorm, final LayerNorm fused with sentence pooling, dtype
// conversion.  One 64-lane wave owns one 1024-wide row (16 B per lane per
// access, fully coalesced), statistics are reduced with DPP/shuffle only.
//
// Reference semantics:
//  * frontend  : TransformerEmbeddingFrontend wired at
//                sonar/models/sonar_text/factory.py:73-100  (E[tok]*sqrt(d) + PE, fp32 add)
//  * LayerNorm : StandardLayerNorm(d, bias=True), eps 1e-5 (factory.py:117)
//  * pooling   : SonarTextTransformerEncoderModel.static_pooling,
//                sonar/models/sonar_text/model.py:86-128
#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"

namespace smi {

// ------------------------------------------------ tile-major rows, pair map
// The fp16 residual stream may be TILE-MAJOR (common.hpp tm_offset): rows r, r+1 (r even) of a 32-column
// block are 128 contiguous bytes.  The row kernels below therefore walk tile-major x in ROW PAIRS:
//   lane -> block (lane>>3) + 8c, row of the pair (lane>>2)&1, 16-B chunk lane&3
// so that one wave-wide 16-B access covers 8 FULL 128-B lines (the row-major lane map, lane -> columns
// lane*8 of ONE row, touches 16 half lines per access on this layout: 0.65-0.75x the bandwidth, r02e).
// A lane then owns 8 columns of every 256-column group c of its row.
__device__ __forceinline__ int pair_rsel(int lane) { return (lane >> 2) & 1; }
__device__ __forceinline__ int pair_col(int lane, int c) { return (((lane >> 3) + 8 * c) << 5) + ((lane & 3) << 3); }
// sum over the 32 lanes that hold the same row of the pair
__device__ __forceinline__ float pair_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// ---------------------------------------------------------------- embed+pack
// grid (N, ceil(max_len/4)), 256 threads: wave w handles position blockIdx.y*4+w
// of sentence blockIdx.x; rows beyond the sentence length do nothing.
// XT: type of the residual stream x (float, or f16 when the encoder runs with SMI_ENC_FP16_RESIDUAL)
template <typename XT>
__global__ __launch_bounds__(256) void embed_pack_kernel(const int64_t* __restrict__ ids,
                                                         const int32_t* __restrict__ cu,
                                                         const f16* __restrict__ table,
                                                         const float* __restrict__ pos_table,
                                                         float scale, int pos_offset,
                                                         XT* __restrict__ x, int S, int d,
                                                         int64_t vocab, int32_t* __restrict__ bad) {
  const int n = blockIdx.x;
  const int p = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int start = cu[n];
  const int len = cu[n + 1] - start;
  if (p >= len) return;
  int64_t tok = ids[(size_t)n * S + p];
  if (tok < 0 || tok >= vocab) {  // the reference's embedding lookup raises: report, never read out of the table
    if (bad && lane == 0) *bad = 1;
    tok = tok < 0 ? 0 : vocab - 1;
  }
  const f16* e = table + (size_t)tok * d;
  const float* pe = pos_table + (size_t)(p + pos_offset) * d;
  XT* o = x + (size_t)(start + p) * d;
  for (int c = lane * 8; c < d; c += 512) {
    XT* oc = o + c;
    const half8 ev = *(const half8*)(e + c);
    const f32x4 p0 = *(const f32x4*)(pe + c);
    const f32x4 p1 = *(const f32x4*)(pe + c + 4);
    f32x4 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // reference: (embed(seqs) * scale) in model dtype, then fp32 add of PE.  fp16 stream = fp16 model: the product is
      // rounded to fp16; fp32 stream (fp32 and bf16 models): no fp16 rounding -- a bf16 model's scaled embedding may lie
      // beyond fp16's range (round 4)
      if constexpr (sizeof(XT) == 4) {
        o0[i] = (float)ev[i] * scale + p0[i];
        o1[i] = (float)ev[i + 4] * scale + p1[i];
      } else {
        o0[i] = (float)(f16)((float)ev[i] * scale) + p0[i];
        o1[i] = (float)(f16)((float)ev[i + 4] * scale) + p1[i];
      }
    }
    if constexpr (sizeof(XT) == 4) {
      *(f32x4*)(oc) = o0;
      *(f32x4*)(oc + 4) = o1;
    } else {
      half8 h;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        h[i] = (f16)o0[i];
        h[i + 4] = (f16)o1[i];
      }
      *(half8*)(oc) = h;
    }
  }
}

// Tile-major residual stream (d % 256 == 0): wave w handles the PAIR of packed rows (ge + 2k, ge + 2k + 1),
// ge = start & ~1, k = blockIdx.y*4 + w, with the pair map above: table rows are read in 512-B runs, x is
// written in full 128-B lines.  Rows of the pair outside the sentence belong to a neighbour and are left alone.
__global__ __launch_bounds__(256) void embed_pack_tm_kernel(const int64_t* __restrict__ ids,
                                                            const int32_t* __restrict__ cu,
                                                            const f16* __restrict__ table,
                                                            const float* __restrict__ pos_table, float scale,
                                                            int pos_offset, f16* __restrict__ x, int S, int d,
                                                            int64_t vocab, int32_t* __restrict__ bad) {
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int start = cu[n];
  const int len = cu[n + 1] - start;
  const int g = (start & ~1) + 2 * (blockIdx.y * 4 + (threadIdx.x >> 6)) + pair_rsel(lane);
  const int p = g - start;
  if (p < 0 || p >= len) return;
  int64_t tok = ids[(size_t)n * S + p];
  if (tok < 0 || tok >= vocab) {
    if (bad && (lane & ~4) == 0) *bad = 1;
    tok = tok < 0 ? 0 : vocab - 1;
  }
  const f16* e = table + (size_t)tok * d;
  const float* pe = pos_table + (size_t)(p + pos_offset) * d;
  for (int cg = 0; cg < d / 256; ++cg) {
    const int c = pair_col(lane, cg);
    const half8 ev = *(const half8*)(e + c);
    const f32x4 p0 = *(const f32x4*)(pe + c);
    const f32x4 p1 = *(const f32x4*)(pe + c + 4);
    half8 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h[i] = (f16)((float)(f16)((float)ev[i] * scale) + p0[i]);
      h[i + 4] = (f16)((float)(f16)((float)ev[i + 4] * scale) + p1[i]);
    }
    *(half8*)(x + tm_offset(g, c, d)) = h;
  }
}

hipError_t launch_embed_pack(const int64_t* ids, const int32_t* cu, const f16* table,
                             const float* pos_table, float scale, int pos_offset, void* x, int N,
                             int S, int max_len, int d, int64_t vocab, hipStream_t stream, int x_f16,
                             int32_t* bad, int x_tm) {
  if (x_tm && !x_f16) return hipErrorInvalidValue;
  if (d % 8 || N <= 0 || max_len <= 0) return hipErrorInvalidValue;
  dim3 grid(N, (max_len + 3) / 4);
  if (x_tm) {
    if (d % 256) return hipErrorInvalidValue;
    // pairs per sentence: up to max_len/2 + 1 (a sentence starting on an odd row straddles one more pair)
    hipLaunchKernelGGL(embed_pack_tm_kernel, dim3(N, (max_len / 2 + 1 + 3) / 4), dim3(256), 0, stream, ids, cu, table,
                       pos_table, scale, pos_offset, (f16*)x, S, d, vocab, bad);
    return hipGetLastError();
  }
  if (x_f16)
    hipLaunchKernelGGL(embed_pack_kernel<f16>, grid, dim3(256), 0, stream, ids, cu, table, pos_table, scale,
                       pos_offset, (f16*)x, S, d, vocab, bad);
  else
    hipLaunchKernelGGL(embed_pack_kernel<float>, grid, dim3(256), 0, stream, ids, cu, table, pos_table, scale,
                       pos_offset, (float*)x, S, d, vocab, bad);
  return hipGetLastError();
}

// ---------------------------------------------------------------- LayerNorm
// NV = d / 256 float4 vectors per lane; lane l owns columns 256*k + 4*l .. +3.
template <int NV, typename XT = float>
__device__ __forceinline__ void ln_row(const XT* __restrict__ xr, const float* __restrict__ w,
                                       const float* __restrict__ b, float eps, int lane,
                                       f32x4 (&y)[NV]) {
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if constexpr (sizeof(XT) == 4) {
      v[k] = *(const f32x4*)(xr + k * 256 + lane * 4);
    } else {
      const half4 hv = *(const half4*)(xr + k * 256 + lane * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[k][i] = (float)hv[i];
    }
    s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  constexpr float inv_d = 1.0f / (NV * 256);
  const float mean = wave_sum(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float c = v[k][i] - mean;
      v[k][i] = c;
      q += c * c;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const f32x4 wv = *(const f32x4*)(w + k * 256 + lane * 4);
    const f32x4 bv = *(const f32x4*)(b + k * 256 + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[k][i] = v[k][i] * rstd * wv[i] + bv[i];
  }
}

template <int NV, typename XT>
__global__ __launch_bounds__(256) void layernorm_kernel(const XT* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps,
                                                        f16* __restrict__ h, int rows) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  constexpr int D = NV * 256;
  for (int r = blockIdx.x * 4 + wv; r < rows; r += gridDim.x * 4) {
    if constexpr (sizeof(XT) == 2 && NV % 2 == 0) {
      // fp16 stream: 8 consecutive columns per lane and 512-column block -> 16-B loads and stores
      constexpr int NH = NV / 2;
      float v[NH][8];
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const half8 raw = *(const half8*)(x + (size_t)r * D + k * 512 + lane * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[k][i] = (float)raw[i];
          sum += v[k][i];
        }
      }
      constexpr float inv_d = 1.0f / D;
      const float mean = wave_sum(sum) * inv_d;
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < NH; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[k][i] -= mean;
          sq += v[k][i] * v[k][i];
        }
      const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + eps);
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const float* wp = w + k * 512 + lane * 8;
        const float* bp = b + k * 512 + lane * 8;
        const f32x4 w0 = *(const f32x4*)wp, w1 = *(const f32x4*)(wp + 4);
        const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
        half8 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[i] = (f16)(v[k][i] * rstd * w0[i] + b0[i]);
          o[4 + i] = (f16)(v[k][4 + i] * rstd * w1[i] + b1[i]);
        }
        *(half8*)(h + (size_t)r * D + k * 512 + lane * 8) = o;
      }
      continue;
    }
    f32x4 y[NV];
    ln_row<NV, XT>(x + (size_t)r * D, w, b, eps, lane, y);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      half4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (f16)y[k][i];
      *(half4*)(h + (size_t)r * D + k * 256 + lane * 4) = o;
    }
  }
}

// Tile-major output (the GEMM operand layout of common.hpp).  A workgroup normalises 16 consecutive
// rows into an LDS tile and then writes them out per 32-column k-block: 16 rows x 64 B are one
// contiguous 1 KiB run of a tile-major block, i.e. one fully coalesced wave store (a row-at-a-time
// writer would scatter 64-B pieces 16 KiB apart).  h holds rows rounded up to 16 (256 in practice).
template <int NV, typename XT, bool XTM>
__global__ __launch_bounds__(256) void layernorm_tm_kernel(const XT* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ b, float eps,
                                                           f16* __restrict__ h, int rows) {
  constexpr int D = NV * 256;
  constexpr int RS = D * 2 + 16;  // LDS row stride in bytes (+16: the 16 rows of a read hit different banks)
  __shared__ __attribute__((aligned(16))) char tile[16 * RS];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  for (int r0 = blockIdx.x * 16; r0 < rows; r0 += gridDim.x * 16) {
    if constexpr (XTM) {
      // tile-major fp16 x: pair map (top of file).  Wave wv owns rows 4wv..4wv+3 of the group as two row pairs,
      // a lane holds 8 columns of each 256-column group of ITS row of each pair; all loads are in flight
      // before the first reduction.  r0 is a multiple of 16, so the pairs are line aligned.
      static_assert(sizeof(XT) == 2, "tile-major x is fp16");
      const int rsel = pair_rsel(lane);
      half8 raw[2][NV];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = r0 + 4 * wv + 2 * j + rsel;  // < padded rows: x is allocated in whole 256-row panels
#pragma unroll
        for (int k = 0; k < NV; ++k)  // every line of x is read by this workgroup only, once: non-temporal
          raw[j][k] = __builtin_nontemporal_load((const half8*)(x + tm_offset(row, pair_col(lane, k), D)));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = 4 * wv + 2 * j + rsel;
        float v[NV][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            v[k][i] = (float)raw[j][k][i];
            sum += v[k][i];
          }
        constexpr float inv_d = 1.0f / D;
        const float mean = pair_sum(sum) * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            v[k][i] -= mean;
            sq += v[k][i] * v[k][i];
          }
        const float var = pair_sum(sq) * inv_d;
        const bool live = r0 + lr < rows;  // rows past the end of x: exact zeros, whatever the padding of x holds
        const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int c = pair_col(lane, k);
          const f32x4 w0 = *(const f32x4*)(w + c), w1 = *(const f32x4*)(w + c + 4);
          const f32x4 b0 = *(const f32x4*)(b + c), b1 = *(const f32x4*)(b + c + 4);
          half8 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            o[i] = live ? (f16)(v[k][i] * rstd * w0[i] + b0[i]) : (f16)0.f;
            o[4 + i] = live ? (f16)(v[k][4 + i] * rstd * w1[i] + b1[i]) : (f16)0.f;
          }
          *(half8*)(tile + lr * RS + c * 2) = o;
        }
      }
    } else if constexpr (sizeof(XT) == 2 && NV % 2 == 0) {
      // fp16 stream: a lane owns 8 consecutive columns per 512-column block, so every global load and
      // every LDS store moves 16 B per lane (8-B accesses run at 0.54-0.70x the 16-B rate), and the
      // loads of the wave's 4 rows are all in flight before the first reduction
      constexpr int NH = NV / 2;
      half8 raw[4][NH];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = min(r0 + wv + 4 * q, rows - 1);
#pragma unroll
        for (int k = 0; k < NH; ++k)
          raw[q][k] = *(const half8*)(x + (size_t)row * D + k * 512 + lane * 8);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int lr = wv + 4 * q;
        float v[NH][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NH; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            v[k][i] = (float)raw[q][k][i];
            sum += v[k][i];
          }
        constexpr float inv_d = 1.0f / D;
        const float mean = wave_sum(sum) * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < NH; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            v[k][i] -= mean;
            sq += v[k][i] * v[k][i];
          }
        const bool live = r0 + lr < rows;  // rows past the end of x: exact zeros
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + eps);
#pragma unroll
        for (int k = 0; k < NH; ++k) {
          const float* wp = w + k * 512 + lane * 8;
          const float* bp = b + k * 512 + lane * 8;
          const f32x4 w0 = *(const f32x4*)wp, w1 = *(const f32x4*)(wp + 4);
          const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
          half8 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            o[i] = live ? (f16)(v[k][i] * rstd * w0[i] + b0[i]) : (f16)0.f;
            o[4 + i] = live ? (f16)(v[k][4 + i] * rstd * w1[i] + b1[i]) : (f16)0.f;
          }
          *(half8*)(tile + lr * RS + (k * 512 + lane * 8) * 2) = o;
        }
      }
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int lr = wv + 4 * q;
      f32x4 y[NV];
      if (r0 + lr < rows) {
        ln_row<NV, XT>(x + (size_t)(r0 + lr) * D, w, b, eps, lane, y);
      } else {  // rows past the end of x inside the last 16-row group: zeros
#pragma unroll
        for (int k = 0; k < NV; ++k) y[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        half4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (f16)y[k][i];
        *(half4*)(tile + lr * RS + (k * 256 + lane * 4) * 2) = o;
      }
    }
    }
    __syncthreads();
    // wave wv copies k-blocks wv, wv+4, ...: lane -> (row lane>>2, slot lane&3)
    const int lr = lane >> 2, slot = lane & 3;
    const int rr = (r0 & 255) + lr;
    const int chunk = slot ^ tm_swz(rr);
    f16* dst = h + (size_t)(r0 >> 8) * (D >> 5) * TM_BLOCK + rr * 32 + slot * 8;
#pragma unroll
    for (int i = 0; i < D / 128; ++i) {
      const int kb = wv + 4 * i;
      const f32x4 v = *(const f32x4*)(tile + lr * RS + (kb * 32 + chunk * 8) * 2);
      *(f32x4*)(dst + (size_t)kb * TM_BLOCK) = v;
    }
    __syncthreads();
  }
}

hipError_t launch_layernorm(const void* x, const float* w, const float* b, float eps, f16* h,
                            int rows, int d, hipStream_t stream, int out_tm, int x_f16, int x_tm) {
  if (rows <= 0) return hipErrorInvalidValue;
  if (x_tm && !(out_tm && x_f16 && d % 512 == 0)) return hipErrorInvalidValue;  // tile-major x: fp16 stream, tile-major h
  const int blocks = out_tm ? min((rows + 15) / 16, 256 * 16) : min((rows + 3) / 4, 256 * 32);
#define SMI_LN_LAUNCH(NV, XT)                                                                                      \
  if (out_tm && x_tm)                                                                                              \
    hipLaunchKernelGGL((layernorm_tm_kernel<NV, f16, true>), dim3(blocks), dim3(256), 0, stream, (const f16*)x, w, \
                       b, eps, h, rows);                                                                           \
  else if (out_tm)                                                                                                 \
    hipLaunchKernelGGL((layernorm_tm_kernel<NV, XT, false>), dim3(blocks), dim3(256), 0, stream, (const XT*)x, w,  \
                       b, eps, h, rows);                                                                           \
  else                                                                                                             \
    hipLaunchKernelGGL((layernorm_kernel<NV, XT>), dim3(blocks), dim3(256), 0, stream, (const XT*)x, w, b, eps, h, \
                       rows);
#define SMI_LN_CASE(NV)             \
  case NV * 256:                    \
    if (x_f16) {                    \
      SMI_LN_LAUNCH(NV, f16)        \
    } else {                        \
      SMI_LN_LAUNCH(NV, float)      \
    }                               \
    break;
  switch (d) {
    SMI_LN_CASE(1)
    SMI_LN_CASE(2)
    SMI_LN_CASE(3)
    SMI_LN_CASE(4)
    SMI_LN_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef SMI_LN_CASE
#undef SMI_LN_LAUNCH
  return hipGetLastError();
}

// ------------------------------------------------- final LayerNorm + pooling
// One workgroup per sentence; wave w walks rows w, w+4, ...; the pooled vector
// is combined across the 4 waves through LDS.  pooling: 0 mean, 1 max, 2 last.
template <int NV, typename OutT, typename XT>
__global__ __launch_bounds__(256) void ln_pool_kernel(const XT* __restrict__ x,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ b, float eps,
                                                      const int32_t* __restrict__ cu,
                                                      OutT* __restrict__ out,
                                                      OutT* __restrict__ encoded, int S,
                                                      int pooling) {
  constexpr int D = NV * 256;
  __shared__ __attribute__((aligned(16))) float red[4][D];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int start = cu[n];
  const int len = cu[n + 1] - start;

  f32x4 accv[NV];
  const float init = pooling == 1 ? -INFINITY : 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) accv[k] = f32x4{init, init, init, init};

  for (int p = wv; p < len; p += 4) {
    f32x4 y[NV];
    ln_row<NV, XT>(x + (size_t)(start + p) * D, w, b, eps, lane, y);
    if (encoded) {
      OutT* e = encoded + ((size_t)n * S + p) * D;
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) e[k * 256 + lane * 4 + i] = (OutT)y[k][i];
    }
    if (pooling == 0) {
#pragma unroll
      for (int k = 0; k < NV; ++k) accv[k] += y[k];
    } else if (pooling == 1) {
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) accv[k][i] = fmaxf(accv[k][i], y[k][i]);
    } else if (p == len - 1) {
#pragma unroll
      for (int k = 0; k < NV; ++k) accv[k] = y[k];
    }
  }
  if (encoded) {  // zero the padded tail so the padded view is deterministic
    for (int p = len + wv; p < S; p += 4) {
      OutT* e = encoded + ((size_t)n * S + p) * D;
      for (int c = lane; c < D; c += 64) e[c] = (OutT)0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) *(f32x4*)(&red[wv][k * 256 + lane * 4]) = accv[k];
  __syncthreads();
  // reference (model.py:115-124): weights = 1/(seq_len + 1e-7), in fp32 here
  const float wgt = 1.0f / ((float)len + 1e-7f);
  for (int c = threadIdx.x; c < D; c += 256) {
    float v;
    if (pooling == 1)
      v = fmaxf(fmaxf(red[0][c], red[1][c]), fmaxf(red[2][c], red[3][c]));
    else
      v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (pooling == 0) v *= wgt;
    if (len <= 0) v = 0.f;
    out[(size_t)n * D + c] = (OutT)v;
  }
}

// Fast path of the headline configuration (d = 1024, fp16 residual stream, no `encoded_seqs` output): the
// kernel is a pure HBM read (2 KB per token in, 2-4 KB per SENTENCE out), so all that matters is bytes in
// flight: 16-B loads, all loads of a wave step issued before the first reduction (the generic kernel above
// walks one row at a time with 8-B loads: 3.9 TB/s).  Template shape: NW waves per sentence, NR rows per lane and
// step, MINW = waves/SIMD the register budget must allow.  Sentences are short (16 tokens in the headline
// workload = 32 KB), so the kernel is a latency chain per sentence and what pays is SENTENCES in flight per CU:
// tile-major x, mean pooling: 4 waves x 1 row pair, 4 workgroups per CU (63 us; 8 waves x 2 pairs needs 168
// VGPRs = 1 workgroup per CU: 71 us; r02 experiment 15).
// Lane -> data map of a wave step (4 rows): row-major x: 2 chunks (columns h*512 + lane*8) of all 4 rows;
// tile-major x: the pair map at the top of the file, 4 chunks of the lane's row of each of the 2 row pairs.
template <bool TM, int NR_>
struct PoolMap {
  static constexpr int NR = NR_;                  // rows per lane and step
  static constexpr int STEP = TM ? 2 * NR_ : NR_;  // rows per wave and step
  static constexpr int NC = TM ? 4 : 2;  // 16-B chunks per row
  __device__ static int row(int lane, int j) { return TM ? 2 * j + pair_rsel(lane) : j; }
  __device__ static int col(int lane, int c) { return TM ? pair_col(lane, c) : c * 512 + lane * 8; }
  __device__ static float row_sum(float v) { return TM ? pair_sum(v) : wave_sum(v); }
};

template <typename OutT, bool TM, bool MEAN, int NW, int NR, int MINW>  // MEAN: pooling == 0 (compile time: the other modes hold w, b per lane)
__global__ __launch_bounds__(NW * 64, MINW) void ln_pool1024_f16_kernel(const f16* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float eps,
                                                              const int32_t* __restrict__ cu, OutT* __restrict__ out,
                                                              int pooling) {
  using M = PoolMap<TM, NR>;
  constexpr int D = 1024;
  __shared__ __attribute__((aligned(16))) float red[NW][D];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int start = cu[n];
  const int len = cu[n + 1] - start;
  const int end = start + len;
  const float init = pooling == 1 ? -INFINITY : 0.f;
  float acc[M::NC][8];
#pragma unroll
  for (int c = 0; c < M::NC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[c][e] = init;
  // wave wv owns packed rows g0 .. g0+3, g0 = base + 4*(wv + NW*i); tile-major: base is even so that the row
  // pairs are line aligned (the row in front of an odd start belongs to the previous sentence: read, not pooled)
  const int base = TM ? (start & ~1) : start;
  for (int g0 = base + wv * M::STEP; g0 < end; g0 += M::STEP * NW) {
    half8 v[M::NR][M::NC];
#pragma unroll
    for (int j = 0; j < M::NR; ++j) {
      const int g = min(g0 + M::row(lane, j), end - 1);  // clamped rows are loaded (cached) but not accumulated
#pragma unroll
      for (int c = 0; c < M::NC; ++c)
        v[j][c] = *(const half8*)(TM ? x + tm_offset(g, M::col(lane, c), D) : x + (size_t)g * D + M::col(lane, c));
    }
#pragma unroll
    for (int j = 0; j < M::NR; ++j) {
      float f[M::NC][8];
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < M::NC; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          f[c][e] = (float)v[j][c][e];
          sum += f[c][e];
        }
      const float mean = M::row_sum(sum) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < M::NC; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          f[c][e] -= mean;
          q += f[c][e] * f[c][e];
        }
      const float rstd = 1.0f / sqrtf(M::row_sum(q) * (1.0f / D) + eps);
      const int g = g0 + M::row(lane, j);
      const bool live = g >= start && g < end;
      const bool last = g == end - 1;
      if constexpr (MEAN) {
        // mean pooling: sum_r LN(x_r) = w * sum_r (x_r - mean_r) * rstd_r + len * b (one fma per element and row
        // here; the kernel is VALU-bound otherwise)
        const float sc = live ? rstd : 0.f;
#pragma unroll
        for (int c = 0; c < M::NC; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[c][e] = __builtin_fmaf(f[c][e], sc, acc[c][e]);
      } else {
        // max / last pooling need the affine part per row; w, b are re-read (L1) to keep the mean path's registers
#pragma unroll
        for (int c = 0; c < M::NC; ++c) {
          const int col = M::col(lane, c);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float y = f[c][e] * rstd * w[col + e] + b[col + e];
            if (pooling == 1)
              acc[c][e] = live ? fmaxf(acc[c][e], y) : acc[c][e];
            else if (last)
              acc[c][e] = y;
          }
        }
      }
    }
  }
  if constexpr (TM) {  // the two rows of a pair sit in lanes 4 apart: fold them, the even-row lanes publish
#pragma unroll
    for (int c = 0; c < M::NC; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float o = __shfl_xor(acc[c][e], 4, 64);
        acc[c][e] = pooling == 1 ? fmaxf(acc[c][e], o) : acc[c][e] + o;
      }
  }
  if (!TM || pair_rsel(lane) == 0) {
#pragma unroll
    for (int c = 0; c < M::NC; ++c) {
      float* r = &red[wv][M::col(lane, c)];
      *(f32x4*)r = f32x4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]};
      *(f32x4*)(r + 4) = f32x4{acc[c][4], acc[c][5], acc[c][6], acc[c][7]};
    }
  }
  __syncthreads();
  // reference (model.py:115-124): weights = 1/(seq_len + 1e-7), in fp32 here
  const float wgt = 1.0f / ((float)len + 1e-7f);
  for (int c = threadIdx.x; c < D; c += 64 * NW) {
    float v = red[0][c];
#pragma unroll
    for (int i = 1; i < NW; ++i) v = pooling == 1 ? fmaxf(v, red[i][c]) : v + red[i][c];
    if (MEAN) v = (v * w[c] + (float)len * b[c]) * wgt;
    if (len <= 0) v = 0.f;
    out[(size_t)n * D + c] = (OutT)v;
  }
}

hipError_t launch_ln_pool(const void* x, const float* w, const float* b, float eps,
                          const int32_t* cu, void* out, int out_is_f32, void* encoded, int N, int S,
                          int d, int pooling, hipStream_t stream, int x_f16, int x_tm) {
  if (N <= 0 || pooling < 0 || pooling > 2) return hipErrorInvalidValue;
  if (x_tm && !(d == 1024 && x_f16 && !encoded)) return hipErrorInvalidValue;  // tile-major x: the fast path only
  if (d == 1024 && x_f16 && !encoded) {
#define SMI_LP1024_K(OutT, TM, MEAN, NW, NR, MINW)                                                                  \
  hipLaunchKernelGGL((ln_pool1024_f16_kernel<OutT, TM, MEAN, NW, NR, MINW>), dim3(N), dim3(NW * 64), 0, stream,      \
                     (const f16*)x, w, b, eps, cu, (OutT*)out, pooling)
#define SMI_LP1024(OutT)                                  \
  if (pooling != 0) {                                     \
    if (x_tm) {                                           \
      SMI_LP1024_K(OutT, true, false, 8, 2, 2);           \
    } else {                                              \
      SMI_LP1024_K(OutT, false, false, 8, 4, 2);          \
    }                                                     \
  } else if (!x_tm) {                                     \
    SMI_LP1024_K(OutT, false, true, 8, 4, 4);             \
  } else {                                                \
    SMI_LP1024_K(OutT, true, true, 4, 1, 4);              \
  }
    if (out_is_f32) {
      SMI_LP1024(float)
    } else {
      SMI_LP1024(f16)
    }
#undef SMI_LP1024_K
#undef SMI_LP1024
    return hipGetLastError();
  }
#define SMI_LP_LAUNCH(NV, OutT, XT)                                                                               \
  hipLaunchKernelGGL((ln_pool_kernel<NV, OutT, XT>), dim3(N), dim3(256), 0, stream, (const XT*)x, w, b, eps, cu, \
                     (OutT*)out, (OutT*)encoded, S, pooling);
#define SMI_LP_CASE(NV)                 \
  case NV * 256:                        \
    if (out_is_f32) {                   \
      if (x_f16) {                      \
        SMI_LP_LAUNCH(NV, float, f16)   \
      } else {                          \
        SMI_LP_LAUNCH(NV, float, float) \
      }                                 \
    } else {                            \
      if (x_f16) {                      \
        SMI_LP_LAUNCH(NV, f16, f16)     \
      } else {                          \
        SMI_LP_LAUNCH(NV, f16, float)   \
      }                                 \
    }                                   \
    break;
  switch (d) {
    SMI_LP_CASE(1)
    SMI_LP_CASE(2)
    SMI_LP_CASE(3)
    SMI_LP_CASE(4)
    SMI_LP_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef SMI_LP_CASE
#undef SMI_LP_LAUNCH
  return hipGetLastError();
}

// ------------------------------------------------- row-major <-> tile-major
// One workgroup per (256-row block, 32-column block): 16 KiB, thread t moves 16-B chunks.
__global__ __launch_bounds__(256) void pack_tile_major_kernel(const f16* __restrict__ src,
                                                              f16* __restrict__ dst, int K, int inverse) {
  const int kb = blockIdx.x, rb = blockIdx.y;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = threadIdx.x + 256 * i;  // (row, chunk)
    const int rr = id >> 2, c = id & 3;
    const size_t rm = (size_t)(rb * 256 + rr) * K + kb * 32 + c * 8;
    const size_t tm = tm_offset(rb * 256 + rr, kb * 32 + c * 8, K);
    if (inverse)
      *(half8*)(dst + rm) = *(const half8*)(src + tm);
    else
      *(half8*)(dst + tm) = *(const half8*)(src + rm);
  }
}

hipError_t launch_pack_tile_major(const f16* src, f16* dst, int rows, int K, int inverse,
                                  hipStream_t stream) {
  if (rows <= 0 || rows % TM_ROWS || K <= 0 || K % 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_tile_major_kernel, dim3(K / 32, rows / 256), dim3(256), 0, stream, src, dst, K,
                     inverse);
  return hipGetLastError();
}

// ------------------------------------------- split-K slabs -> residual stream
// x[i] += sum_z parts[z][i] (fp32 adds in slab order, one rounding to the stream's type): the consumer of a
// split-K GEMM that writes fp32 slabs (launch_gemm_tn_splitk; the bias sits in slab 0).  Small batches only
// (api.hip): a K = 8192 GEMM with a handful of output tiles would otherwise run 256 K slices per tile on a few CUs.
template <typename XT, typename PT>
__global__ __launch_bounds__(256) void fold_residual_kernel(XT* __restrict__ x, const PT* __restrict__ parts,
                                                            int nparts, size_t part_elems, size_t n8) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float acc[8];
    if constexpr (sizeof(XT) == 2) {
      const half8 xv = *(const half8*)(x + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = (float)xv[e];
    } else {
      const f32x4 a = *(const f32x4*)(x + i * 8), b = *(const f32x4*)(x + i * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[e] = a[e];
        acc[4 + e] = b[e];
      }
    }
    float sum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sum[e] = 0.f;
    for (int z = 0; z < nparts; ++z) {
      const PT* pz = parts + (size_t)z * part_elems + i * 8;
      if constexpr (sizeof(PT) == 2) {
        const half8 hv = *(const half8*)pz;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] += (float)hv[e];
      } else {
        const f32x4 a = *(const f32x4*)pz, b = *(const f32x4*)(pz + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sum[e] += a[e];
          sum[4 + e] += b[e];
        }
      }
    }
    if constexpr (sizeof(XT) == 2) {
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)(acc[e] + sum[e]);
      *(half8*)(x + i * 8) = o;
    } else {
      *(f32x4*)(x + i * 8) = f32x4{acc[0] + sum[0], acc[1] + sum[1], acc[2] + sum[2], acc[3] + sum[3]};
      *(f32x4*)(x + i * 8 + 4) = f32x4{acc[4] + sum[4], acc[5] + sum[5], acc[6] + sum[6], acc[7] + sum[7]};
    }
  }
}

hipError_t launch_fold_residual(void* x, int x_f16, const void* parts, int nparts, size_t part_elems, size_t n,
                                hipStream_t stream, int parts_f16) {
  if (n % 8 || nparts < 1) return hipErrorInvalidValue;
  const size_t n8 = n / 8;
  const int blocks = (int)std::min<size_t>((n8 + 255) / 256, 256 * 16);
#define SMI_FR(XT, PT)                                                                                               \
  hipLaunchKernelGGL((fold_residual_kernel<XT, PT>), dim3(blocks), dim3(256), 0, stream, (XT*)x, (const PT*)parts, nparts, \
                     part_elems, n8)
  if (x_f16) {
    if (parts_f16) SMI_FR(f16, f16); else SMI_FR(f16, float);
  } else {
    if (parts_f16) SMI_FR(float, f16); else SMI_FR(float, float);
  }
#undef SMI_FR
  return hipGetLastError();
}

// ------------------------------------------------------------- LayerNorm folded into the GEMMs (kernels.hpp: GemmLnFold)
// One wave per weight row n: Wf[n][k] = f16(W[n][k] * g[k]); c1[n] = sum_k float(Wf[n][k]) -- the ROUNDED values, because
// that is what the GEMM multiplies --; c2[n] = sum_k b[k] * W[n][k] + bias[n].
__global__ __launch_bounds__(256) void ln_fold_prep_kernel(const f16* __restrict__ W, const float* __restrict__ g,
                                                           const float* __restrict__ b, const float* __restrict__ bias,
                                                           f16* __restrict__ Wf, float* __restrict__ c1,
                                                           float* __restrict__ c2, int N, int K, int centered) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float shift = 0.f;
  if (centered) {  // row mean of the scaled weights (unrounded), subtracted from every element
    float t = 0.f;
    for (int k = lane; k < K; k += 64) t += (float)W[(size_t)n * K + k] * g[k];
    shift = wave_sum(t) / K;
  }
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = (float)W[(size_t)n * K + k];
    const f16 wf = (f16)(w * g[k] - shift);
    Wf[(size_t)n * K + k] = wf;
    s1 += (float)wf;
    s2 += b[k] * w;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    c1[n] = s1;
    c2[n] = s2 + (bias ? bias[n] : 0.f);
  }
}

hipError_t launch_ln_fold_prep(const f16* W, const float* g, const float* b, const float* bias, f16* Wf, float* c1,
                               float* c2, int N, int K, int centered, hipStream_t stream) {
  if (N <= 0 || K <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ln_fold_prep_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, W, g, b, bias, Wf, c1, c2, N, K, centered);
  return hipGetLastError();
}

// (sum, sum of squares) of every row of the tile-major fp16 stream (the embedding output: the first LayerNorm of a
// forward has no producing GEMM).  One wave per row, 16-B loads.
__global__ __launch_bounds__(256) void row_stats_tm_kernel(const f16* __restrict__ x, float2* __restrict__ part, int M, int d,
                                                           int nparts) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= M) return;
  float s = 0.f, q = 0.f;
  for (int c = lane * 8; c < d; c += 512) {
    const half8 v = *(const half8*)(x + tm_offset(r, c, d));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = (float)v[i];
      s += t;
      q += t * t;
    }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if (lane == 0) part[r] = float2{s, q};
  if (lane >= 1 && lane < nparts) part[(size_t)lane * M + r] = float2{0.f, 0.f};
}

hipError_t launch_row_stats_tm(const f16* x_tm, float2* part, int M, int d, int nparts, hipStream_t stream) {
  if (M <= 0 || d % 8 || nparts < 1 || nparts > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(row_stats_tm_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, x_tm, part, M, d, nparts);
  return hipGetLastError();
}

// ------------------------------------------------------------- conversions
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ s,
                                                         f16* __restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t nv = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
    const f32x4 v = ((const f32x4*)s)[i];
    half4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (f16)v[k];
    ((half4*)d)[i] = o;
  }
  for (size_t i = nv * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    d[i] = (f16)s[i];
}
__global__ __launch_bounds__(256) void f16_to_f32_kernel(const f16* __restrict__ s,
                                                         float* __restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = (float)s[i];
}

// generic element-wise cast between the boundary dtypes (fp32 / fp16 / bf16): the engine computes in fp16 operands with
// fp32 accumulation whatever the model's nominal dtype; a bf16 model's tensors enter and leave through this kernel
// (bf16 -> fp32 is exact; fp32 -> bf16 rounds to nearest even, NaN kept quiet)
struct Bf16 { uint16_t bits; };
__device__ __forceinline__ float cast_load(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float cast_load(const f16* p, size_t i) { return (float)p[i]; }
__device__ __forceinline__ float cast_load(const Bf16* p, size_t i) { return __uint_as_float((uint32_t)p[i].bits << 16); }
__device__ __forceinline__ void cast_store(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void cast_store(f16* p, size_t i, float v) { p[i] = (f16)v; }
__device__ __forceinline__ void cast_store(Bf16* p, size_t i, float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) {
    u = (u >> 16) | 0x40u;  // NaN: keep it a (quiet) NaN
  } else {
    u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;  // round to nearest even
  }
  p[i].bits = (uint16_t)u;
}
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, D* __restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) cast_store(d, i, cast_load(s, i));
}

hipError_t launch_cast(const void* src, int src_dtype, void* dst, int dst_dtype, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if (src_dtype < 0 || src_dtype > 2 || dst_dtype < 0 || dst_dtype > 2) return hipErrorInvalidValue;
  const int blocks = (int)min((size_t)8192, (n + 255) / 256);
#define SMI_CAST(SI, ST, DI, DT)                                                                              \
  if (src_dtype == SI && dst_dtype == DI) {                                                                   \
    hipLaunchKernelGGL((cast_kernel<ST, DT>), dim3(blocks), dim3(256), 0, stream, (const ST*)src, (DT*)dst, n); \
    return hipGetLastError();                                                                                 \
  }
  SMI_CAST(0, float, 0, float) SMI_CAST(0, float, 1, f16) SMI_CAST(0, float, 2, Bf16)
  SMI_CAST(1, f16, 0, float) SMI_CAST(1, f16, 1, f16) SMI_CAST(1, f16, 2, Bf16)
  SMI_CAST(2, Bf16, 0, float) SMI_CAST(2, Bf16, 1, f16) SMI_CAST(2, Bf16, 2, Bf16)
#undef SMI_CAST
  return hipErrorInvalidValue;
}

hipError_t launch_f32_to_f16(const float* src, f16* dst, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)min((size_t)8192, (n / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, n);
  return hipGetLastError();
}
hipError_t launch_f16_to_f32(const f16* src, float* dst, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)min((size_t)8192, (n + 255) / 256);
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, n);
  return hipGetLastError();
}

}  // namespace smi
