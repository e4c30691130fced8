// Generic-dimension kernels: the part of the reference's configuration space the fast engines do not tile for.
//
// The MFMA engines of this library are shaped for the released SONAR models (head_dim 64, model_dim a multiple of
// 256, mean / max / last pooling).  The reference's factories accept more: any model_dim / head count
// (`toy` decoder: model_dim 32, 4 heads of 8 -- sonar/models/sonar_text/config.py:232-255), a conditioning vector whose
// width differs from model_dim (`input_dim`, factory.py:264, 276-282), and a text encoder whose sentence vector comes
// from an attention pooler with `embedding_dim != model_dim` (factory.py:155-226), all exercised by
// tests/unit_tests/test_low_dimension_text_models.py.  Those models run HERE: fp32 activations and fp32 weights,
// plain FMA tiles and one-wave-per-query attention, no shape restrictions beyond head_dim <= 256.  It is a
// correctness path on the GPU (there is still no CPU fallback), not a tuned one: the models it serves are test-sized.
#include "common.hpp"
#include "kernels.hpp"

namespace smi {

// ------------------------------------------------------------------ Y = act(X . W^T + b) (+ R)
// 64 x 64 output tile per 256-thread workgroup, K in steps of 16 through LDS, 4 x 4 outputs per thread.
__global__ __launch_bounds__(256) void flex_linear_kernel(const float* __restrict__ X, int ldx,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          float* __restrict__ Y, int ldy, int M, int N, int K, int act,
                                                          const float* __restrict__ R, int ldr) {
  __shared__ float xs[16][64 + 1], ws[16][64 + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int r = i >> 4, kk = i & 15;
      xs[kk][r] = (m0 + r < M && k0 + kk < K) ? X[(size_t)(m0 + r) * ldx + k0 + kk] : 0.f;
      ws[kk][r] = (n0 + r < N && k0 + kk < K) ? W[(size_t)(n0 + r) * K + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = xs[kk][ty * 4 + i];
        b[i] = ws[kk][tx * 4 + i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = acc[i][j] + (bias ? bias[n] : 0.f);
      if (act == 1) v = fmaxf(v, 0.f);
      if (R) v += R[(size_t)m * ldr + n];
      Y[(size_t)m * ldy + n] = v;
    }
  }
}

hipError_t launch_flex_linear(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M,
                              int N, int K, int act, const float* R, int ldr, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(flex_linear_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, stream, X, ldx, W, bias, Y,
                     ldy, M, N, K, act, R, ldr);
  return hipGetLastError();
}

// ------------------------------------------------------------------ y = LN(x) * w + b, one wave per row, any d
__global__ __launch_bounds__(256) void flex_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float eps,
                                                             float* __restrict__ y, int rows, int d) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xr = x + (size_t)r * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += xr[c];
  const float mean = wave_sum(s) / d;
  float q = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float t = xr[c] - mean;
    q += t * t;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / d + eps);
  float* yr = y + (size_t)r * d;
  for (int c = lane; c < d; c += 64) yr[c] = (xr[c] - mean) * rstd * w[c] + b[c];
}

hipError_t launch_flex_layernorm(const float* x, const float* w, const float* b, float eps, float* y, int rows, int d,
                                 hipStream_t stream) {
  if (rows <= 0 || d <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(flex_layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, w, b, eps, y, rows, d);
  return hipGetLastError();
}

// ------------------------------------------------------------------ x[r] = E[id[r]] * scale (+ PE[pos(r)])
// ids: [rows] int64 (encoder: the padded [n, s] batch, position = r % s) or int32 (decoder: one token per row, all at
// position `fixed_pos`).  Out-of-range ids raise *bad (if given) and read row 0.
__global__ __launch_bounds__(256) void flex_embed_kernel(const int64_t* __restrict__ ids64, const int32_t* __restrict__ ids32,
                                                         const float* __restrict__ table, const float* __restrict__ pe,
                                                         float scale, float* __restrict__ x, int rows, int d, int s,
                                                         int pos_offset, int fixed_pos, int64_t vocab,
                                                         int32_t* __restrict__ bad) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  int64_t t = ids64 ? ids64[r] : (int64_t)ids32[r];
  if (t < 0 || t >= vocab) {
    if (bad && lane == 0) *bad = 1;
    t = 0;
  }
  const int p = (fixed_pos >= 0 ? fixed_pos : r % s) + pos_offset;
  const float* e = table + (size_t)t * d;
  for (int c = lane; c < d; c += 64) x[(size_t)r * d + c] = e[c] * scale + (pe ? pe[(size_t)p * d + c] : 0.f);
}

hipError_t launch_flex_embed(const int64_t* ids64, const int32_t* ids32, const float* table, const float* pe,
                             float scale, float* x, int rows, int d, int s, int pos_offset, int fixed_pos,
                             int64_t vocab, int32_t* bad, hipStream_t stream) {
  if (rows <= 0 || (!ids64 && !ids32)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(flex_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, ids64, ids32, table, pe, scale, x,
                     rows, d, s, pos_offset, fixed_pos, vocab, bad);
  return hipGetLastError();
}

// ------------------------------------------------------------------ attention, one wave per (batch item, head, query)
// Scores of the wave's query against keys j = lane, lane + 64, ...; two passes (maximum, then exp-sum and the weighted
// V sum), every reduction a wave reduction.  head_dim <= 256.  Key row of (item b, position j): kbase + krow(b, j) * ldk.
struct FlexKeys {
  const float* k;
  const float* v;
  int ldk;             // row stride (elements) of k and v
  int sk;              // key positions per item (contiguous layout)
  const int32_t* anc;  // decoder: ancestry table [rows][anc_stride] -> cache row of position j (self at j == nk - 1)
  int anc_stride;
  size_t pos_stride;   // decoder: elements between the cache slabs of consecutive positions
};

__device__ __forceinline__ size_t flex_key_row(const FlexKeys& K, int b, int j, int nk) {
  if (K.anc) return (size_t)j * K.pos_stride + (size_t)(j == nk - 1 ? b : K.anc[(size_t)b * K.anc_stride + j]) * K.ldk;
  return ((size_t)b * K.sk + j) * K.ldk;
}

__global__ __launch_bounds__(256) void flex_attention_kernel(const float* __restrict__ q, int ldq, FlexKeys K,
                                                             float* __restrict__ out, int ldo, int items, int sq,
                                                             const int32_t* __restrict__ klens, int nk_fixed, int heads,
                                                             int hd, int causal, float scale) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= items * heads * sq) return;
  const int i = wid % sq, h = (wid / sq) % heads, b = wid / (sq * heads);
  int nk = klens ? klens[b] : nk_fixed;
  if (causal) nk = min(nk, i + 1);
  const float* qr = q + ((size_t)b * sq + i) * ldq + h * hd;
  float* orow = out + ((size_t)b * sq + i) * ldo + h * hd;
  if (nk <= 0) {  // no valid key (an empty sequence): zeros, as a fully masked softmax row is defined here
    for (int e = lane; e < hd; e += 64) orow[e] = 0.f;
    return;
  }
  float mx = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    const float* kr = K.k + flex_key_row(K, b, j, nk) + h * hd;
    float s = 0.f;
    for (int e = 0; e < hd; ++e) s += qr[e] * kr[e];
    mx = fmaxf(mx, s * scale);
  }
  mx = wave_max(mx);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // lane's output dims e = lane, lane + 64, ... (hd <= 256)
  float l = 0.f;
  for (int j0 = 0; j0 < nk; j0 += 64) {
    const int j = j0 + lane;
    float p = 0.f;
    if (j < nk) {
      const float* kr = K.k + flex_key_row(K, b, j, nk) + h * hd;
      float s = 0.f;
      for (int e = 0; e < hd; ++e) s += qr[e] * kr[e];
      p = __expf(s * scale - mx);
    }
    l += p;
    // every lane needs every p of the chunk: broadcast lane by lane (the chunk has <= 64 keys)
    const int cnt = min(64, nk - j0);
    for (int t = 0; t < cnt; ++t) {
      const float pt = __shfl(p, t, 64);
      const float* vr = K.v + flex_key_row(K, b, j0 + t, nk) + h * hd;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = lane + 64 * u;
        if (e < hd) acc[u] += pt * vr[e];
      }
    }
  }
  l = wave_sum(l);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = lane + 64 * u;
    if (e < hd) orow[e] = acc[u] / l;
  }
}

hipError_t launch_flex_attention(const float* q, int ldq, const float* k, const float* v, int ldk, float* out, int ldo,
                                 int items, int sq, int sk, const int32_t* klens, int heads, int hd, int causal,
                                 hipStream_t stream) {
  if (items <= 0 || sq <= 0 || sk <= 0 || hd <= 0 || hd > 256) return hipErrorInvalidValue;
  FlexKeys K{k, v, ldk, sk, nullptr, 0, 0};
  const int waves = items * heads * sq;
  hipLaunchKernelGGL(flex_attention_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, q, ldq, K, out, ldo, items, sq,
                     klens, sk, heads, hd, causal, 1.0f / sqrtf((float)hd));
  return hipGetLastError();
}

// decoder step: one query per row at position `pos`, keys 0..pos gathered through the ancestry table from the
// fp32 cache kv[pos][rows_pad][3 d] (q | k | v)
hipError_t launch_flex_dec_attention(const float* kv, const int32_t* anc, int anc_stride, float* ctx, int rows,
                                     int rows_pad, int d, int heads, int pos, hipStream_t stream) {
  const int hd = d / heads;
  if (rows <= 0 || hd <= 0 || hd > 256) return hipErrorInvalidValue;
  const size_t slab = (size_t)rows_pad * 3 * d;
  FlexKeys K{kv + d, kv + 2 * d, 3 * d, 0, anc, anc_stride, slab};
  const int waves = rows * heads;
  hipLaunchKernelGGL(flex_attention_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, kv + (size_t)pos * slab, 3 * d,
                     K, ctx, d, rows, 1, nullptr, pos + 1, heads, hd, 0, 1.0f / sqrtf((float)hd));
  return hipGetLastError();
}

// ------------------------------------------------------------------ static pooling over the valid positions
// (model.py:86-128: mean with 1 / (len + 1e-7), max, last) of x [n, s, d] -> out [n, d]
__global__ __launch_bounds__(256) void flex_pool_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                        int pooling, float* __restrict__ out, int n, int s, int d) {
  const int b = blockIdx.x;
  const int len = lens ? lens[b] : s;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float* col = x + (size_t)b * s * d + c;
    float v;
    if (pooling == 2) {
      v = col[(size_t)max(len - 1, 0) * d];
    } else if (pooling == 1) {
      v = -INFINITY;
      for (int j = 0; j < len; ++j) v = fmaxf(v, col[(size_t)j * d]);
    } else {
      v = 0.f;
      for (int j = 0; j < len; ++j) v += col[(size_t)j * d];
      v *= 1.0f / ((float)len + 1e-7f);
    }
    out[(size_t)b * d + c] = v;
  }
}

hipError_t launch_flex_pool(const float* x, const int32_t* lens, int pooling, float* out, int n, int s, int d,
                            hipStream_t stream) {
  if (n <= 0 || s <= 0 || d <= 0 || pooling < 0 || pooling > 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(flex_pool_kernel, dim3(n), dim3(256), 0, stream, x, lens, pooling, out, n, s, d);
  return hipGetLastError();
}

// ------------------------------------------------------------------ per (row, 256-column tile) softmax statistics
// of a logits matrix: what the MFMA logits GEMM leaves for vocab_select (decoder.hip): tile maximum of
// v * scale and sum exp(v * scale - max), layout [tile][stat_rows].
__global__ __launch_bounds__(256) void flex_tile_stats_kernel(const float* __restrict__ logits, int ld, int vocab,
                                                              float scale, float* __restrict__ tile_max,
                                                              float* __restrict__ tile_sum, int stat_rows) {
  __shared__ float red[4];
  const int row = blockIdx.x, tile = blockIdx.y, tid = threadIdx.x;
  const int c = tile * 256 + tid;
  const float v = c < vocab ? logits[(size_t)row * ld + c] * scale : -INFINITY;
  float m = wave_max(v);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float e = (c < vocab && m != -INFINITY) ? __expf(v - m) : 0.f;
  e = wave_sum(e);
  if ((tid & 63) == 0) red[tid >> 6] = e;
  __syncthreads();
  if (tid == 0) {
    tile_max[(size_t)tile * stat_rows + row] = m;
    tile_sum[(size_t)tile * stat_rows + row] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

hipError_t launch_flex_tile_stats(const float* logits, int ld, int rows, int vocab, float scale, float* tile_max,
                                  float* tile_sum, int stat_rows, hipStream_t stream) {
  if (rows <= 0 || vocab <= 0 || stat_rows < rows) return hipErrorInvalidValue;
  hipLaunchKernelGGL(flex_tile_stats_kernel, dim3(rows, (vocab + 255) / 256), dim3(256), 0, stream, logits, ld, vocab,
                     scale, tile_max, tile_sum, stat_rows);
  return hipGetLastError();
}

// x[r] += c[r / group]  (the decoder's per-sentence cross-attention constant)
__global__ void flex_add_rows_kernel(float* __restrict__ x, const float* __restrict__ c, int rows, int d, int group) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * d) return;
  const int r = (int)(i / d), col = (int)(i % d);
  x[i] += c[(size_t)(r / group) * d + col];
}

hipError_t launch_flex_add_rows(float* x, const float* c, int rows, int d, int group, hipStream_t stream) {
  const size_t n = (size_t)rows * d;
  hipLaunchKernelGGL(flex_add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, c, rows, d, group);
  return hipGetLastError();
}

// encoded_seqs output: out[b, j] = x[b, j] for valid positions, zeros for padding; fp32 or fp16
__global__ void flex_store_encoded_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens, int s, int d,
                                          void* __restrict__ out, int out_f16, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t row = i / d;
  const int b = (int)(row / s), j = (int)(row % s);
  const float v = j < lens[b] ? x[i] : 0.f;
  if (out_f16)
    ((f16*)out)[i] = (f16)v;
  else
    ((float*)out)[i] = v;
}

hipError_t launch_flex_store_encoded(const float* x, const int32_t* lens, int n, int s, int d, void* out, bool out_f16,
                                     hipStream_t stream) {
  const size_t total = (size_t)n * s * d;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(flex_store_encoded_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, lens, s, d, out,
                     out_f16 ? 1 : 0, total);
  return hipGetLastError();
}

}  // namespace smi
