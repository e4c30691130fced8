// Device kernels of the SpeechToEmbedding hot path: Kaldi-style log-mel filterbank,
// frame stacking + LayerNorm, the conformer block's relative-position attention and
// depthwise convolution, LayerNorm variants, and the attention pooler's single-query
// cross-attention.  GEMMs (FFNs, projections, pointwise convolutions) run on the shared
// MFMA tile engines in gemm.hip.
//
// Reference: sonar/inference_pipelines/speech.py:283-308,431-474 (fbank options, frame
// padding to a multiple of 2), sonar/models/sonar_speech/{factory.py:53-152, model.py:59-77},
// sonar/nn/encoder_pooler.py:70-89; conformer / Wav2Vec2Frontend / RelativePositionSDPA
// semantics of fairseq2 ~=0.4 as listed in SURVEY a26-a29.
#include "common.hpp"
#include "kernels.hpp"

namespace smi {

// ================================================================== filterbank
// One workgroup per frame: 400 samples -> remove DC -> pre-emphasis 0.97 -> povey window ->
// zero-pad to 512 -> radix-2 FFT in LDS -> power -> 80 triangular mel bins -> log.
constexpr int FB_WIN = 400, FB_SHIFT = 160, FB_NFFT = 512, FB_BINS = 80;

__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wave, float scale,
                                                    const float* __restrict__ window,
                                                    const float* __restrict__ mel_w,  // [80][256]
                                                    const int* __restrict__ mel_range,  // [80][2]
                                                    float* __restrict__ out) {
  __shared__ float re[FB_NFFT], im[FB_NFFT];
  __shared__ float red[4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* src = wave + (size_t)f * FB_SHIFT;
  // load (two samples per thread, 400 valid)
  float a0 = tid < FB_WIN ? src[tid] * scale : 0.f;
  float a1 = tid + 256 < FB_WIN ? src[tid + 256] * scale : 0.f;
  float s = wave_sum(a0 + a1);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / FB_WIN;
  if (tid < FB_WIN) re[tid] = a0 - mean;
  if (tid + 256 < FB_WIN) re[tid + 256] = a1 - mean;
  __syncthreads();
  // pre-emphasis + window, bit-reversed scatter for the in-place DIT FFT
  float y0 = 0.f, y1 = 0.f;
  if (tid < FB_WIN) y0 = (re[tid] - 0.97f * re[tid == 0 ? 0 : tid - 1]) * window[tid];
  if (tid + 256 < FB_WIN) y1 = (re[tid + 256] - 0.97f * re[tid + 255]) * window[tid + 256];
  __syncthreads();
  {
    const int r0 = __brev((unsigned)tid) >> 23, r1 = __brev((unsigned)(tid + 256)) >> 23;
    re[r0] = y0;
    im[r0] = 0.f;
    re[r1] = y1;
    im[r1] = 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int st = 1; st <= 9; ++st) {
    const int half = 1 << (st - 1);
    const int grp = tid >> (st - 1), pos = tid & (half - 1);
    const int i0 = grp * (half << 1) + pos, i1 = i0 + half;
    float sn, cs;
    sincospif(-(float)pos / (float)half, &sn, &cs);
    const float xr = re[i1], xi = im[i1];
    const float tr = xr * cs - xi * sn, ti = xr * sn + xi * cs;
    const float ur = re[i0], ui = im[i0];
    re[i0] = ur + tr;
    im[i0] = ui + ti;
    re[i1] = ur - tr;
    im[i1] = ui - ti;
    __syncthreads();
  }
  // power spectrum of bins 0..255 (Kaldi's mel banks do not use the Nyquist bin)
  const float pw = re[tid] * re[tid] + im[tid] * im[tid];
  __syncthreads();
  re[tid] = pw;
  __syncthreads();
  if (tid < FB_BINS) {
    const int k0 = mel_range[tid * 2], k1 = mel_range[tid * 2 + 1];
    float e = 0.f;
    for (int k = k0; k < k1; ++k) e += mel_w[tid * 256 + k] * re[k];
    out[(size_t)f * FB_BINS + tid] = __logf(fmaxf(e, 1.1920929e-07f));
  }
}

// per-utterance standardisation over time (two-pass mean / unbiased std, as the reference's
// converter does).  Three short launches over `nblk` workgroups of 12 x 80 threads (thread (g, b)
// owns frames g, g+12, ... of its workgroup's frame range for mel bin b, loads independent so they
// overlap); column partials go through a stream-ordered scratch [2][nblk][80] (+ [80]: frame 0) and every workgroup
// folds the partials it needs itself (nblk <= 256).
//   pass 0: part0[blk][b] = sum_f (x - x[0])   pass 1: part1[blk][b] = sum_f (x - mean)^2
//   pass 2: x = (x - mean) / std
__global__ __launch_bounds__(960) void fbank_standardize_kernel(float* __restrict__ fb, int frames, int per_blk,
                                                                float* __restrict__ part, int nblk, int pass) {
  __shared__ float red[12][FB_BINS];
  __shared__ float stat[2][FB_BINS];
  const int b = threadIdx.x % FB_BINS, g = threadIdx.x / FB_BINS;
  float* part0 = part;
  float* part1 = part + (size_t)nblk * FB_BINS;
  float* x0_saved = part + (size_t)2 * nblk * FB_BINS;  // frame 0 as pass 0 saw it (pass 2 rewrites the frames in place)
  // fold the partials of the earlier passes
  for (int q = 0; q < pass; ++q) {
    const float* src = q == 0 ? part0 : part1;
    float t = 0.f;
    for (int i = g; i < nblk; i += 12) t += src[(size_t)i * FB_BINS + b];
    red[g][b] = t;
    __syncthreads();
    if (g == 0) {
      float v = 0.f;
      for (int i = 0; i < 12; ++i) v += red[i][b];
      stat[q][b] = q == 0 ? x0_saved[b] + v / frames : 1.0f / sqrtf(v / (frames - 1));
    }
    __syncthreads();
  }
  const int f0 = blockIdx.x * per_blk, f1 = min(frames, f0 + per_blk);
  // pass 0 sums around the column's first value (frame 0): exact mean for a constant column (see the batch kernel)
  const float mean = pass > 0 ? stat[0][b] : fb[b];
  if (pass == 0 && blockIdx.x == 0 && g == 0) x0_saved[b] = fb[b];
  if (pass == 2) {
    const float inv = stat[1][b];
    for (int f = f0 + g; f < f1; f += 12) {
      float* p = fb + (size_t)f * FB_BINS + b;
      *p = (*p - mean) * inv;
    }
    return;
  }
  float acc = 0.f;
  for (int f = f0 + g; f < f1; f += 12) {
    const float d = fb[(size_t)f * FB_BINS + b] - mean;
    acc += pass == 0 ? d : d * d;
  }
  red[g][b] = acc;
  __syncthreads();
  if (g == 0) {
    float v = 0.f;
    for (int i = 0; i < 12; ++i) v += red[i][b];
    (pass == 0 ? part0 : part1)[(size_t)blockIdx.x * FB_BINS + b] = v;
  }
}

// ---- batched filterbank: every clip of a batch in one launch, output zero-padded [n][tpad][80] ----
// grid (tpad, n): workgroup (f, c) computes frame f of clip c (samples off[c] + 160 f ..), or writes the
// zero padding the encoder's collation expects (speech.py:444, pad value 0) when f >= frames(c).
__global__ __launch_bounds__(256) void fbank_batch_kernel(const float* __restrict__ waves,
                                                          const int64_t* __restrict__ off, float scale,
                                                          const float* __restrict__ window,
                                                          const float* __restrict__ mel_w,
                                                          const int* __restrict__ mel_range,
                                                          float* __restrict__ out, int tpad) {
  __shared__ float re[FB_NFFT], im[FB_NFFT];
  __shared__ float red[4];
  const int f = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t ns = off[c + 1] - off[c];
  const int frames = ns < FB_WIN ? 0 : (int)(1 + (ns - FB_WIN) / FB_SHIFT);
  float* dst = out + ((size_t)c * tpad + f) * FB_BINS;
  if (f >= frames) {
    if (tid < FB_BINS) dst[tid] = 0.f;
    return;
  }
  const float* src = waves + off[c] + (size_t)f * FB_SHIFT;
  float a0 = tid < FB_WIN ? src[tid] * scale : 0.f;
  float a1 = tid + 256 < FB_WIN ? src[tid + 256] * scale : 0.f;
  float s = wave_sum(a0 + a1);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / FB_WIN;
  if (tid < FB_WIN) re[tid] = a0 - mean;
  if (tid + 256 < FB_WIN) re[tid + 256] = a1 - mean;
  __syncthreads();
  float y0 = 0.f, y1 = 0.f;
  if (tid < FB_WIN) y0 = (re[tid] - 0.97f * re[tid == 0 ? 0 : tid - 1]) * window[tid];
  if (tid + 256 < FB_WIN) y1 = (re[tid + 256] - 0.97f * re[tid + 255]) * window[tid + 256];
  __syncthreads();
  {
    const int r0 = __brev((unsigned)tid) >> 23, r1 = __brev((unsigned)(tid + 256)) >> 23;
    re[r0] = y0;
    im[r0] = 0.f;
    re[r1] = y1;
    im[r1] = 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int st = 1; st <= 9; ++st) {
    const int half = 1 << (st - 1);
    const int grp = tid >> (st - 1), pos = tid & (half - 1);
    const int i0 = grp * (half << 1) + pos, i1 = i0 + half;
    float sn, cs;
    sincospif(-(float)pos / (float)half, &sn, &cs);
    const float xr = re[i1], xi = im[i1];
    const float tr = xr * cs - xi * sn, ti = xr * sn + xi * cs;
    const float ur = re[i0], ui = im[i0];
    re[i0] = ur + tr;
    im[i0] = ui + ti;
    re[i1] = ur - tr;
    im[i1] = ui - ti;
    __syncthreads();
  }
  const float pw = re[tid] * re[tid] + im[tid] * im[tid];
  __syncthreads();
  re[tid] = pw;
  __syncthreads();
  if (tid < FB_BINS) {
    const int k0 = mel_range[tid * 2], k1 = mel_range[tid * 2 + 1];
    float e = 0.f;
    for (int k = k0; k < k1; ++k) e += mel_w[tid * 256 + k] * re[k];
    dst[tid] = __logf(fmaxf(e, 1.1920929e-07f));
  }
}

// per-clip standardisation of the batch: one workgroup of 12 x 80 threads per clip walks its frames
// three times (mean, unbiased variance around it, normalise) -- the same arithmetic order per column
// group as the single-clip path's partial sums is not needed for parity (both are two-pass fp32).
__global__ __launch_bounds__(960) void fbank_batch_standardize_kernel(float* __restrict__ fb,
                                                                      const int64_t* __restrict__ off, int tpad) {
  __shared__ float red[12][FB_BINS];
  __shared__ float stat[2][FB_BINS];
  const int c = blockIdx.x, b = threadIdx.x % FB_BINS, g = threadIdx.x / FB_BINS;
  const int64_t ns = off[c + 1] - off[c];
  const int frames = ns < FB_WIN ? 0 : (int)(1 + (ns - FB_WIN) / FB_SHIFT);
  if (frames < 2) return;
  float* x = fb + (size_t)c * tpad * FB_BINS + b;
  // The mean is accumulated around the column's first value: a CONSTANT column (digital silence: every frame is log(eps)) then
  // has mean == that value exactly, deviations 0, variance 0 and 0 * inf = NaN features -- what at::std_mean gives the
  // reference's converter and torch.std_mean the oracle (tests/unit_tests/test_sonar_speech.py:29-32 feeds zeros).  A plain
  // sum of 1 098 equal values is off by a few ulp, and the clip came out as finite noise of unit variance instead.
  const float x0 = x[0];
  for (int pass = 0; pass < 2; ++pass) {
    const float mean = pass ? stat[0][b] : x0;
    float acc = 0.f;
    for (int f = g; f < frames; f += 12) {
      const float d = x[(size_t)f * FB_BINS] - mean;
      acc += pass ? d * d : d;
    }
    red[g][b] = acc;
    __syncthreads();
    if (g == 0) {
      float v = 0.f;
      for (int i = 0; i < 12; ++i) v += red[i][b];
      stat[pass][b] = pass ? 1.0f / sqrtf(v / (frames - 1)) : x0 + v / frames;
    }
    __syncthreads();
  }
  const float mean = stat[0][b], inv = stat[1][b];
  for (int f = g; f < frames; f += 12) x[(size_t)f * FB_BINS] = (x[(size_t)f * FB_BINS] - mean) * inv;
}

hipError_t launch_fbank_batch(const float* waves, const int64_t* off_dev, int n, int tpad, float scale,
                              int standardize, const float* window, const float* mel_w, const int* mel_range,
                              float* out, hipStream_t stream) {
  if (n <= 0 || tpad <= 0) return hipSuccess;
  hipLaunchKernelGGL(fbank_batch_kernel, dim3(tpad, n), dim3(256), 0, stream, waves, off_dev, scale, window, mel_w,
                     mel_range, out, tpad);
  if (standardize)
    hipLaunchKernelGGL(fbank_batch_standardize_kernel, dim3(n), dim3(960), 0, stream, out, off_dev, tpad);
  return hipGetLastError();
}

hipError_t launch_fbank(const float* wave, int64_t nsamples, float scale, int standardize, const float* window,
                        const float* mel_w, const int* mel_range, float* out, hipStream_t stream) {
  if (nsamples < FB_WIN) return hipSuccess;
  const int frames = (int)(1 + (nsamples - FB_WIN) / FB_SHIFT);
  hipLaunchKernelGGL(fbank_kernel, dim3(frames), dim3(256), 0, stream, wave, scale, window, mel_w, mel_range, out);
  if (standardize && frames >= 2) {
    int per_blk = 48;
    if ((frames + per_blk - 1) / per_blk > 256) per_blk = ((frames + 255) / 256 + 11) / 12 * 12;
    const int nblk = (frames + per_blk - 1) / per_blk;
    float* part = nullptr;
    hipError_t e = hipMallocAsync((void**)&part, (size_t)(2 * nblk + 1) * FB_BINS * sizeof(float), stream);
    if (e != hipSuccess) return e;
    for (int pass = 0; pass < 3; ++pass)
      hipLaunchKernelGGL(fbank_standardize_kernel, dim3(nblk), dim3(960), 0, stream, out, frames, per_blk, part, nblk,
                         pass);
    e = hipFreeAsync(part, stream);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

// =========================================================== frame stacking + LayerNorm(160)
// out[row(n, j), 0..159] = f16(LN([fb[n, 2j, :], fb[n, 2j+1, :]])), columns 160..191 zero
// (K padded to a multiple of 64 for the projection GEMM).  One wave per stacked frame.
__global__ __launch_bounds__(256) void stack_ln_kernel(const float* __restrict__ fb, int t, int nb,
                                                       const int32_t* __restrict__ cu,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ b, float eps,
                                                       f16* __restrict__ out, int ldo) {
  const int n = blockIdx.x;
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int start = cu[n], len = cu[n + 1] - start;
  if (j >= len) return;
  const int fd = 2 * nb;  // 160
  const float* src = fb + ((size_t)n * t + 2 * j) * nb;  // 2 consecutive frames are contiguous
  float v[3];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < fd ? src[c] : 0.f;
    s += v[k];
  }
  const float mean = wave_sum(s) / fd;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < fd ? v[k] - mean : 0.f;
    q += v[k] * v[k];
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / fd + eps);
  f16* o = out + (size_t)(start + j) * ldo;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = lane + 64 * k;
    if (c < ldo) o[c] = c < fd ? (f16)(v[k] * rstd * w[c] + b[c]) : (f16)0.f;
  }
}

hipError_t launch_stack_ln(const float* fb, int n, int t, int nb, const int32_t* cu, int max_len, const float* w,
                           const float* b, float eps, f16* out, int ldo, hipStream_t stream) {
  if (2 * nb > 192 || ldo > 192) return hipErrorInvalidValue;
  dim3 grid(n, (max_len + 3) / 4);
  hipLaunchKernelGGL(stack_ln_kernel, grid, dim3(256), 0, stream, fb, t, nb, cu, w, b, eps, out, ldo);
  return hipGetLastError();
}

// =============================================================== LayerNorm variants (fp32 stream)
// x = LN1(x) in place (fp32); h = f16(w2 ? LN2(x) : x).  One wave per row.
template <int NV, bool TM, typename XT>
__global__ __launch_bounds__(256) void ln2_kernel(XT* __restrict__ x, const float* __restrict__ w1,
                                                  const float* __restrict__ b1,
                                                  const float* __restrict__ w2,
                                                  const float* __restrict__ b2, float eps,
                                                  f16* __restrict__ h, int rows, int x_tm) {
  constexpr int D = NV * 256;
  constexpr float inv_d = 1.0f / D;
  const int lane = threadIdx.x & 63;
  if constexpr (sizeof(XT) == 2 && NV % 2 == 0) {
    // fp16 stream: TWO rows per wave (round 6), lanes 0..31 the even row, 32..63 the odd one, a lane owns the 16-B chunks l, l + 32,
    // ... of its row.  In the tile-major layout (common.hpp) a row's share of a 32-column block is 64 B and the next row's follows
    // it: with one row per wave every access was a 64-B piece of a 128-B line (the stream read, its write-back and the tile-major
    // h); a row pair covers whole lines.  x_tm: the stream itself is tile-major; TM: h is.  8 rows per workgroup.
    const int half = lane >> 5, l = lane & 31;
    const int rr = blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + half;
    const bool live = rr < rows;
    const int r2 = live ? rr : rows - 1;
    XT* xr2 = x + (size_t)r2 * D;
    auto xp = [&](int k) { return x_tm ? x + tm_offset(r2, (l + 32 * k) * 8, D) : xr2 + (l + 32 * k) * 8; };
    auto half_sum = [](float v) {  // over the 32 lanes of a row: wave_sum without its last step
      v += dpp_f<0xB1>(v);
      v += dpp_f<0x4E>(v);
      v += dpp_f<0x141>(v);
      v += dpp_f<0x140>(v);
      const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      return __uint_as_float(a[0]) + __uint_as_float(a[1]);
    };
    float u[NV][8];
    float s8 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const half8 raw = *(const half8*)xp(k);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        u[k][i] = (float)raw[i];
        s8 += u[k][i];
      }
    }
    float mean8 = half_sum(s8) * inv_d, q8 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        u[k][i] -= mean8;
        q8 += u[k][i] * u[k][i];
      }
    float rstd8 = 1.0f / sqrtf(half_sum(q8) * inv_d + eps);
    s8 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float* wp = w1 + (l + 32 * k) * 8;
      const float* bp = b1 + (l + 32 * k) * 8;
      const f32x4 wa = *(const f32x4*)wp, wb = *(const f32x4*)(wp + 4), ba = *(const f32x4*)bp, bb = *(const f32x4*)(bp + 4);
      half8 hv;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float y = u[k][i] * rstd8 * (i < 4 ? wa[i] : wb[i - 4]) + (i < 4 ? ba[i] : bb[i - 4]);
        hv[i] = (f16)y;
        u[k][i] = (float)hv[i];  // the second LayerNorm sees what the stream holds
        s8 += u[k][i];
      }
      if (live) *(half8*)xp(k) = hv;
    }
    if (!h) return;
    if (w2) {
      mean8 = half_sum(s8) * inv_d;
      q8 = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          u[k][i] -= mean8;
          q8 += u[k][i] * u[k][i];
        }
      rstd8 = 1.0f / sqrtf(half_sum(q8) * inv_d + eps);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float* wp = w2 + (l + 32 * k) * 8;
        const float* bp = b2 + (l + 32 * k) * 8;
        const f32x4 wa = *(const f32x4*)wp, wb = *(const f32x4*)(wp + 4), ba = *(const f32x4*)bp, bb = *(const f32x4*)(bp + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) u[k][i] = u[k][i] * rstd8 * (i < 4 ? wa[i] : wb[i - 4]) + (i < 4 ? ba[i] : bb[i - 4]);
      }
    }
    if (!live) return;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      half8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (f16)u[k][i];
      if constexpr (TM)
        *(half8*)(h + tm_offset(rr, (l + 32 * k) * 8, D)) = o;  // one 16-B chunk of the tile-major operand
      else
        *(half8*)(h + (size_t)rr * D + (l + 32 * k) * 8) = o;
    }
    return;
  }
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  XT* xr = x + (size_t)r * D;
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
  float mean = wave_sum(s) * inv_d, q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[k][i] -= mean;
      q += v[k][i] * v[k][i];
    }
  float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
  s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const f32x4 wv = *(const f32x4*)(w1 + k * 256 + lane * 4);
    const f32x4 bv = *(const f32x4*)(b1 + k * 256 + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[k][i] = v[k][i] * rstd * wv[i] + bv[i];
    if constexpr (sizeof(XT) == 4) {
      *(f32x4*)(xr + k * 256 + lane * 4) = v[k];
    } else {
      half4 hv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hv[i] = (f16)v[k][i];
        v[k][i] = (float)hv[i];  // the second LayerNorm sees what the stream holds
      }
      *(half4*)(xr + k * 256 + lane * 4) = hv;
    }
    s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  if (!h) return;
  if (w2) {
    mean = wave_sum(s) * inv_d;
    q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[k][i] -= mean;
        q += v[k][i] * v[k][i];
      }
    rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const f32x4 wv = *(const f32x4*)(w2 + k * 256 + lane * 4);
      const f32x4 bv = *(const f32x4*)(b2 + k * 256 + lane * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[k][i] = v[k][i] * rstd * wv[i] + bv[i];
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    half4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (f16)v[k][i];
    if constexpr (TM)
      *(half4*)(h + tm_offset(r, k * 256 + lane * 4, D)) = o;  // tile-major GEMM operand (common.hpp)
    else
      *(half4*)(h + (size_t)r * D + k * 256 + lane * 4) = o;
  }
}

hipError_t launch_ln2(void* x, const float* w1, const float* b1, const float* w2, const float* b2, float eps,
                      f16* h, int rows, int d, hipStream_t stream, int out_tm, int x_f16, int x_tm) {
  if (rows <= 0) return hipErrorInvalidValue;
  if (x_tm && (!x_f16 || d % 512)) return hipErrorInvalidValue;  // the tile-major stream is fp16, 16-B chunks per lane
  // 4 rows per workgroup (one wave per row); the fp16 stream with an even NV: 8 (a row pair per wave)
#define SMI_LN2_LAUNCH(NV, TMF, XT)                                                                                      \
  hipLaunchKernelGGL((ln2_kernel<NV, TMF, XT>), dim3(sizeof(XT) == 2 && NV % 2 == 0 ? (rows + 7) / 8 : (rows + 3) / 4), dim3(256), 0, \
                     stream, (XT*)x, w1, b1, w2, b2, eps, h, rows, x_tm);
#define SMI_LN2_CASE(NV)                 \
  case NV * 256:                         \
    if (out_tm) {                        \
      if (x_f16) {                       \
        SMI_LN2_LAUNCH(NV, true, f16)    \
      } else {                           \
        SMI_LN2_LAUNCH(NV, true, float)  \
      }                                  \
    } else {                             \
      if (x_f16) {                       \
        SMI_LN2_LAUNCH(NV, false, f16)   \
      } else {                           \
        SMI_LN2_LAUNCH(NV, false, float) \
      }                                  \
    }                                    \
    break;
  switch (d) {
    SMI_LN2_CASE(1)
    SMI_LN2_CASE(2)
    SMI_LN2_CASE(3)
    SMI_LN2_CASE(4)
    SMI_LN2_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef SMI_LN2_CASE
#undef SMI_LN2_LAUNCH
  return hipGetLastError();
}

// ========================================================= relative-position self-attention
// scores[i][j] = ((q_i + u) . k_j + (q_i + v) . rp[i - j]) / 8, softmax over the clip's frames.
// Same transposed-score / lane-local-softmax structure as attention.hip; the position term of a
// 32-key block needs rp rows for the 63 relative offsets the (32 queries x 32 keys) block spans:
// G[rho][i] = rp[rel_lo + rho] . (q_i + v) comes out of 8 more MFMAs, goes through a per-wave LDS
// pad and is read back at row (ii - jj + 31), column ii -- bank = ii, conflict-free both ways.
constexpr int RA_QB = 128, RA_KB = 32;

// K and V blocks (32 keys x 128 B each) go global -> LDS by DMA, double-buffered: the block of the next
// iteration is in flight while this one is consumed, one barrier per block, no staging registers.
//  * both are ROW-major [key][128 B]; 16-B chunk c of key r sits at slot c ^ swz(r).  K: swz = (r>>1)&7
//    (conflict-free ds_read_b128 of the S^T = K (Q+u)^T A fragments).  V: swz = ((r>>1)&1)<<2, read with
//    ds_read_b64_tr_b16: a 16-lane group fetches a [4 keys][16 dims] block and every lane receives the 4
//    keys of ITS dim -- the V^T fragment of O^T += V^T P^T without a transposed copy (the swizzle puts
//    keys r, r+2 of a group in different halves of the 128-B row: 4 rows x 64 B = all 64 banks).
//  * a DMA instruction writes wave-base + lane*16: thread -> (key = tid>>3, slot = tid&7) fetches the
//    chunk that belongs in that slot.
// The transpose reads are inline asm: for the builtin, hipcc's LDS-DMA alias tracking puts
// `s_waitcnt vmcnt(0)` in front of the first read while the NEXT block's DMA is in flight (it cannot see
// that the DMA targets the other buffer), which would expose that latency in every iteration.  The asm
// results are waited for explicitly (lgkmcnt(0)) before the MFMAs read them.
template <int OFF>
__device__ __forceinline__ half4 ra_tr_read(unsigned lds_addr) {
  half4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
  return v;
}

// 3 waves per SIMD (<= 168 registers; 3 workgroups x 48 KiB of LDS per CU): the kernel is a chain of LDS round trips, DMA
// waits and one barrier per 32-key block, so a third resident workgroup is what hides them (round 4; it took 204 registers =
// 2 waves before the pad addresses and the first block's second pad half stopped occupying 31 registers across the loop).
// -DSMI_RELPOS_WAVES=2 restores the two-wave allocation for A/B builds.
#ifndef SMI_RELPOS_WAVES
#define SMI_RELPOS_WAVES 3
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMI_RELPOS_WAVES, SMI_RELPOS_WAVES))) void relpos_attention_kernel(const f16* __restrict__ qkv,
                                                               const int32_t* __restrict__ cu,
                                                               const f16* __restrict__ rp, int rp_zero,
                                                               int rp_rows,
                                                               const float* __restrict__ u_bias,
                                                               const float* __restrict__ v_bias,
                                                               f16* __restrict__ ctx, int d, float sl2e,
                                                               int ctx_tm) {
  constexpr int BLK = RA_KB * 128;  // 4 KiB
  __shared__ __attribute__((aligned(16))) char lds[4 * BLK + 4 * 64 * 32 * 4];
  const int n = blockIdx.x, h = blockIdx.y;
  const int start = cu[n], len = cu[n + 1] - start;
  const int q0 = blockIdx.z * RA_QB;
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  float* Gs = (float*)(lds + 4 * BLK) + wave * 64 * 32;  // [64 rho][32 queries]
  char* const kvb = lds;
  const size_t ld = (size_t)3 * d;
  const f16* qbase = qkv + (size_t)start * ld + h * 64;
  const f16* kbase = qbase + d;
  const f16* vbase = qbase + 2 * d;
  const f16* rph = rp + h * 64;

  const int i0 = q0 + wave * 32;
  const int qi = i0 + l31;
  const f16* qptr = qbase + (size_t)min(qi, len - 1) * ld;
  half8 qu[4], qv[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c0 = (ks * 2 + hi) * 8;
    const half8 q = *(const half8*)(qptr + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qu[ks][e] = (f16)((float)q[e] + u_bias[h * 64 + c0 + e]);
      qv[ks][e] = (f16)((float)q[e] + v_bias[h * 64 + c0 + e]);
    }
  }

  // DMA sources of this thread: key tid>>3 of a block, the chunk that lands in slot tid&7
  const int skey = tid >> 3, sslot = tid & 7;
  const int kchunk = sslot ^ ((skey >> 1) & 7), vchunk = sslot ^ (((skey >> 1) & 1) << 2);
  auto stage = [&](int j0, int buf) {
    const int row = min(j0 + skey, len - 1);
    glds16(kbase + (size_t)row * ld + kchunk * 8, kvb + buf * 2 * BLK + wave * 1024);
    glds16(vbase + (size_t)row * ld + vchunk * 8, kvb + buf * 2 * BLK + BLK + wave * 1024);
  };
  // position rows of a key block: rp[rel_lo + 32 gb + l31], rel_lo = i0 - j0 - 31
  auto rp_row = [&](int j0, int gb) {
    const int row = min(max(i0 - j0 - 31 + gb * 32 + l31 + rp_zero, 0), rp_rows - 1);
    return rph + (size_t)row * d;
  };

  float m = -1e30f, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

  // V^T fragments: lane p of a 16-lane group feeds row (p>>2) of its [4 keys][16 dims] block and gets
  // the 4 keys of dim p; block of MFMA (db, u), half `part`: keys 16u + 4hi + 8 part + (0..3), dims
  // 32 db + 16 (l31>>4) + (0..15).  16u + 8 part never changes the swizzle bit: immediate offsets.
  unsigned vaddr[2];
  {
    const int p16 = lane & 15, g16 = (lane >> 4) & 1;
    const int key = 4 * hi + (p16 >> 2);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = db * 4 + g16 * 2 + ((p16 & 3) >> 1);
      vaddr[db] = (unsigned)(size_t)(kvb + BLK + key * 128 + ((chunk ^ (((key >> 1) & 1) << 2)) << 4) + (p16 & 1) * 8);
    }
  }
  // pad read addresses: query l31 needs G[rho][l31] for rho = l31 - jj + 31, jj = the key of accumulator register r.  Row rho
  // of an EVEN key block sits at rho * 128 B (rows 0..31 = this block's half, 32..63 = the previous block's); an odd block
  // swaps the halves, i.e. adds 4 KiB modulo 8 KiB.  With jj(r) = c_r + 4 hi, c_r = (r & 3) + 8 (r >> 2), every read is
  // (gbase + parity * 4096 - c_r * 128) mod 8192: ONE base register and compile-time offsets (round 4: the 16 per-register
  // offsets of round 1-3 cost 15 VGPRs of a kernel that sits one register class above 3 waves per SIMD).
  const int gbase = (l31 + 31 - 4 * hi) * 128 + l31 * 4;
  stage(0, 0);
  half8 rf[4];  // the position rows of the NEXT block travel through the softmax / PV half of this one
  {
    // The first key block needs BOTH pad halves (rho 32..63 have no previous block to come from): that half is computed
    // here, in a prologue, so that its four position-row registers are not live inside the loop (they were: `if (kb == 0)`
    // in the loop body kept 16 more VGPRs alive across every iteration's peak).  All eight 16-B loads are issued
    // together, ahead of the K / V wait.
    half8 rf_hi[4];
    const f16* rrow = rp_row(0, 0);
    const f16* rrow1 = rp_row(0, 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rf[ks] = *(const half8*)(rrow + (ks * 2 + hi) * 8);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rf_hi[ks] = *(const half8*)(rrow1 + (ks * 2 + hi) * 8);
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) g = __builtin_amdgcn_mfma_f32_32x32x16_f16(rf_hi[ks], qv[ks], g, 0, 0, 0);
    float* Gold0 = Gs + 32 * 32;  // block 0 reads rho 32..63 from half 1
#pragma unroll
    for (int r = 0; r < 16; ++r) Gold0[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = g[r];
  }

  for (int j0 = 0, kb = 0; j0 < len; j0 += RA_KB, ++kb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // block kb has landed for everyone; everyone is done with the other buffer
    const char* Ks = kvb + (kb & 1) * 2 * BLK;

    // ---- content term: S^T = K . (Q+u)^T ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const half8 kf = *(const half8*)(Ks + l31 * 128 + (((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4));
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qu[ks], s, 0, 0, 0);
    }
    // ---- position term: G[rho][i] = rp[rel_lo + rho] . (q_i + v), rho = 0..63 ----
    // rel_lo drops by 32 per key block, so rows 32..63 of this block are rows 0..31 of the previous
    // one: the pad is two 32-row halves used alternately, and only the first key block computes both.
    float* Gnew = Gs + (kb & 1) * 32 * 32;        // rho 0..31 of this block (rho 32..63 = the previous block's rho 0..31)
    {
      f32x16 g;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) g = __builtin_amdgcn_mfma_f32_32x32x16_f16(rf[ks], qv[ks], g, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) Gnew[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = g[r];
    }
    // next block: K / V by DMA into the other buffer, its position rows into rf
    if (j0 + RA_KB < len) {
      stage(j0 + RA_KB, (kb + 1) & 1);
      const f16* rrow = rp_row(j0 + RA_KB, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) rf[ks] = *(const half8*)(rrow + (ks * 2 + hi) * 8);
    }
    // (each wave reads back only what it wrote: no workgroup barrier needed, only LDS ordering)
    // the 16 pad reads are issued back to back and waited for once (left to itself hipcc sinks each
    // read into the branch of its `< len` mask: 16 exec-masked blocks, each with its own LDS wait)
    float bd[16];
    int gb = gbase + ((kb & 1) << 12);
    asm volatile("" : "+v"(gb));  // one live base: keep hipcc from materialising the 16 (or 32) addresses in registers
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bd[r] = *(const float*)((const char*)Gs + ((gb - ((r & 3) + 8 * (r >> 2)) * 128) & 8191));
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bd[r]));
    // x = content + position (unscaled); only the last key block of a clip has keys to mask
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] += bd[r];
    if (j0 + RA_KB > len) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (j0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= len) s[r] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    {  // the other half of the keys sits 32 lanes away: v_permlane32_swap (VALU) instead of __shfl_xor (an LDS round trip in
       // the middle of the softmax chain, once per key block)
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sl2e;  // sl2e > 0: the scaled maximum
    }
    // Lazy rescale: the running reference m only moves when some query's block maximum exceeds it by
    // more than 2^8; until then p = exp2(x - m) <= 256 (exact in fp32, in range for the fp16 P operand)
    // and the 32 output accumulators, which live in AGPRs, are left alone.
    if (__any(mx > m + 8.0f)) {
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      m = m_new;
      lsum *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    float psum = 0.f;
    half8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2e, -m));
      psum += p;
      pf[r >> 3][r & 7] = (f16)p;
    }
    lsum += psum;
    // ---- O^T += V^T . P^T ----  k slot e of MFMA u <-> key 16u + 4hi + 8(e>>2) + (e&3) (the S^T layout)
    half4 va[2][2][2];  // [db][u][part]
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const unsigned a = vaddr[db] + (kb & 1) * 2 * BLK;
      va[db][0][0] = ra_tr_read<0>(a);
      va[db][0][1] = ra_tr_read<1024>(a);
      va[db][1][0] = ra_tr_read<2048>(a);
      va[db][1][1] = ra_tr_read<3072>(a);
    }
    // the wait carries the 8 results as operands: the MFMAs below depend on IT, not just on the reads
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(va[0][0][0]), "+v"(va[0][0][1]), "+v"(va[0][1][0]), "+v"(va[0][1][1]), "+v"(va[1][0][0]),
                   "+v"(va[1][0][1]), "+v"(va[1][1][0]), "+v"(va[1][1][1])
                 :
                 : "memory");
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        half8 vf;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vf[e] = va[db][u][0][e];
          vf[4 + e] = va[db][u][1][e];
        }
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u], o[db], 0, 0, 0);
      }
  }
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = 1.0f / ltot;
  if (qi < len) {
    // ctx_tm: the context is the X operand of the output projection and is written in the tile-major layout of the
    // 256x256 engine (common.hpp: any packed row addresses its own 64-B slice of a 16 KiB block, no clip alignment needed)
    f16* op = ctx + (size_t)(start + qi) * d + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        half4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (f16)(o[db][q * 4 + e] * inv);
        const int col = db * 32 + 8 * q + 4 * hi;
        if (ctx_tm) *(half4*)(ctx + tm_offset(start + qi, h * 64 + col, d)) = v;
        else *(half4*)(op + col) = v;
      }
  }
}

// -DSMI_RPL_PAD32 (variant builds): fp32 score pad, 68 KiB of LDS = two workgroups per CU
#ifdef SMI_RPL_PAD32
typedef float RplPad;
#else
typedef f16 RplPad;
#endif
// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same kernel with the position rows staged ONCE per workgroup through LDS and the score pad in fp16.
// What bounded the kernel above (profiles/r06_experiments.txt, experiment 14): not its ~250 VALU instructions per key block and
// not exposed latency, but the throughput of its global loads -- every wave fetched its own 32 position rows per block as four
// global_load_dwordx4 that each touch 32 different 128-B lines (the MFMA A-operand layout wants lane = row): 128 line requests
// for 4 KiB, and every row is fetched by all four waves of the workgroup one block apart (a quarter of the kernel's time).
//  * The rows a workgroup needs at key block kb are the aligned 32-row blocks b = w - kb of rp[A + 32 b + (0..31)], A = q0 - 31 +
//    rp_zero, for its waves w = 0..3 (plus b = w + 1 for the first block's second pad half): from one key block to the next ONE
//    new block enters.  A ring of five 4 KiB slots (slot = b mod 5) holds them; the new block arrives by one coalesced LDS-DMA
//    instruction per thread (rows clamped to the table as before), in the K layout (row-major, 16-B chunk c of row r at c ^
//    ((r >> 1) & 7)), and every wave reads its A fragments with four conflict-free ds_read_b128: a quarter of the L2 -> CU bytes,
//    a sixteenth of the line requests.
//  * LDS: 16 KiB K / V + 20 KiB ring leave 16 KiB of the 52 KiB that let three workgroups share a CU, so the pad holds the
//    position scores in fp16 ([64 rho][32 queries] x 2 B per wave).  The reference's fp16 model rounds the position scores (and
//    the content scores) to fp16 before adding them (fairseq2 RelativePositionSDPA: two fp16 matmuls); here only the position
//    term takes that rounding, the sum and the softmax stay fp32.
// SMI_SPEECH_RP_LDS=0: the kernel above (A/B runs).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMI_RELPOS_WAVES, SMI_RELPOS_WAVES))) void relpos_attention_lds_kernel(
    const f16* __restrict__ qkv, const int32_t* __restrict__ cu, const f16* __restrict__ rp, int rp_zero, int rp_rows,
    const float* __restrict__ u_bias, const float* __restrict__ v_bias, f16* __restrict__ ctx, int d, float sl2e, int ctx_tm,
    int qkv_tm) {
  constexpr int BLK = RA_KB * 128;  // 4 KiB
  __shared__ __attribute__((aligned(16))) char lds[4 * BLK + 4 * BLK * (int)sizeof(RplPad) / 2 + 5 * BLK];  // K / V x 2 | fp16 pads | position-row ring
  const int n = blockIdx.x, h = blockIdx.y;
  const int start = cu[n], len = cu[n + 1] - start;
  const int q0 = blockIdx.z * RA_QB;
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  char* const kvb = lds;
  RplPad* Gs = (RplPad*)(lds + 4 * BLK) + wave * 64 * 32;  // [64 rho][32 queries]
  char* ring = lds + 4 * BLK + 4 * BLK * (int)sizeof(RplPad) / 2;
  const size_t ld = (size_t)3 * d;
  // qkv_tm: the fused QKV projection leaves q | k | v in the tile-major layout (common.hpp; K = 3 d), which lets that GEMM run on the
  // 4-wave engine: a 16-B chunk of a row is a 16-B chunk there too, the chunks of 8 consecutive rows of a 32-column block are 512
  // contiguous bytes, so the per-thread DMA sources below only change their address arithmetic
  auto src = [&](int row, int col) -> const f16* {  // 16-B chunk (start + row, col .. col + 7) of qkv
    return qkv_tm ? qkv + tm_offset(start + row, col, 3 * d) : qkv + (size_t)(start + row) * ld + col;
  };
  const f16* rph = rp + h * 64;

  const int i0 = q0 + wave * 32;
  const int qi = i0 + l31;
  half8 qu[4], qv[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c0 = (ks * 2 + hi) * 8;
    const half8 q = *(const half8*)src(min(qi, len - 1), h * 64 + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qu[ks][e] = (f16)((float)q[e] + u_bias[h * 64 + c0 + e]);
      qv[ks][e] = (f16)((float)q[e] + v_bias[h * 64 + c0 + e]);
    }
  }

  // DMA sources of this thread: row tid>>3 of a 32-row block, the chunk that lands in slot tid&7
  const int skey = tid >> 3, sslot = tid & 7;
  const int kchunk = sslot ^ ((skey >> 1) & 7), vchunk = sslot ^ (((skey >> 1) & 1) << 2);
  auto stage = [&](int j0, int buf) {
    const int row = min(j0 + skey, len - 1);
    glds16(src(row, d + h * 64 + kchunk * 8), kvb + buf * 2 * BLK + wave * 1024);
    glds16(src(row, 2 * d + h * 64 + vchunk * 8), kvb + buf * 2 * BLK + BLK + wave * 1024);
  };
  // position rows: aligned block b = rows A + 32 b + (0..31), clamped to the table, into ring slot `slot` (K layout)
  const int rowA = q0 - 31 + rp_zero + skey;
  auto stage_rp = [&](int b, int slot) {
    const int row = min(max(rowA + 32 * b, 0), rp_rows - 1);
    glds16(rph + (size_t)row * d + kchunk * 8, ring + slot * BLK + wave * 1024);
  };
  // A fragments of the block in ring slot `slot`: row l31, chunks (2 ks + hi).  asm: a C++ LDS load behind an LDS-DMA issue
  // gets `s_waitcnt vmcnt(0)` from hipcc while the next block's transfers are in flight
  const unsigned frag_off = (unsigned)(size_t)ring + l31 * 128;
  auto rp_frags = [&](int slot, half8 (&rf)[4]) {
    const unsigned a = frag_off + slot * BLK;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      asm volatile("ds_read_b128 %0, %1" : "=v"(rf[ks]) : "v"(a + (((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rf[0]), "+v"(rf[1]), "+v"(rf[2]), "+v"(rf[3]) : : "memory");
  };

  float m = -1e30f, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

  unsigned vaddr[2];
  {
    const int p16 = lane & 15, g16 = (lane >> 4) & 1;
    const int key = 4 * hi + (p16 >> 2);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = db * 4 + g16 * 2 + ((p16 & 3) >> 1);
      vaddr[db] = (unsigned)(size_t)(kvb + BLK + key * 128 + ((chunk ^ (((key >> 1) & 1) << 2)) << 4) + (p16 & 1) * 8);
    }
  }
  // pad reads: byte (gbase + parity * 2048 - c_r * 64) mod 4096 of the wave's pad (rows of 32 fp16), c_r = (r & 3) + 8 (r >> 2)
  [[maybe_unused]] const int gbase = (l31 + 31 - 4 * hi) * 64 + l31 * 2;
  // ... of an even block: rd_base + (27 - c_r) * 64, no wrap; of an odd block: the same xor 2048
  [[maybe_unused]] const unsigned rd_base = (unsigned)(size_t)Gs + (l31 + 31 - 4 * hi - 27) * 64 + l31 * 2;
  stage(0, 0);
#pragma unroll
  for (int b = 0; b < 5; ++b) stage_rp(b, b);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    // the first key block needs BOTH pad halves: rho 32..63 = block wave + 1
    half8 rf_hi[4];
    rp_frags(wave + 1, rf_hi);
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) g = __builtin_amdgcn_mfma_f32_32x32x16_f16(rf_hi[ks], qv[ks], g, 0, 0, 0);
    RplPad* Gold0 = Gs + 32 * 32;  // block 0 reads rho 32..63 from half 1
#pragma unroll
    for (int r = 0; r < 16; ++r) Gold0[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = (RplPad)g[r];
  }

  int rslot = wave;  // ring slot of block (wave - kb)
  int dslot = 4;     // ring slot of block -(kb + 1), the one that arrives during key block kb
  // one 32-key block; P = parity of the block index (which K / V buffer, which pad half is the new one): the loop is unrolled by
  // two so that the pad addresses of an even block are one base + compile-time offsets and those of an odd block one xor away
  // (+2 KiB modulo the 4 KiB pad), the exponent arguments come from v_pk_fma_f32 and the row sum from a tree of packed adds
  // (experiment 14: 16 adds + 16 ands + 16 adds per block for the wrapped reads, 16 + 16 scalar VALU operations for the rest)
  auto block = [&](auto ptag, int j0) {
    constexpr int kb = decltype(ptag)::value;  // only its parity is used below
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // block kb (K, V, its position rows) has landed for everyone; everyone is done with the other buffers
    const char* Ks = kvb + (kb & 1) * 2 * BLK;

    // ---- content term: S^T = K . (Q+u)^T ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
      half8 kf[4];
      const unsigned ka = (unsigned)(size_t)Ks + l31 * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[ks]) : "v"(ka + (((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]) : : "memory");
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qu[ks], s, 0, 0, 0);
    }
    // ---- position term: G[rho][i] = rp[rel_lo + rho] . (q_i + v), rho = 0..31 new, 32..63 = the previous block's ----
    RplPad* Gnew = Gs + (kb & 1) * 32 * 32;
    {
      half8 rf[4];
      rp_frags(rslot, rf);
      f32x16 g;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) g = __builtin_amdgcn_mfma_f32_32x32x16_f16(rf[ks], qv[ks], g, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) Gnew[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = (RplPad)g[r];
    }
    // next block: K / V into the other buffer, the one new block of position rows into the slot that just fell out of use
    if (j0 + RA_KB < len) {
      stage(j0 + RA_KB, (kb + 1) & 1);
      stage_rp(-(j0 / RA_KB + 1), dslot);
    }
    rslot = rslot ? rslot - 1 : 4;
    dslot = dslot ? dslot - 1 : 4;
    float bd[16];
#ifdef SMI_RPL_PAD32
    {
      int gb = 2 * gbase + ((kb & 1) << 12);
      asm volatile("" : "+v"(gb));
#pragma unroll
      for (int r = 0; r < 16; ++r) bd[r] = *(const float*)((const char*)Gs + ((gb - ((r & 3) + 8 * (r >> 2)) * 128) & 8191));
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bd[r]));
    }
#else
    {
      // 32-bit destinations: ds_read_u16 zero-extends itself.  With 16-bit asm outputs hipcc masks every result with 0xffff right
      // behind its load -- in front of the wait below, i.e. on a register the data has not reached yet (stale values whenever a
      // second workgroup on the CU delays the LDS: the run-to-run differences this kernel had at first).
      unsigned raw[16];
      unsigned rb = rd_base;
      asm volatile("" : "+v"(rb));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr ((kb & 1) == 0) {
          asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(raw[r]) : "v"(rb), "n"((27 - ((r & 3) + 8 * (r >> 2))) * 64) : "memory");
        } else {
          const unsigned a = (rb + (27 - ((r & 3) + 8 * (r >> 2))) * 64) ^ 2048u;
          asm volatile("ds_read_u16 %0, %1" : "=v"(raw[r]) : "v"(a) : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]),
                     "+v"(raw[8]), "+v"(raw[9]), "+v"(raw[10]), "+v"(raw[11]), "+v"(raw[12]), "+v"(raw[13]), "+v"(raw[14]),
                     "+v"(raw[15])
                   :
                   : "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) bd[r] = (float)__builtin_bit_cast(f16, (unsigned short)raw[r]);
    }
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] += bd[r];
    if (j0 + RA_KB > len) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (j0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= len) s[r] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sl2e;
    }
    if (__any(mx > m + 8.0f)) {  // lazy rescale (see the kernel above)
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      m = m_new;
      lsum *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    f32x2 ps2[8];
    half8 pf[2];
    const f32x2 sc2 = {sl2e, sl2e}, nm2 = {-m, -m};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 a = __builtin_elementwise_fma(f32x2{s[r], s[r + 1]}, sc2, nm2);
      const f32x2 pp = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
      ps2[r >> 1] = pp;
      pf[r >> 3][r & 7] = (f16)pp[0];
      pf[r >> 3][(r & 7) + 1] = (f16)pp[1];
    }
    {
      const f32x2 t0 = (ps2[0] + ps2[1]) + (ps2[2] + ps2[3]), t1 = (ps2[4] + ps2[5]) + (ps2[6] + ps2[7]);
      const f32x2 t = t0 + t1;
      lsum += t[0] + t[1];
    }
    half4 va[2][2][2];  // [db][u][part]
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const unsigned a = vaddr[db] + (kb & 1) * 2 * BLK;
      va[db][0][0] = ra_tr_read<0>(a);
      va[db][0][1] = ra_tr_read<1024>(a);
      va[db][1][0] = ra_tr_read<2048>(a);
      va[db][1][1] = ra_tr_read<3072>(a);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(va[0][0][0]), "+v"(va[0][0][1]), "+v"(va[0][1][0]), "+v"(va[0][1][1]), "+v"(va[1][0][0]),
                   "+v"(va[1][0][1]), "+v"(va[1][1][0]), "+v"(va[1][1][1])
                 :
                 : "memory");
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        half8 vf;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vf[e] = va[db][u][0][e];
          vf[4 + e] = va[db][u][1][e];
        }
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u], o[db], 0, 0, 0);
      }
  };
  for (int j0 = 0;;) {
    block(std::integral_constant<int, 0>{}, j0);
    j0 += RA_KB;
    if (j0 >= len) break;
    block(std::integral_constant<int, 1>{}, j0);
    j0 += RA_KB;
    if (j0 >= len) break;
  }
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = 1.0f / ltot;
  if (qi < len) {
    f16* op = ctx + (size_t)(start + qi) * d + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        half4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (f16)(o[db][q * 4 + e] * inv);
        const int col = db * 32 + 8 * q + 4 * hi;
        if (ctx_tm) *(half4*)(ctx + tm_offset(start + qi, h * 64 + col, d)) = v;
        else *(half4*)(op + col) = v;
      }
  }
}

bool relpos_attention_reads_tile_major() { return tune(TUNE_SPEECH_RP_LDS, 1) != 0 && tune(TUNE_SPEECH_QKV_TM, 1) != 0; }

hipError_t launch_relpos_attention(const f16* qkv, const int32_t* cu, const f16* rp, int rp_zero, int rp_rows,
                                   const float* u_bias, const float* v_bias, f16* ctx, int n, int max_len, int d,
                                   int heads, hipStream_t stream, int ctx_tm, int qkv_tm) {
  if (heads <= 0 || d != heads * 64 || n <= 0 || max_len <= 0) return hipErrorInvalidValue;
  if (qkv_tm && tune(TUNE_SPEECH_RP_LDS, 1) == 0) return hipErrorInvalidValue;  // only the LDS-ring kernel reads tile-major q | k | v
  const float sl2e = 0.125f * 1.4426950408889634f;
  dim3 grid(n, heads, (max_len + RA_QB - 1) / RA_QB);
  if (tune(TUNE_SPEECH_RP_LDS, 1) != 0)
    hipLaunchKernelGGL(relpos_attention_lds_kernel, grid, dim3(256), 0, stream, qkv, cu, rp, rp_zero, rp_rows, u_bias, v_bias,
                       ctx, d, sl2e, ctx_tm, qkv_tm);
  else
    hipLaunchKernelGGL(relpos_attention_kernel, grid, dim3(256), 0, stream, qkv, cu, rp, rp_zero, rp_rows, u_bias,
                       v_bias, ctx, d, sl2e, ctx_tm);
  return hipGetLastError();
}

// ============================================= depthwise conv (k taps, 'same') + BatchNorm + SiLU
// y[t][c] = silu(scale[c] * sum_k w[c][k] x[t + k - (K-1)/2][c] + shift[c]), zero outside the clip.
// Workgroup = 32 output frames x 256 channels of one clip.  The 32 + (K-1) input rows are staged in LDS
// with 16-B loads (a 512-B row per 32 lanes) and the outputs leave through LDS with 16-B stores; in
// between thread = channel reads its column (2 B per lane, conflict-free) once into registers, so every
// tap index is a compile-time constant.  (The first version loaded and stored 2 B per lane straight
// from / to global memory: 126 us per call for 131 MB.)
template <int KT>
__global__ __launch_bounds__(256) void dwconv_bn_silu_kernel(const f16* __restrict__ x,
                                                             const int32_t* __restrict__ cu,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             f16* __restrict__ y, int d, int y_tm, int x_tm) {
  constexpr int TT = 32, HALF = (KT - 1) / 2, NIN = TT + KT - 1;
  __shared__ __attribute__((aligned(16))) f16 xin[NIN][256];
  __shared__ __attribute__((aligned(16))) f16 yout[TT][256];
  const int n = blockIdx.x, t0 = blockIdx.y * TT, c0 = blockIdx.z * 256, tid = threadIdx.x;
  const int start = cu[n], len = cu[n + 1] - start;
  if (t0 >= len) return;
  for (int i = tid; i < NIN * 32; i += 256) {
    const int r = i >> 5, ch = (i & 31) * 8;
    const int t = t0 - HALF + r;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};  // zero outside the clip
    if (t >= 0 && t < len)  // x_tm: the GLU output arrives tile-major (a 16-B chunk of a row is a 16-B chunk there too)
      v = *(const half8*)(x_tm ? x + tm_offset(start + t, c0 + ch, d) : x + (size_t)(start + t) * d + c0 + ch);
    *(half8*)&xin[r][ch] = v;
  }
  const int c = c0 + tid;
  // Two taps per instruction (round 6: the kernel is VALU-bound -- 32 x 31 fp32 FMAs per thread were ~80 % of its time): the
  // window is held as fp16 PAIRS of consecutive frames, once starting at even and once at odd offsets (62 registers, as many as
  // the fp32 window took), the taps as fp16 pairs (the fp16 model's conv weights ARE fp16; an fp32 model's are rounded like every
  // GEMM weight of the engine), and v_dot2_f32_f16 forms a0 b0 + a1 b1 + c with exact products and fp32 accumulation.
  constexpr int NP = (KT + 1) / 2;
  half2v wk2[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const float w0 = w[(size_t)c * KT + 2 * j], w1 = 2 * j + 1 < KT ? w[(size_t)c * KT + 2 * j + 1] : 0.f;
    wk2[j] = half2v{(f16)w0, (f16)w1};
  }
  const float sc = scale[c], sh = shift[c];
  __syncthreads();
  constexpr int NW = NIN + 1;  // one padding frame so that the last odd pair exists (its tap is the zero pad of an odd KT)
  half2v pe[NW / 2], po[NW / 2];  // pe[i] = (x[2i], x[2i+1]), po[i] = (x[2i+1], x[2i+2])
  {
    f16 prev = xin[0][tid];
#pragma unroll
    for (int i = 0; i < NW / 2; ++i) {
      const f16 a = xin[2 * i + 1][tid];
      const f16 b = 2 * i + 2 < NIN ? xin[2 * i + 2][tid] : (f16)0.f;
      pe[i] = half2v{prev, a};
      po[i] = half2v{a, b};
      prev = b;
    }
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j)  // frames tt + 2j, tt + 2j + 1
      acc = __builtin_amdgcn_fdot2((tt & 1) ? po[(tt + 2 * j) / 2] : pe[(tt + 2 * j) / 2], wk2[j], acc, false);
    const float v = acc * sc + sh;
    yout[tt][tid] = (f16)(v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)));  // SiLU
  }
  __syncthreads();
  for (int i = tid; i < TT * 32; i += 256) {
    const int tt = i >> 5, ch = (i & 31) * 8;
    if (t0 + tt < len) {  // y_tm: tile-major X operand of pointwise_conv2 (a 16-B chunk of a row is a 16-B chunk there too)
      f16* dst = y_tm ? y + tm_offset(start + t0 + tt, c0 + ch, d) : y + (size_t)(start + t0 + tt) * d + c0 + ch;
      *(half8*)dst = *(const half8*)&yout[tt][ch];
    }
  }
}

hipError_t launch_dwconv_bn_silu(const f16* x, const int32_t* cu, const float* w, const float* scale,
                                 const float* shift, f16* y, int n, int max_len, int d, int ktaps,
                                 hipStream_t stream, int y_tm, int x_tm) {
  if (d % 256 || n <= 0 || max_len <= 0) return hipErrorInvalidValue;
  dim3 grid(n, (max_len + 31) / 32, d / 256);
  switch (ktaps) {
    case 31:
      hipLaunchKernelGGL(dwconv_bn_silu_kernel<31>, grid, dim3(256), 0, stream, x, cu, w, scale, shift, y, d, y_tm, x_tm);
      break;
    case 7:
      hipLaunchKernelGGL(dwconv_bn_silu_kernel<7>, grid, dim3(256), 0, stream, x, cu, w, scale, shift, y, d, y_tm, x_tm);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ================================================== pooler: one query per clip over its frames
// q: [n][d] (row per clip), kv: [R][2d] (k | v) packed frames; one wave per (clip, head).
__global__ __launch_bounds__(256) void pool_attention_kernel(const f16* __restrict__ q,
                                                             const f16* __restrict__ kv,
                                                             const int32_t* __restrict__ cu,
                                                             f16* __restrict__ ctx, int n, int d, int heads,
                                                             float sl2e) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= n * heads) return;
  const int c = wid / heads, h = wid % heads;
  const int lane = threadIdx.x & 63;
  const int start = cu[c], len = cu[c + 1] - start;
  const f16* qp = q + (size_t)c * d + h * 64;
  half8 qf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) qf[i] = *(const half8*)(qp + i * 8);
  const size_t ld = (size_t)2 * d;
  const f16* kb = kv + (size_t)start * ld + h * 64;
  // P.V: lane (pg, c8) accumulates head dims c8*8..+7 over the positions j0 + 8 it + pg (16-B row pieces,
  // the probability comes from the lane that scored that position), joined across pg at the end
  const int pg = lane >> 3, c8 = lane & 7;
  float m = -1e30f, l = 0.f;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int j0 = 0; j0 < len; j0 += 64) {
    const int j = j0 + lane;
    float s = -INFINITY;
    if (j < len) {
      const f16* kp = kb + (size_t)j * ld;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const half8 kk = *(const half8*)(kp + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)qf[i][e] * (float)kk[e];
      }
      s = acc * sl2e;
    }
    const float m_new = fmaxf(m, wave_max(s));
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    const float p = __builtin_amdgcn_exp2f(s - m_new);
    l = l * alpha + wave_sum(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= alpha;
    m = m_new;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int t = it * 8 + pg;
      const float pt = __shfl(p, t, 64);  // 0 for positions past the clip
      if (j0 + t < len) {
        const half8 vv = *(const half8*)(kb + (size_t)(j0 + t) * ld + d + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += pt * (float)vv[e];
      }
    }
  }
  half8 out;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = o[e];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    out[e] = (f16)(len > 0 ? v / l : 0.f);
  }
  if (pg == 0) *(half8*)(ctx + (size_t)c * d + h * 64 + c8 * 8) = out;
}

hipError_t launch_pool_attention(const f16* q, const f16* kv, const int32_t* cu, f16* ctx, int n, int d, int heads,
                                 hipStream_t stream) {
  if (heads <= 0 || d != heads * 64 || n <= 0) return hipErrorInvalidValue;
  const float sl2e = 0.125f * 1.4426950408889634f;
  hipLaunchKernelGGL(pool_attention_kernel, dim3((n * heads + 3) / 4), dim3(256), 0, stream, q, kv, cu, ctx, n, d,
                     heads, sl2e);
  return hipGetLastError();
}

// x[r, :] = row[:] for r < rows (pooler query initialisation)
__global__ void broadcast_row_kernel(const float* __restrict__ row, float* __restrict__ x, int rows, int d) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)rows * d) x[i] = row[i % d];
}

hipError_t launch_broadcast_row(const float* row, float* x, int rows, int d, hipStream_t stream) {
  const size_t total = (size_t)rows * d;
  hipLaunchKernelGGL(broadcast_row_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, row, x, rows, d);
  return hipGetLastError();
}

}  // namespace smi
