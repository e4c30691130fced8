// Sampling step of the EmbeddingToText generator: top-k / top-p (nucleus) filtering of one softmax row
// and the multinomial draw, one workgroup per row, without sorting the vocabulary.
//   sonar/inference_pipelines/text.py:315-320 builds fairseq2's SamplingSeq2SeqGenerator with a
//   TopKSampler / TopPSampler; their published behaviour (fairseq2 ~= 0.4, un-vendored):
//     probs = softmax(logits / temperature, fp32); probs[pad] = 0; probs[eos] = 0 before min_len;
//     probs[unk] -= unk_penalty;
//     top-p: sort descending, keep rank r while (cumsum - prob)[r] <= p; top-k: keep the k largest;
//     renormalise the kept set, draw one token; step score = log(probs[token]) (not renormalised).
// Arithmetic: a token's mass is exp(l/T - M) in Q40 fixed point (u64), so every sum is an integer and
// independent of the order LDS atomics arrive in: the filter and the draw are bit-reproducible.  The
// threshold "largest sorted prefix whose exclusive mass (count) stays <= p * Z (k - 1)" is found by an
// 11/11/10-bit radix descent over the order-preserving key of the logit (3 histogram passes over the
// 1 MB row), value ties at the threshold are kept lowest token id first, and the draw walks the kept
// mass in a fixed (thread-major) order with the integer target floor(z * kept / 2^64).
#include <cstdint>

#include "api_common.hpp"
#include "common.hpp"
#include "kernels.hpp"

using namespace smi;
using namespace smi_host;

namespace smi {

typedef unsigned long long u64;

constexpr int SMP_THREADS = 1024;  // one workgroup per row; 16 waves keep enough 16-B loads in flight
constexpr int SMP_WAVES = SMP_THREADS / 64;
constexpr int SMP_BUCKETS = 2048;
constexpr int SMP_PER_THREAD = SMP_BUCKETS / SMP_THREADS;  // 2

__device__ __forceinline__ uint32_t smp_key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// Q40 mass of a scaled logit t = l / T - M (t <= 0)
__device__ __forceinline__ u64 smp_mass(float t) {
  return (u64)(__builtin_amdgcn_exp2f(t * 1.4426950408889634f) * 1099511627776.0f);
}

__device__ __forceinline__ u64 shfl_up_u64(u64 v, int delta) {
  const uint32_t lo = __shfl_up((uint32_t)v, delta, 64), hi = __shfl_up((uint32_t)(v >> 32), delta, 64);
  return ((u64)hi << 32) | lo;
}
// exclusive prefix sum over the workgroup's threads (thread order); returns the total through *total
__device__ u64 block_scan_excl(u64 v, u64* s_w /* [SMP_WAVES] */, u64* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  u64 inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u64 t = shfl_up_u64(inc, o);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  u64 base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SMP_WAVES; ++w) {
    if (w < wv) base += s_w[w];
    tot += s_w[w];
  }
  *total = tot;
  return base + inc - v;
}
__device__ int block_max_int(int v, int* s_i /* [SMP_WAVES] */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if (lane == 0) s_i[wv] = v;
  __syncthreads();
  int r = s_i[0];
#pragma unroll
  for (int w = 1; w < SMP_WAVES; ++w) r = max(r, s_i[w]);
  return r;
}

__device__ __forceinline__ u64 smp_hash(u64 seed, int row, int step) {
  u64 z = seed + 0x9E3779B97F4A7C15ull * ((u64)row * 65536ull + (u64)step + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(SMP_THREADS) void sample_rows_kernel(SampleRowsArgs a) {
  __shared__ u64 h_mass[SMP_BUCKETS];
  __shared__ uint32_t h_cnt[SMP_BUCKETS];
  __shared__ u64 s_w[SMP_WAVES];
  __shared__ int s_i[SMP_WAVES];
  __shared__ float s_f[SMP_WAVES];
  __shared__ u64 s_bc[4];  // broadcast slots
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (a.done && a.done[row]) return;
  const float* lg = a.logits + (size_t)row * a.ld;
  const int V = a.vocab;
  const int niter = (V + SMP_THREADS * 4 - 1) / (SMP_THREADS * 4);
  const float it = a.inv_temp;
  // UNK penalty (probs[unk] -= unk_penalty, in the Q40 domain: mass[unk] -= floor(penalty * Z)): the token keeps an exact
  // integer mass w_unk and takes the rank of a logit with that mass; a mass <= 0 leaves the kept set like pad.
  const bool pen = a.unk_penalty != 0.f && a.forced_tok < 0 && a.unk_idx >= 0 && a.unk_idx < V;
  bool unk_dead = false;
  u64 w_unk = 0;
  uint32_t key_unk = 0;
  auto masked = [&](int idx) {
    return idx == a.pad_idx || (a.block_eos && idx == a.eos_idx) || (unk_dead && idx == a.unk_idx);
  };
  // element visitor: f(idx, raw logit) for every idx < V this thread owns (iteration-major, coalesced)
  auto for_each = [&](auto&& f) {
    for (int i0 = 0; i0 < niter; i0 += 4) {  // four 16-B loads in flight per thread
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e0 = ((i0 + u) * SMP_THREADS + tid) * 4;
        v[u] = e0 < V ? *(const f32x4*)(lg + e0) : f32x4{0.f, 0.f, 0.f, 0.f};  // rows are padded: readable
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e0 = ((i0 + u) * SMP_THREADS + tid) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (e0 + c < V) f(e0 + c, v[u][c]);
      }
    }
  };

  // ---- A. M = max of the scaled logits (over every token: the softmax normaliser includes the masked ones)
  float m = -INFINITY;
  for_each([&](int, float v) { m = fmaxf(m, v * it); });
  m = wave_max(m);
  if (lane == 0) s_f[wv] = m;
  __syncthreads();
  float M = s_f[0];
#pragma unroll
  for (int w = 1; w < SMP_WAVES; ++w) M = fmaxf(M, s_f[w]);

  u64 z_orig = 0;
  if (pen) {  // Z of the untouched distribution first (one more pass over the row, this mode only)
    u64 zloc = 0;
    for_each([&](int, float v) { zloc += smp_mass(v * it - M); });
    (void)block_scan_excl(zloc, s_w, &z_orig);
    const u64 w0 = smp_mass(lg[a.unk_idx] * it - M);
    const double d = (double)a.unk_penalty * (double)z_orig;
    const long long delta = (long long)d;  // toward zero, as the oracle's int()
    const long long w1 = (long long)w0 - delta;
    unk_dead = w1 <= 0;
    w_unk = unk_dead ? 0 : (u64)w1;
    // rank: the logit whose softmax mass is w_unk (approximate; only its ORDER among the other tokens is used)
    key_unk = smp_key((__logf((float)w_unk * (1.0f / 1099511627776.0f)) + M) / it);
    __syncthreads();
  }
  auto mass_of = [&](int idx, float v) { return (pen && idx == a.unk_idx) ? w_unk : smp_mass(v * it - M); };
  auto key_of = [&](int idx, float v) { return (pen && idx == a.unk_idx) ? key_unk : smp_key(v); };

  if (a.forced_tok >= 0) {  // prompt forcing / forced EOS: only the step score is needed
    u64 zloc = 0, zf;
    for_each([&](int, float v) { zloc += smp_mass(v * it - M); });
    (void)block_scan_excl(zloc, s_w, &zf);
    if (tid == 0) {
      a.out_tok[row] = a.forced_tok;
      a.out_logp[row] = (lg[a.forced_tok] * it - M) - __logf((float)zf * (1.0f / 1099511627776.0f));
    }
    return;
  }

  // ---- B-D. radix descent for the threshold key
  const bool topk = a.mode == 0;
  u64 above_mass = 0, above_cnt = 0, zfull = 0, T = 0;
  uint32_t prefix = 0;  // the key bits fixed so far, right-aligned
  u64 tie_cnt = 0;
  for (int level = 0; level < 3; ++level) {
    const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
    const int bits = level == 2 ? 10 : 11;
    for (int b = tid; b < SMP_BUCKETS; b += SMP_THREADS) {
      h_mass[b] = 0;
      h_cnt[b] = 0;
    }
    __syncthreads();
    u64 zloc = 0;
    for_each([&](int idx, float v) {
      if (level == 0) zloc += smp_mass(v * it - M);  // Z: the untouched distribution (the reference does not renormalise)
      if (masked(idx)) return;
      const u64 w = mass_of(idx, v);
      const uint32_t key = key_of(idx, v);
      if (level > 0 && (key >> (shift + bits)) != prefix) return;
      const int b = (key >> shift) & ((1 << bits) - 1);
      atomicAdd(&h_mass[b], w);
      atomicAdd(&h_cnt[b], 1u);
    });
    __syncthreads();
    if (level == 0) {
      u64 tot;
      (void)block_scan_excl(zloc, s_w, &tot);
      zfull = tot;
      // exclusive-mass budget: p * Z (top-p), k - 1 tokens (top-k)
      T = topk ? (u64)(a.top_k - 1) : (u64)((double)a.top_p * (double)zfull);
    }
    // reversed bucket order: rb = nb - 1 - b, thread t owns rb in [8t, 8t + 8)
    const int nb = 1 << bits;
    u64 lm[SMP_PER_THREAD], lc[SMP_PER_THREAD], sm = 0, sc = 0;
#pragma unroll
    for (int j = 0; j < SMP_PER_THREAD; ++j) {
      const int rb = tid * SMP_PER_THREAD + j;
      const bool in = rb < nb;
      lm[j] = in ? h_mass[nb - 1 - rb] : 0;
      lc[j] = in ? h_cnt[nb - 1 - rb] : 0;
      sm += lm[j];
      sc += lc[j];
    }
    u64 tot;
    u64 em = block_scan_excl(sm, s_w, &tot);
    u64 ec = block_scan_excl(sc, s_w, &tot);
    int best = -1;
    u64 bm = 0, bcn = 0;
#pragma unroll
    for (int j = 0; j < SMP_PER_THREAD; ++j) {
      const u64 first = topk ? above_cnt + ec : above_mass + em;  // exclusive weight of the bucket's first token
      if (lc[j] > 0 && first <= T) {
        best = tid * SMP_PER_THREAD + j;
        bm = em;
        bcn = ec;
      }
      em += lm[j];
      ec += lc[j];
    }
    const int rbs = block_max_int(best, s_i);  // >= 0: the first token of the range always fits
    if (best == rbs) {
      s_bc[0] = bm;
      s_bc[1] = bcn;
      s_bc[2] = h_cnt[nb - 1 - rbs];
    }
    __syncthreads();
    above_mass += s_bc[0];
    above_cnt += s_bc[1];
    tie_cnt = s_bc[2];
    prefix = (prefix << bits) | (uint32_t)(nb - 1 - rbs);
    __syncthreads();
  }
  const uint32_t kstar = prefix;  // full 32-bit key of the threshold value; tie_cnt tokens carry it
  // mass of one threshold token (every tie has the same scaled logit bits or at least the same key
  // order; take it from the histogram to stay consistent with the sums)
  const u64 wstar = tie_cnt ? h_mass[kstar & 1023] / tie_cnt : 0;
  u64 ckeep;
  if (topk)
    ckeep = min(tie_cnt, T - above_cnt + 1);
  else
    ckeep = wstar ? min(tie_cnt, (T - above_mass) / wstar + 1) : tie_cnt;
  __syncthreads();

  // ---- E. ties at the threshold: keep the ckeep lowest token ids (two 9-bit levels over the ids)
  int id_thr = 0x7fffffff;
  u64 tie_mass_kept = h_mass[kstar & 1023];  // all ties kept
  __syncthreads();
  if (ckeep < tie_cnt) {
    uint32_t idp = 0;
    u64 need = ckeep;  // the need-th smallest id (1-based) among the ties
    u64 mass_lo = 0;
    for (int level = 0; level < 2; ++level) {
      const int shift = level == 0 ? 9 : 0;
      for (int b = tid; b < 512; b += SMP_THREADS) {
        h_cnt[b] = 0;
        h_mass[1024 + b] = 0;
      }
      __syncthreads();
      for_each([&](int idx, float v) {
        if (masked(idx) || key_of(idx, v) != kstar) return;
        if (level == 1 && (uint32_t)(idx >> 9) != idp) return;
        atomicAdd(&h_cnt[(idx >> shift) & 511], 1u);
        atomicAdd(&h_mass[1024 + ((idx >> shift) & 511)], mass_of(idx, v));
      });
      __syncthreads();
      // thread t < 512 owns bucket t (ascending ids)
      const u64 c0 = tid < 512 ? h_cnt[tid] : 0;
      u64 tot;
      const u64 e = block_scan_excl(c0, s_w, &tot);
      const u64 em = block_scan_excl(tid < 512 ? h_mass[1024 + tid] : 0, s_w, &tot);
      const int hit = (c0 && e < need && need <= e + c0) ? tid : -1;
      const int bsel = block_max_int(hit, s_i);
      if (hit == bsel) {
        s_bc[0] = e;   // ties in lower buckets
        s_bc[1] = em;  // their mass
      }
      __syncthreads();
      need -= s_bc[0];
      mass_lo += s_bc[1];
      idp = (idp << 9) | (uint32_t)bsel;
      __syncthreads();
    }
    id_thr = (int)idp;  // need == 1 now: ids are unique
    // mass of the kept ties = ties below id_thr + the one at id_thr
    tie_mass_kept = mass_lo + h_mass[1024 + (idp & 511)];
    __syncthreads();
  }
  const u64 kept_mass = above_mass + tie_mass_kept;
  const u64 kept_cnt = above_cnt + ckeep;
  if (tid == 0) {
    if (a.out_kept_mass) a.out_kept_mass[row] = kept_mass;
    if (a.out_kept_count) a.out_kept_count[row] = (int32_t)kept_cnt;
  }
  const float logz = __logf((float)zfull * (1.0f / 1099511627776.0f));

  // ---- F. draw: integer target in [0, kept_mass); order = thread-major over the coalesced ownership
  auto kept = [&](int idx, float v) {
    if (masked(idx)) return false;
    const uint32_t key = key_of(idx, v);
    return key > kstar || (key == kstar && idx <= id_thr);
  };
  const u64 zr = a.z ? a.z[row] : smp_hash(a.seed, row, a.step);
  const u64 target = __umul64hi(zr, kept_mass);
  u64 mine = 0;
  for_each([&](int idx, float v) {
    if (kept(idx, v)) mine += mass_of(idx, v);
  });
  u64 tot;
  const u64 ex = block_scan_excl(mine, s_w, &tot);
  int owner = (mine > 0 && ex <= target && target < ex + mine) ? tid : -1;
  owner = block_max_int(owner, s_i);
  if (tid == owner) s_bc[0] = ex;
  __syncthreads();
  u64 base = s_bc[0];
  __syncthreads();
  // ---- G. inside the owner's elements (iteration order), all threads cooperate: thread j takes
  // iteration j, j + 256, ...
  for (int i0 = 0; i0 < niter; i0 += SMP_THREADS) {
    const int i = i0 + tid;
    const int e0 = (i * SMP_THREADS + owner) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    u64 w4[4] = {0, 0, 0, 0}, wsum = 0;
    if (i < niter && e0 < V) {
      v = *(const f32x4*)(lg + e0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (e0 + c < V && kept(e0 + c, v[c])) {
          w4[c] = mass_of(e0 + c, v[c]);
          wsum += w4[c];
        }
    }
    const u64 exi = block_scan_excl(wsum, s_w, &tot);
    if (wsum > 0 && base + exi <= target && target < base + exi + wsum) {
      u64 run = base + exi;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (w4[c] > 0 && run <= target && target < run + w4[c]) {
          a.out_tok[row] = e0 + c;
          a.out_logp[row] = (pen && e0 + c == a.unk_idx) ? __logf((float)w_unk * (1.0f / 1099511627776.0f)) - logz
                                                        : (v[c] * it - M) - logz;
        }
        run += w4[c];
      }
    }
    base += tot;
    if (base > target) break;  // uniform: found in this round
  }
}

// generation state after one sampling step: append the token, accumulate the score, retire rows at EOS
__global__ void sample_update_kernel(SampleUpdateArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n || a.done[r]) return;
  const int t = a.samp_tok[r];
  const int step_nr = a.pos + 1;  // index of the token just produced
  const float cum = a.cum[r] + a.samp_logp[r];
  a.cum[r] = cum;
  a.tok[r] = t;
  if (step_nr < a.prompt_len) return;
  a.out_tokens[(size_t)r * a.out_stride + (step_nr - a.prompt_len)] = t;
  if (t == a.eos_idx) {
    a.done[r] = 1;
    atomicAdd(a.ndone, 1);
    a.out_lens[r] = step_nr - a.prompt_len + 1;
    a.out_scores[r] = a.normalize ? cum / powf((float)step_nr, a.len_penalty) : cum;
  }
}

hipError_t launch_sample_update(const SampleUpdateArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(sample_update_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_sample_rows(const SampleRowsArgs& a, hipStream_t stream) {
  if (a.rows <= 0 || a.vocab <= 0 || !(a.inv_temp > 0.f) || a.ld % 4 || a.ld < (a.vocab + 3) / 4 * 4)
    return hipErrorInvalidValue;
  if (a.vocab > (1 << 18)) return hipErrorInvalidValue;  // the tie-break descent covers 18-bit token ids
  if (a.forced_tok < 0) {
    if (a.mode == 0 && a.top_k < 1) return hipErrorInvalidValue;
    if (a.mode == 1 && !(a.top_p > 0.f && a.top_p <= 1.f)) return hipErrorInvalidValue;
    if (a.mode != 0 && a.mode != 1) return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(sample_rows_kernel, dim3(a.rows), dim3(SMP_THREADS), 0, stream, a);
  return hipGetLastError();
}

}  // namespace smi

extern "C" int smi_sample_rows(const float* logits, int64_t ld, int32_t rows, int32_t vocab, int32_t sampler,
                               int32_t top_k, float top_p, float temperature, int32_t pad_idx, int32_t eos_idx,
                               int32_t block_eos, int32_t unk_idx, float unk_penalty, const uint64_t* z,
                               int32_t* out_token, float* out_logprob, uint64_t* out_kept_mass, int32_t* out_kept_count,
                               void* stream) {
  if (!logits || !z || !out_token || !out_logprob) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (rows <= 0 || vocab <= 0) return fail(SMI_ERR_INVALID_ARG, "empty input");
  if (vocab > (1 << 18)) return fail(SMI_ERR_UNSUPPORTED, "vocab %d: sampling covers up to 2^18 tokens", vocab);
  if (ld % 4 || ld < (vocab + 3) / 4 * 4) return fail(SMI_ERR_INVALID_ARG, "ld %lld must be a multiple of 4 >= vocab", (long long)ld);
  if (!(temperature > 0.f)) return fail(SMI_ERR_INVALID_ARG, "temperature must be positive");
  if (sampler == SMI_SAMPLER_TOP_K) {
    if (top_k < 1) return fail(SMI_ERR_INVALID_ARG, "top_k must be >= 1");
  } else if (sampler == SMI_SAMPLER_TOP_P) {
    if (!(top_p > 0.f && top_p <= 1.f)) return fail(SMI_ERR_INVALID_ARG, "top_p must be in (0, 1]");
  } else {
    return fail(SMI_ERR_INVALID_ARG, "unknown sampler %d", sampler);
  }
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  SampleRowsArgs a{};
  a.logits = logits; a.ld = ld; a.rows = rows; a.vocab = vocab; a.inv_temp = 1.0f / temperature;
  a.pad_idx = pad_idx; a.eos_idx = eos_idx; a.block_eos = block_eos; a.forced_tok = -1;
  a.unk_idx = unk_idx; a.unk_penalty = unk_penalty;
  a.mode = sampler; a.top_k = top_k; a.top_p = top_p; a.z = (const unsigned long long*)z;
  a.out_tok = out_token; a.out_logp = out_logprob; a.out_kept_mass = (unsigned long long*)out_kept_mass;
  a.out_kept_count = out_kept_count;
  HIP_TRY(launch_sample_rows(a, (hipStream_t)stream));
  return SMI_OK;
}
