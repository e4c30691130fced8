// Device side of the EmbeddingToText hot path: one incremental decoder step and the
// beam-search bookkeeping, all on the GPU (no host round trip per step, no KV-cache copy).
//
// Reference behaviour restated here:
//  * decoder frontend / layer / final LN / tied projection:
//      sonar/models/sonar_text/factory.py:229-315, sonar/nn/conditional_decoder_model.py:66-94
//  * the sentence embedding is a length-1 encoder output
//      (sonar/models/sonar_translation/model.py:48-53): the cross-attention softmax is 1,
//      so cross_attn(x) = W_o (W_v e + b_v) + b_o, a per-sentence per-layer constant that is
//      precomputed once per generate() call (the reference recomputes q/k/softmax every step);
//  * beam search: fairseq2 ~=0.4 BeamSearchSeq2SeqGenerator + StandardBeamSearchAlgorithm
//      (SURVEY a24): fp32 log-softmax, PAD never, EOS blocked below min length and forced at
//      max length, top-(2*beam) over beam x vocab, EOS candidates among the first `beam`
//      finish (score / (len-1)^len_penalty), the rest continue.
//
// Beam re-indexing: every step's fused q|k|v GEMM output IS the cache slab of that position
// (kv[layer][pos][row][3d]); a hypothesis is an ancestry table anc[row][pos] -> row slot that
// produced that position.  Reordering beams copies 4 B per (row, pos) instead of 4 KB per
// (row, pos, layer) -- fairseq2's index_select on the whole cache is ~7 GB/step at t = 64.
#include <cstdlib>

#include "common.hpp"
#include <type_traits>

#include "kernels.hpp"

namespace smi {

// ------------------------------------------------------------- embed one position
// x[r,:] = E[tok[r]] * scale + PE[pos + off]; one wave per row.
__global__ __launch_bounds__(256) void dec_embed_kernel(const int32_t* __restrict__ tok,
                                                        const f16* __restrict__ table,
                                                        const float* __restrict__ pe_row, float scale,
                                                        float* __restrict__ x, int rows, int d,
                                                        int64_t vocab) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  int64_t t = tok[r];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  const f16* e = table + (size_t)t * d;
  float* o = x + (size_t)r * d;
  for (int c = lane * 8; c < d; c += 512) {
    const half8 ev = *(const half8*)(e + c);
    const f32x4 p0 = *(const f32x4*)(pe_row + c);
    const f32x4 p1 = *(const f32x4*)(pe_row + c + 4);
    f32x4 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o0[i] = (float)(f16)((float)ev[i] * scale) + p0[i];
      o1[i] = (float)(f16)((float)ev[i + 4] * scale) + p1[i];
    }
    *(f32x4*)(o + c) = o0;
    *(f32x4*)(o + c + 4) = o1;
  }
}

hipError_t launch_dec_embed(const int32_t* tok, const f16* table, const float* pe_row, float scale,
                            float* x, int rows, int d, int64_t vocab, hipStream_t stream) {
  hipLaunchKernelGGL(dec_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, tok, table, pe_row,
                     scale, x, rows, d, vocab);
  return hipGetLastError();
}

// ---------------------- x += sum_z parts[z] (+ c[row / group]); h = LN(x)  (fused)
// parts: split-K slabs of the preceding projection GEMM (fp32 [nparts][rows_pad][d]); c: the
// per-sentence cross-attention constant.  Either may be null.
__device__ unsigned g_prefetch_sink;  // never written in practice (common.hpp: prefetch_range)

// NP: the number of slabs when it is one of 0 / 1 / 2 / 4 / 8 (straight-line loads), -1: any number (rounds of ZB).
// PT: slab element type (float, or f16: an fp16 model's split-K projections store fp16 partials; the sum is formed in fp32).
template <int NV, typename XT, int NP, typename PT>
__global__ __launch_bounds__(256) void sum_ln_kernel(XT* __restrict__ x, const PT* __restrict__ parts,
                                                     int nparts, size_t part_stride,
                                                     const float* __restrict__ c, int group,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ b, float eps,
                                                     f16* __restrict__ h, int rows, int h_tm,
                                                     const void* __restrict__ pf, size_t pf_bytes, int main_blocks) {
  constexpr int D = NV * 256;
  if ((int)blockIdx.x >= main_blocks) {  // surplus workgroups: weight prefetch for a later GEMM (common.hpp)
    prefetch_range(pf, pf_bytes, blockIdx.x - main_blocks, gridDim.x - main_blocks, &g_prefetch_sink);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  XT* xr = x + (size_t)r * D;
  // The kernel is one dependent chain per wave (loads -> two wave reductions -> store) at ~1 wave per SIMD, so EVERY
  // load -- the row, the constant, the slabs, the LayerNorm weights -- is issued before the first use of any of them
  // and without a branch in between: one memory round trip instead of three (round 4: the fp16 -> fp32 conversion
  // of the row used to sit in front of the slab loads, the slab loads were separated by wave-uniform branches and the
  // weights were fetched after the reductions; r04 experiment 12).  The constant is read through a pointer that
  // falls back to a valid address and zeroed by a select.
  // The summation order (x, slabs ascending, constant) is fixed.  XT = f16: the residual stream is fp16 (the text
  // encoder's small-batch path): fp32 adds, ONE rounding when the row is written back; LayerNorm sees the rounded row.
  typedef typename std::conditional<sizeof(XT) == 2, half4, f32x4>::type XV;
  XV xv[NV];
  f32x4 v[NV], cv[NV], wv[NV], bv[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) xv[k] = *(const XV*)(xr + k * 256 + lane * 4);
  const float* cp = c ? c + (size_t)(r / group) * D + lane * 4 : w + lane * 4;
#pragma unroll
  for (int k = 0; k < NV; ++k) cv[k] = *(const f32x4*)(cp + k * 256);
  const PT* pr = parts + (size_t)r * D + lane * 4;
  typedef typename std::conditional<sizeof(PT) == 2, half4, f32x4>::type PV;
  auto widen = [](const PV& v) {
    if constexpr (sizeof(PT) == 2)
      return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    else
      return v;
  };
  constexpr int NPC = NP > 0 ? NP : 1;
  PV p[NPC][NV];
  if constexpr (NP > 0) {
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int k = 0; k < NV; ++k) p[j][k] = *(const PV*)(pr + (size_t)j * part_stride + k * 256);
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    wv[k] = *(const f32x4*)(w + k * 256 + lane * 4);
    bv[k] = *(const f32x4*)(b + k * 256 + lane * 4);
  }
  // the machine scheduler would otherwise interleave loads and adds in a rolling window of two or three loads in flight
  // (it minimises registers: 74 instead of ~200) -- the opposite of what a latency chain wants
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if constexpr (sizeof(XT) == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[k][i] = (float)xv[k][i];
    } else {
      v[k] = xv[k];
    }
    if (!c) cv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (NP > 0) {
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] += widen(p[j][k]);
  } else if constexpr (NP < 0) {
    constexpr int ZB = NV <= 4 ? 8 : 4;  // slabs per round trip (register budget: ZB * NV f32x4)
    for (int z0 = 0; z0 < nparts; z0 += ZB) {
      f32x4 q[ZB][NV];
#pragma unroll
      for (int j = 0; j < ZB; ++j)
#pragma unroll
        for (int k = 0; k < NV; ++k)
          q[j][k] = z0 + j < nparts ? widen(*(const PV*)(pr + (size_t)(z0 + j) * part_stride + k * 256))
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < ZB; ++j)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] += q[j][k];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    v[k] += cv[k];
    if constexpr (sizeof(XT) == 2) {
      half4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = (f16)v[k][i];
        v[k][i] = (float)o[i];
      }
      *(half4*)(xr + k * 256 + lane * 4) = o;
    } else {
      *(f32x4*)(xr + k * 256 + lane * 4) = v[k];
    }
    s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  constexpr float inv_d = 1.0f / D;
  const float mean = wave_sum(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = v[k][i] - mean;
      v[k][i] = t;
      q += t * t;
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    half4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (f16)(v[k][i] * rstd * wv[k][i] + bv[k][i]);
    if (h_tm)  // tile-major GEMM operand (common.hpp): the FFN inner projection's X
      *(half4*)(h + tm_offset(r, k * 256 + lane * 4, D)) = o;
    else
      *(half4*)(h + (size_t)r * D + k * 256 + lane * 4) = o;
  }
}

template <int NV, typename XT, typename PT>
static void launch_sum_ln_np(int blocks, hipStream_t stream, XT* x, const PT* parts, int nparts, size_t part_stride,
                             const float* c, int group, const float* w, const float* b, float eps, f16* h, int rows,
                             int h_tm, const void* pf, size_t pf_bytes) {
  // surplus workgroups for the weight prefetch: one per CU the row work leaves idle
  const int cus = num_cus();  // gemm.hip: per-device, atomically cached (decode chains call this from several host threads)
  // ... when the row work takes at most a quarter of the chip (<= 256 rows: the surplus then streams a 16.8 MB matrix in
  // one round trip; at 512 rows the prefetch cost more than it saved)
  const bool pf_env = tune(TUNE_PREFETCH, 1) != 0;
  const int extra = pf_env && pf && pf_bytes && blocks * 4 <= cus ? cus - blocks : 0;
  const int main_blocks = blocks;
#define SMI_SL(NP)                                                                                                     \
  hipLaunchKernelGGL((sum_ln_kernel<NV, XT, NP, PT>), dim3(blocks + extra), dim3(256), 0, stream, x, parts, nparts,    \
                     part_stride, c, group, w, b, eps, h, rows, h_tm, pf, pf_bytes, main_blocks)
  switch (nparts) {
    case 0: SMI_SL(0); break;
    case 1: SMI_SL(1); break;
    case 2: SMI_SL(2); break;
    case 4: SMI_SL(4); break;
    case 8:
      if constexpr (NV <= 4) {
        SMI_SL(8);
        break;
      }
      [[fallthrough]];
    default: SMI_SL(-1); break;
  }
#undef SMI_SL
}

hipError_t launch_sum_layernorm(void* x, const void* parts, int nparts, size_t part_stride,
                                const float* c, int group, const float* w, const float* b, float eps,
                                f16* h, int rows, int d, hipStream_t stream, int h_tm, int x_f16, const void* pf,
                                size_t pf_bytes, int parts_f16) {
  const int blocks = (rows + 3) / 4;
  if (!parts) nparts = 0;
#define SMI_AL_GO(NV, XT, PT)                                                                                           \
  launch_sum_ln_np<NV, XT, PT>(blocks, stream, (XT*)x, (const PT*)parts, nparts, part_stride, c, group, w, b, eps, h, rows, \
                               h_tm, pf, pf_bytes)
#define SMI_AL_CASE(NV)                       \
  case NV * 256:                              \
    if (x_f16) {                              \
      if (parts_f16) SMI_AL_GO(NV, f16, f16); \
      else SMI_AL_GO(NV, f16, float);         \
    } else {                                  \
      if (parts_f16) SMI_AL_GO(NV, float, f16); \
      else SMI_AL_GO(NV, float, float);       \
    }                                         \
    break;
  switch (d) {
    SMI_AL_CASE(1)
    SMI_AL_CASE(2)
    SMI_AL_CASE(3)
    SMI_AL_CASE(4)
    SMI_AL_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef SMI_AL_CASE
#undef SMI_AL_GO
  return hipGetLastError();
}

// -------------------------------------------------- single-query (decode) attention
// One wave per (row, head).  kv: [pos][rows_pad][3d] (q|k|v), anc: [rows][anc_stride].
// Lane (pg = lane>>3, c = lane&7) owns the 8 head dims c*8..c*8+7 of the positions j0 + 8*it + pg:
// every load instruction fetches 8 whole 128-B K (or V) rows as 16 B per lane, a score is joined
// across the 8 lanes of its position by 3 xor-shuffles, the softmax statistics across the 8 position
// groups by 3 more, and the probabilities never leave the lanes that multiply them into V.
// NG = 8-position groups per chunk (chunk = 8 NG positions), chosen by the launcher: the first chunk is rarely full at
// decode time, and a compile-time chunk size keeps the loop body straight-line (run-time skipping of masked groups
// was slower than loading them, r02 experiment 20).
template <int NG>
__global__ __launch_bounds__(256) void dec_attention_kernel(const f16* __restrict__ kv,
                                                            const int32_t* __restrict__ anc,
                                                            int anc_stride, f16* __restrict__ ctx,
                                                            int rows, int rows_pad, int d, int heads,
                                                            int pos, float sl2e) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= rows * heads) return;
  const int r = wid / heads, h = wid % heads;
  const int lane = threadIdx.x & 63, pg = lane >> 3, c = lane & 7;
  const size_t ld = (size_t)3 * d;
  const size_t slab = (size_t)rows_pad * ld;
  const half8 qh = *(const half8*)(kv + (size_t)pos * slab + (size_t)r * ld + h * 64 + c * 8);
  float q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] = (float)qh[e] * sl2e;

  const int32_t* ar = anc + (size_t)r * anc_stride;
  const f16* kbase = kv + d + h * 64 + c * 8;
  float m = -1e30f, l = 0.f;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int j0 = 0; j0 <= pos; j0 += 8 * NG) {
    float s[NG];
    // masked positions (j > pos) load ONE shared cached line (position 0, slot 0) in straight-line code: skipping
    // them per 8-position group (uniform branches) measured 4.38 against 4.26 ms per step (r02 experiment 20)
    const f16* row[NG];
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const int j = j0 + it * 8 + pg;
      const bool valid = j <= pos;
      const int src = valid ? (j == pos ? r : ar[j]) : 0;
      row[it] = kbase + (size_t)(valid ? j : 0) * slab + (size_t)src * ld;
    }
    // all 8 K rows and all 8 V rows (the V row follows its K row at +d) are requested before the
    // first score is formed: one memory round trip per 64 positions instead of two
    half8 kr[NG], vr[NG];
#pragma unroll
    for (int it = 0; it < NG; ++it) kr[it] = *(const half8*)row[it];
#pragma unroll
    for (int it = 0; it < NG; ++it) vr[it] = *(const half8*)(row[it] + d);
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const bool valid = j0 + it * 8 + pg <= pos;
      const half8 kk = kr[it];
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += q[e] * (float)kk[e];
      acc = sum_over_8(acc);
      s[it] = valid ? acc : -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int it = 1; it < NG; ++it) mx = fmaxf(mx, s[it]);
    mx = max_over_groups_of_8(mx);
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float ps = 0.f;
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      s[it] = __builtin_amdgcn_exp2f(s[it] - m_new);  // 0 for masked positions
      ps += s[it];
    }
    ps = sum_over_groups_of_8(ps);
    l = l * alpha + ps;
    m = m_new;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const half8 vv = vr[it];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += s[it] * (float)vv[e];
    }
  }
  half8 out;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = sum_over_groups_of_8(o[e]);
    out[e] = (f16)(v / l);
  }
  if (pg == 0) *(half8*)(ctx + (size_t)r * d + h * 64 + c * 8) = out;
}

hipError_t launch_dec_attention(const f16* kv, const int32_t* anc, int anc_stride, f16* ctx, int rows,
                                int rows_pad, int d, int heads, int pos, hipStream_t stream) {
  const float sl2e = 0.125f * 1.4426950408889634f;
  const int waves = rows * heads;
  const dim3 grid((waves + 3) / 4);
#define SMI_DA_LAUNCH(NG)                                                                                          \
  hipLaunchKernelGGL(dec_attention_kernel<NG>, grid, dim3(256), 0, stream, kv, anc, anc_stride, ctx, rows, rows_pad, \
                     d, heads, pos, sl2e)
  // as few passes as 64-position chunks would need (each pass is a dependent memory round trip), and the smallest
  // chunk that covers positions 0..pos in that many passes
  const int passes = pos / 64 + 1;
  const int ng = (pos / 8 + 1 + passes - 1) / passes;
  switch (ng) {
    case 1: SMI_DA_LAUNCH(1); break;
    case 2: SMI_DA_LAUNCH(2); break;
    case 3: SMI_DA_LAUNCH(3); break;
    case 4: SMI_DA_LAUNCH(4); break;
    case 5: SMI_DA_LAUNCH(5); break;
    case 6: SMI_DA_LAUNCH(6); break;
    case 7: SMI_DA_LAUNCH(7); break;
    default: SMI_DA_LAUNCH(8); break;
  }
#undef SMI_DA_LAUNCH
  return hipGetLastError();
}

// candidate ordering shared by the selection and the beam step: value desc, token asc
constexpr int VS_K2MAX = 16;

__device__ __forceinline__ bool cand_better(float a, int ia, float b, int ib) {
  return a > b || (a == b && ia < ib);
}

// ------------------------------------------------- vocabulary select (per row, tile statistics)
// The logits GEMM leaves, per row and 256-column tile, the tile maximum and sum exp(v - max)
// (GemmTileStats).  One workgroup per row then
//  1. folds the tile statistics into the row's softmax normaliser (pmax, psum), and
//  2. finds the top-k2 candidates WITHOUT reading the whole logits row: the k2 best tiles by maximum
//     (ties: lower tile first) among tiles >= 1 contain every candidate that can reach the top-k2 --
//     each of them holds an element >= the k2-th best tile maximum, and an element of a later tile
//     loses every value tie against them (lower token wins).  Tile 0 is always read too: it holds the
//     tokens the generation masks touch (PAD, UNK penalty, blocked EOS), so its raw maximum says
//     nothing.  Thread t owns column t of every selected tile; k2 rounds of a workgroup-wide arg-max
//     over sortable (value desc, token asc) keys produce the exact ordered list.
constexpr int VSEL_SLOTS = VS_K2MAX + 1;

__device__ __forceinline__ unsigned long long cand_key(float v, int idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned)(0xffffffffu - (unsigned)idx);
}
// Wave-wide arg-best reductions of the selection kernels on DPP row operations + v_permlane16/32_swap (VALU rate) instead of
// __shfl_xor (= ds_bpermute, an LDS round trip per step and word: vocab_select ran 20 rounds x 12 of them back to back,
// the beam step ~420 per sentence).  Step order as wave_max (common.hpp): xor 1, xor 2, half-row mirror, row mirror, then the
// two lane swaps, whose both results are folded (one is the lane's own value, the other its partner's).  Every lane ends
// with the same winner: the orders are total.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
  auto fold = [&](unsigned lo, unsigned hi) {
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    k = o > k ? o : k;
  };
  fold(dpp_u<0xB1>((unsigned)k), dpp_u<0xB1>((unsigned)(k >> 32)));
  fold(dpp_u<0x4E>((unsigned)k), dpp_u<0x4E>((unsigned)(k >> 32)));
  fold(dpp_u<0x141>((unsigned)k), dpp_u<0x141>((unsigned)(k >> 32)));
  fold(dpp_u<0x140>((unsigned)k), dpp_u<0x140>((unsigned)(k >> 32)));
  {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)k, (unsigned)k, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(k >> 32), (unsigned)(k >> 32), false, false);
    const unsigned long long a = ((unsigned long long)hi[0] << 32) | lo[0], b = ((unsigned long long)hi[1] << 32) | lo[1];
    k = a > b ? a : b;
  }
  {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)k, (unsigned)k, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(k >> 32), (unsigned)(k >> 32), false, false);
    const unsigned long long a = ((unsigned long long)hi[0] << 32) | lo[0], b = ((unsigned long long)hi[1] << 32) | lo[1];
    k = a > b ? a : b;
  }
  return k;
}
// best (value desc, token asc) candidate of the wave, in every lane
__device__ __forceinline__ void wave_best2(float& bv, int& bi) {
  auto fold = [&](unsigned ov_, unsigned oi_) {
    const float ov = __uint_as_float(ov_);
    const int oi = (int)oi_;
    if (cand_better(ov, oi, bv, bi)) {
      bv = ov;
      bi = oi;
    }
  };
  fold(dpp_u<0xB1>(__float_as_uint(bv)), dpp_u<0xB1>((unsigned)bi));
  fold(dpp_u<0x4E>(__float_as_uint(bv)), dpp_u<0x4E>((unsigned)bi));
  fold(dpp_u<0x141>(__float_as_uint(bv)), dpp_u<0x141>((unsigned)bi));
  fold(dpp_u<0x140>(__float_as_uint(bv)), dpp_u<0x140>((unsigned)bi));
  {
    const auto v = __builtin_amdgcn_permlane16_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
    const auto i = __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false);
    bv = __uint_as_float(v[0]);
    bi = (int)i[0];
    fold(v[1], i[1]);
  }
  {
    const auto v = __builtin_amdgcn_permlane32_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
    const auto i = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
    bv = __uint_as_float(v[0]);
    bi = (int)i[0];
    fold(v[1], i[1]);
  }
}
// best (score desc, row asc, token asc) continuation of the wave, in every lane
__device__ __forceinline__ void wave_best3(float& bv, int& br, int& bt) {
  auto fold = [&](unsigned ov_, unsigned or_, unsigned ot_) {
    const float ov = __uint_as_float(ov_);
    const int orow = (int)or_, otok = (int)ot_;
    if (ov > bv || (ov == bv && (orow < br || (orow == br && otok < bt)))) {
      bv = ov;
      br = orow;
      bt = otok;
    }
  };
  fold(dpp_u<0xB1>(__float_as_uint(bv)), dpp_u<0xB1>((unsigned)br), dpp_u<0xB1>((unsigned)bt));
  fold(dpp_u<0x4E>(__float_as_uint(bv)), dpp_u<0x4E>((unsigned)br), dpp_u<0x4E>((unsigned)bt));
  fold(dpp_u<0x141>(__float_as_uint(bv)), dpp_u<0x141>((unsigned)br), dpp_u<0x141>((unsigned)bt));
  fold(dpp_u<0x140>(__float_as_uint(bv)), dpp_u<0x140>((unsigned)br), dpp_u<0x140>((unsigned)bt));
  {
    const auto v = __builtin_amdgcn_permlane16_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)br, (unsigned)br, false, false);
    const auto t = __builtin_amdgcn_permlane16_swap((unsigned)bt, (unsigned)bt, false, false);
    bv = __uint_as_float(v[0]);
    br = (int)r[0];
    bt = (int)t[0];
    fold(v[1], r[1], t[1]);
  }
  {
    const auto v = __builtin_amdgcn_permlane32_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)br, (unsigned)br, false, false);
    const auto t = __builtin_amdgcn_permlane32_swap((unsigned)bt, (unsigned)bt, false, false);
    bv = __uint_as_float(v[0]);
    br = (int)r[0];
    bt = (int)t[0];
    fold(v[1], r[1], t[1]);
  }
}

// logits of (row, token): fp32 row-major [rows][ldl], or -- f16_tm -- fp16 in the tile-major layout of common.hpp with K = ldl
// (what the logits GEMM of an fp16 model stores, round 4)
__device__ __forceinline__ float logit_at(const float* __restrict__ logits, int ldl, int f16_tm, int row, int tok) {
  return f16_tm ? (float)((const f16*)logits)[tm_offset(row, tok, ldl)] : logits[(size_t)row * ldl + tok];
}

__global__ __launch_bounds__(256) void vocab_select_kernel(const float* __restrict__ logits, int ldl, int f16_tm, int vocab,
                                                           const float* __restrict__ tile_max,
                                                           const float* __restrict__ tile_sum, int ntiles,
                                                           int stat_rows, int k2, float inv_temp, int pad_idx, int eos_idx, int unk_idx,
                                                           float unk_penalty, int block_eos,
                                                           float* __restrict__ pmax, float* __restrict__ psum,
                                                           float* __restrict__ pval, int* __restrict__ pidx) {
  __shared__ float s_f[4];
  __shared__ int s_sel[VSEL_SLOTS];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* tm = tile_max + row;  // [tile][stat_rows]
  const float* ts = tile_sum + row;
  constexpr int TPT = 8;  // tiles per thread: up to 2048 tiles = 524288 tokens
  float m[TPT], sm[TPT];
  float lm = -INFINITY;
#pragma unroll
  for (int j = 0; j < TPT; ++j) {
    const int t = tid + 256 * j;
    m[j] = t < ntiles ? tm[(size_t)t * stat_rows] : -INFINITY;
    sm[j] = t < ntiles ? ts[(size_t)t * stat_rows] : 0.f;
    lm = fmaxf(lm, m[j]);
  }
  // ---- 1. softmax normaliser of the row
  lm = wave_max(lm);
  if (lane == 0) s_f[wv] = lm;
  __syncthreads();
  const float M = fmaxf(fmaxf(s_f[0], s_f[1]), fmaxf(s_f[2], s_f[3]));
  float se = 0.f;
#pragma unroll
  for (int j = 0; j < TPT; ++j)
    if (m[j] != -INFINITY) se += sm[j] * __expf(m[j] - M);
  se = wave_sum(se);
  __syncthreads();
  if (lane == 0) s_f[wv] = se;
  __syncthreads();
  if (tid == 0) {
    pmax[row] = M;
    psum[row] = (s_f[0] + s_f[1]) + (s_f[2] + s_f[3]);
  }
  if (k2 == 0) return;
  // The two selections below are k2 rounds of "workgroup-wide maximum of a sortable key, then retire the winner".  The kernel is
  // VALU-bound (1280 rows x 4 waves on 1024 SIMDs: a round used to rebuild and compare the keys of all of a thread's slots,
  // ~230 VALU instructions, 20 rounds), so every thread now keeps its keys and its current best: a round is one wave reduction
  // of the per-thread bests, one LDS exchange behind ONE barrier (the slots alternate between two buffers), and only the
  // winner's thread rescans its slots.
  __shared__ unsigned long long s_k2[2][4];
  auto wg_max = [&](unsigned long long mine, int round) {
    const unsigned long long w = wave_max_u64(mine);
    if (lane == 0) s_k2[round & 1][wv] = w;
    __syncthreads();
    unsigned long long b = s_k2[round & 1][0];
#pragma unroll
    for (int q = 1; q < 4; ++q) b = s_k2[round & 1][q] > b ? s_k2[round & 1][q] : b;
    return b;
  };
  // ---- 2a. the k2 best tiles among tiles >= 1 (value desc, tile asc), plus tile 0
  unsigned long long tk[TPT];
  unsigned long long tbest = 0ull;
#pragma unroll
  for (int j = 0; j < TPT; ++j) {
    const int t = tid + 256 * j;
    tk[j] = (m[j] != -INFINITY && t != 0) ? cand_key(m[j], t) : 0ull;  // tile 0 is taken unconditionally
    tbest = tk[j] > tbest ? tk[j] : tbest;
  }
  if (tid == 0) s_sel[0] = 0;
  int nsel = 1;
  for (int round = 0; round < k2; ++round) {
    const unsigned long long b = wg_max(tbest, round);
    if (b == 0ull) break;  // fewer than k2 tiles
    if (tid == 0) s_sel[nsel] = (int)(0xffffffffu - (unsigned)(b & 0xffffffffu));
    ++nsel;
    if (b == tbest) {  // keys are unique (the tile index is part of them): exactly one thread
      tbest = 0ull;
#pragma unroll
      for (int j = 0; j < TPT; ++j) {
        if (tk[j] == b) tk[j] = 0ull;
        tbest = tk[j] > tbest ? tk[j] : tbest;
      }
    }
  }
  __syncthreads();
  // ---- 2b. thread t owns column t of every selected tile
  unsigned long long ck[VSEL_SLOTS];
  unsigned long long cbest = 0ull;
#pragma unroll
  for (int j = 0; j < VSEL_SLOTS; ++j) {
    ck[j] = 0ull;
    if (j < nsel) {
      const int tok = s_sel[j] * 256 + tid;
      if (tok < vocab && tok != pad_idx && !(block_eos && tok == eos_idx)) {
        float v = logit_at(logits, ldl, f16_tm, row, tok) * inv_temp;
        if (tok == unk_idx) v -= unk_penalty;
        if (v != -INFINITY) ck[j] = cand_key(v, tok);
      }
    }
    cbest = ck[j] > cbest ? ck[j] : cbest;
  }
  // ---- 2c. ordered top-k2 by k2 workgroup-wide arg-max rounds
  for (int round = 0; round < k2; ++round) {
    const unsigned long long b = wg_max(cbest, round + k2);
    float val = -INFINITY;
    int idx = 0x7fffffff;
    if (b != 0ull) {
      unsigned u = (unsigned)(b >> 32);
      u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
      val = __uint_as_float(u);
      idx = (int)(0xffffffffu - (unsigned)(b & 0xffffffffu));
    }
    if (tid == 0) {
      pval[(size_t)row * VS_K2MAX + round] = val;
      pidx[(size_t)row * VS_K2MAX + round] = idx;
    }
    if (b != 0ull && b == cbest) {
      cbest = 0ull;
#pragma unroll
      for (int j = 0; j < VSEL_SLOTS; ++j) {
        if (ck[j] == b) ck[j] = 0ull;
        cbest = ck[j] > cbest ? ck[j] : cbest;
      }
    }
  }
}

hipError_t launch_vocab_select(const float* logits, int ldl, int f16_tm, int rows, int vocab, const float* tile_max,
                               const float* tile_sum, int ntiles, int stat_rows, int k2, float inv_temp, int pad_idx, int eos_idx,
                               int unk_idx, float unk_penalty, int block_eos, float* pmax, float* psum, float* pval,
                               int* pidx, hipStream_t stream) {
  if (rows <= 0 || stat_rows < rows || ntiles <= 0 || ntiles > 2048 || k2 < 0 || k2 > VS_K2MAX || (int64_t)ntiles * 256 < vocab)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(vocab_select_kernel, dim3(rows), dim3(256), 0, stream, logits, ldl, f16_tm, vocab, tile_max, tile_sum,
                     ntiles, stat_rows, k2, inv_temp, pad_idx, eos_idx, unk_idx, unk_penalty, block_eos, pmax, psum, pval, pidx);
  return hipGetLastError();
}

// ----------------------------------------------------------------- beam step (per sentence)
// One workgroup (256 threads) per sentence.  Merges the chunk partials of each live row,
// forms the 2*beam best (row, token) continuations and applies the fairseq2 EOS rules.
struct BeamState {
  int32_t* tok;        // [R] token fed at the current position
  float* cum;          // [R] cumulative log-prob
  int32_t* nactive;    // [n] live rows of the sentence (1 until the first expansion)
  int32_t* done;       // [n]
  int32_t* ndone;      // [1]
  int32_t* parent;     // [R] out: parent row (global slot) of the row's next state
  int32_t* new_tok;    // [R] out
  float* new_cum;      // [R] out
  const int32_t* hist; // [R][hist_stride] tokens of every position <= pos
  int32_t* fin_tok;    // [n][beam][hist_stride]
  int32_t* fin_len;    // [n][beam]
  float* fin_score;    // [n][beam]
  int32_t* fin_count;  // [n]
  float* margins;      // [n][2] smallest decision gap so far: {step candidates (log-prob), final ranking}
};

__global__ __launch_bounds__(256) void beam_step_kernel(BeamState st, const float* __restrict__ logits,
                                                        int ldl, int f16_tm, const float* __restrict__ pmax,
                                                        const float* __restrict__ psum,
                                                        const float* __restrict__ pval,
                                                        const int* __restrict__ pidx, int nchunks,
                                                        int beam, int k2, int pos, int prompt_len,
                                                        int forced_tok, int max_len, float inv_temp,
                                                        float len_penalty, int normalize, int eos_idx,
                                                        int hist_stride) {
  __shared__ float s_lse[8];
  __shared__ float c_val[8 * VS_K2MAX];
  __shared__ int c_tok[8 * VS_K2MAX];
  __shared__ int c_row[8 * VS_K2MAX];
  __shared__ int f_row[8];
  __shared__ float f_score[8];
  __shared__ int f_n, f_base;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int base = s * beam;
  const int step_nr = pos + 1;  // index of the token chosen now
  if (st.done[s]) {  // keep the rows inert
    if (tid < beam) {
      st.parent[base + tid] = base + tid;
      st.new_tok[base + tid] = st.tok[base + tid];
      st.new_cum[base + tid] = st.cum[base + tid];
    }
    return;
  }
  const int na = st.nactive[s];
  // (1) log-sum-exp of every live row
  for (int r = wv; r < na; r += 4) {
    const size_t o = (size_t)(base + r) * nchunks;
    float m = -INFINITY;
    for (int c = lane; c < nchunks; c += 64) m = fmaxf(m, pmax[o + c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < nchunks; c += 64) sum += psum[o + c] * __expf(pmax[o + c] - m);
    sum = wave_sum(sum);
    if (lane == 0) s_lse[r] = m + __logf(sum);
  }
  __syncthreads();

  const bool forced_prompt = step_nr < prompt_len;
  const bool force_eos = !forced_prompt && step_nr == max_len - 1;
  if (forced_prompt || force_eos) {
    // candidate of every live row is one given token
    if (tid < na) {
      const int tk = forced_prompt ? forced_tok : eos_idx;
      const float lp = logit_at(logits, ldl, f16_tm, base + tid, tk) * inv_temp - s_lse[tid];
      c_val[tid] = st.cum[base + tid] + lp;
      c_tok[tid] = tk;
      c_row[tid] = base + tid;
    }
    __syncthreads();
    if (tid == 0) {
      if (forced_prompt) {
        for (int r = 0; r < beam; ++r) {
          const bool live = r < na;
          st.parent[base + r] = base + r;
          st.new_tok[base + r] = live ? c_tok[r] : st.tok[base + r];
          st.new_cum[base + r] = live ? c_val[r] : st.cum[base + r];
        }
        f_n = 0;
      } else {
        // sort the <= beam EOS candidates by score (desc, row asc) and finish them all
        int order[8];
        for (int r = 0; r < na; ++r) order[r] = r;
        for (int a = 1; a < na; ++a)
          for (int b = a; b > 0 && cand_better(c_val[order[b]], order[b], c_val[order[b - 1]], order[b - 1]); --b) {
            const int t = order[b];
            order[b] = order[b - 1];
            order[b - 1] = t;
          }
        int cnt = st.fin_count[s];
        f_base = cnt;
        int nf = 0;
        for (int a = 0; a < na && cnt < beam; ++a) {
          f_row[nf] = c_row[order[a]];
          f_score[nf] = c_val[order[a]];
          ++nf;
          ++cnt;
        }
        f_n = nf;
        st.fin_count[s] = cnt;
        st.done[s] = 1;  // nothing can continue past the maximum length
        atomicAdd(st.ndone, 1);
        for (int r = 0; r < beam; ++r) {
          st.parent[base + r] = base + r;
          st.new_tok[base + r] = st.tok[base + r];
          st.new_cum[base + r] = st.cum[base + r];
        }
      }
    }
  } else {
    // (2) per live row: best k2 tokens over its chunk partials (value desc, token asc).
    // One wave per row; its <= 1024 partial candidates sit in registers (16 per lane).
    for (int r = wv; r < na; r += 4) {
      const size_t o = (size_t)(base + r) * nchunks * VS_K2MAX;
      if (nchunks == 1) {  // the tile-statistics selection already leaves ONE ordered list per row: nothing to merge
        if (lane < k2) {
          const float v = pval[o + lane];
          c_val[r * VS_K2MAX + lane] = v == -INFINITY ? -INFINITY : st.cum[base + r] + v - s_lse[r];
          c_tok[r * VS_K2MAX + lane] = pidx[o + lane];
          c_row[r * VS_K2MAX + lane] = base + r;
        }
        continue;
      }
      const int total = nchunks * k2;
      float cv[16];
      int ci[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int e = q * 64 + lane;
        cv[q] = -INFINITY;
        ci[q] = 0x7fffffff;
        if (e < total) {
          const int off = (e / k2) * VS_K2MAX + (e % k2);
          cv[q] = pval[o + off];
          ci[q] = pidx[o + off];
        }
      }
      for (int round = 0; round < k2; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (cand_better(cv[q], ci[q], bv, bi)) {
            bv = cv[q];
            bi = ci[q];
          }
        wave_best2(bv, bi);
        if (lane == 0) {
          c_val[r * VS_K2MAX + round] = bv == -INFINITY ? -INFINITY : st.cum[base + r] + bv - s_lse[r];
          c_tok[r * VS_K2MAX + round] = bi;
          c_row[r * VS_K2MAX + round] = base + r;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (ci[q] == bi) cv[q] = -INFINITY;
      }
    }
    __syncthreads();
    // (3) wave 0: overall top-k2 over the na*k2 row candidates (score desc, row asc, token asc),
    //     two candidates per lane; then thread 0 applies the EOS rules on the sorted list.
    __shared__ float t_val[VS_K2MAX];
    __shared__ int t_tok[VS_K2MAX], t_row[VS_K2MAX];
    __shared__ int t_n;
    if (wv == 0) {
      float v2[2];
      int r2[2], k2t[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = u * 64 + lane;  // candidate id = row * VS_K2MAX + rank
        const bool ok = e < na * VS_K2MAX && (e % VS_K2MAX) < k2;
        v2[u] = ok ? c_val[e] : -INFINITY;
        r2[u] = ok ? c_row[e] : 0x7fffffff;
        k2t[u] = ok ? c_tok[e] : 0x7fffffff;
      }
      int nsel = 0;
      for (int sel = 0; sel < k2; ++sel) {
        // lane-local best of its two, then wave arg-best
        int u = (v2[1] > v2[0] || (v2[1] == v2[0] && (r2[1] < r2[0] || (r2[1] == r2[0] && k2t[1] < k2t[0])))) ? 1 : 0;
        float bv = v2[u];
        int br = r2[u], bt = k2t[u];
        wave_best3(bv, br, bt);
        if (bv == -INFINITY) break;
        if (lane == 0) {
          t_val[sel] = bv;
          t_tok[sel] = bt;
          t_row[sel] = br;
        }
        ++nsel;
#pragma unroll
        for (int w = 0; w < 2; ++w)
          if (r2[w] == br && k2t[w] == bt) v2[w] = -INFINITY;
      }
      if (lane == 0) t_n = nsel;
    }
    __syncthreads();
    if (tid == 0) {
      const int nsel = t_n;
      const float* tv = t_val;
      const int* tt = t_tok;
      const int* tr = t_row;
      int cnt = st.fin_count[s];
      f_base = cnt;
      int nf = 0;
      bool finished_all = false;
      for (int i = 0; i < nsel && i < beam; ++i) {
        if (tt[i] == eos_idx) {
          f_row[nf] = tr[i];
          f_score[nf] = tv[i];
          ++nf;
          if (++cnt == beam) {
            finished_all = true;
            break;
          }
        }
      }
      f_n = nf;
      st.fin_count[s] = cnt;
      if (st.margins) {
        // decision margin of this step: the smallest gap between neighbours of the sorted candidate
        // list up to (and including) the first candidate that is NOT consumed.  For beam 1 this is
        // the greedy top-1 / top-2 log-prob margin.  A parity test may excuse a token mismatch
        // only where this measured margin is below its stated epsilon.
        int last = nsel - 1;
        if (finished_all) {  // nothing after the hypothesis that completed the beam matters
          int c2 = f_base;
          for (int i = 0; i < nsel && i < beam; ++i)
            if (tt[i] == eos_idx && ++c2 == beam) {
              last = i;
              break;
            }
        } else {
          int taken = 0;
          for (int i = 0; i < nsel; ++i) {
            if (tt[i] == eos_idx) continue;
            if (++taken == beam) {
              last = i;
              break;
            }
          }
        }
        float g = st.margins[2 * s];
        for (int i = 0; i <= last && i + 1 < nsel; ++i) g = fminf(g, tv[i] - tv[i + 1]);
        st.margins[2 * s] = g;
      }
      if (finished_all) {
        st.done[s] = 1;
        atomicAdd(st.ndone, 1);
        for (int r = 0; r < beam; ++r) {
          st.parent[base + r] = base + r;
          st.new_tok[base + r] = st.tok[base + r];
          st.new_cum[base + r] = st.cum[base + r];
        }
      } else {
        int w = 0;
        for (int i = 0; i < nsel && w < beam; ++i) {
          if (tt[i] == eos_idx) continue;
          st.parent[base + w] = tr[i];
          st.new_tok[base + w] = tt[i];
          st.new_cum[base + w] = tv[i];
          ++w;
        }
        st.nactive[s] = w;
        for (; w < beam; ++w) {  // unreachable with vocab > 2*beam; keep the slot inert
          st.parent[base + w] = base + w;
          st.new_tok[base + w] = st.tok[base + w];
          st.new_cum[base + w] = -INFINITY;
        }
      }
    }
  }
  __syncthreads();
  // (4) copy out the hypotheses that finished at this step: generated tokens + EOS
  const int nf = f_n;
  for (int f = 0; f < nf; ++f) {
    const int slot = f_base + f;
    const int row = f_row[f];
    const int glen = step_nr - prompt_len;  // generated tokens before the EOS
    int32_t* dst = st.fin_tok + ((size_t)s * beam + slot) * hist_stride;
    const int32_t* src = st.hist + (size_t)row * hist_stride + prompt_len;
    for (int i = tid; i < glen; i += 256) dst[i] = src[i];
    if (tid == 0) {
      dst[glen] = eos_idx;
      st.fin_len[s * beam + slot] = glen + 1;
      const float seq_len = (float)(step_nr + 1);
      st.fin_score[s * beam + slot] = normalize ? f_score[f] / __powf(seq_len - 1.f, len_penalty) : f_score[f];
    }
  }
}

hipError_t launch_beam_step(const BeamStepArgs& a, hipStream_t stream) {
  BeamState st{a.tok, a.cum, a.nactive, a.done, a.ndone, a.parent, a.new_tok, a.new_cum,
               a.hist, a.fin_tok, a.fin_len, a.fin_score, a.fin_count, a.margins};
  hipLaunchKernelGGL(beam_step_kernel, dim3(a.n), dim3(256), 0, stream, st, a.logits, a.ldl, a.logits_f16_tm, a.pmax,
                     a.psum, a.pval, a.pidx, a.nchunks, a.beam, a.k2, a.pos, a.prompt_len, a.forced_tok,
                     a.max_len, a.inv_temp, a.len_penalty, a.normalize, a.eos_idx, a.hist_stride);
  return hipGetLastError();
}

// ------------------------------------------------------------------------ beam reorder
// new_anc[r][j] = anc[parent[r]][j] (j < pos), new_anc[r][pos] = parent[r];
// new_hist[r][j] = hist[parent[r]][j] (j <= pos), new_hist[r][pos+1] = new_tok[r];
// tok/cum take their new values.
__global__ __launch_bounds__(256) void beam_reorder_kernel(const int32_t* __restrict__ parent,
                                                           const int32_t* __restrict__ new_tok,
                                                           const float* __restrict__ new_cum,
                                                           const int32_t* __restrict__ anc,
                                                           int32_t* __restrict__ anc2,
                                                           const int32_t* __restrict__ hist,
                                                           int32_t* __restrict__ hist2,
                                                           int32_t* __restrict__ tok,
                                                           float* __restrict__ cum, int stride,
                                                           int pos) {
  const int r = blockIdx.x;
  const int p = parent[r];
  for (int j = threadIdx.x; j <= pos; j += 256) {
    anc2[(size_t)r * stride + j] = j == pos ? p : anc[(size_t)p * stride + j];
    hist2[(size_t)r * stride + j] = hist[(size_t)p * stride + j];
  }
  if (threadIdx.x == 0) {
    hist2[(size_t)r * stride + pos + 1] = new_tok[r];
    tok[r] = new_tok[r];
    cum[r] = new_cum[r];
  }
}

hipError_t launch_beam_reorder(const int32_t* parent, const int32_t* new_tok, const float* new_cum,
                               const int32_t* anc, int32_t* anc2, const int32_t* hist, int32_t* hist2,
                               int32_t* tok, float* cum, int rows, int stride, int pos,
                               hipStream_t stream) {
  hipLaunchKernelGGL(beam_reorder_kernel, dim3(rows), dim3(256), 0, stream, parent, new_tok, new_cum,
                     anc, anc2, hist, hist2, tok, cum, stride, pos);
  return hipGetLastError();
}

// ------------------------------------------------------------------------- small helpers
__global__ void beam_init_kernel(int32_t* tok, float* cum, int32_t* nactive, int32_t* done,
                                 int32_t* ndone, int32_t* fin_count, int32_t* hist, int32_t* anc,
                                 float* margins, int rows, int n, int stride, int first_tok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) {
    tok[i] = first_tok;
    cum[i] = 0.f;
    hist[(size_t)i * stride] = first_tok;
    anc[(size_t)i * stride] = i;
  }
  if (i < n) {
    nactive[i] = 1;
    done[i] = 0;
    fin_count[i] = 0;
    if (margins) margins[2 * i] = margins[2 * i + 1] = INFINITY;
  }
  if (i == 0) *ndone = 0;
}

hipError_t launch_beam_init(int32_t* tok, float* cum, int32_t* nactive, int32_t* done, int32_t* ndone,
                            int32_t* fin_count, int32_t* hist, int32_t* anc, float* margins, int rows, int n,
                            int stride, int first_tok, hipStream_t stream) {
  const int total = rows > n ? rows : n;
  hipLaunchKernelGGL(beam_init_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, tok, cum, nactive,
                     done, ndone, fin_count, hist, anc, margins, rows, n, stride, first_tok);
  return hipGetLastError();
}

// finished hypotheses of each sentence sorted best first -> caller buffers
__global__ void beam_output_kernel(const int32_t* __restrict__ fin_tok, const int32_t* __restrict__ fin_len,
                                   const float* __restrict__ fin_score, const int32_t* __restrict__ fin_count,
                                   int beam, int stride, int out_stride, int32_t* __restrict__ out_tok,
                                   int32_t* __restrict__ out_len, float* __restrict__ out_score,
                                   float* __restrict__ margins) {
  __shared__ int order[8];
  const int s = blockIdx.x;
  const int cnt = fin_count[s];
  if (threadIdx.x == 0) {
    for (int i = 0; i < cnt; ++i) order[i] = i;
    for (int a = 1; a < cnt; ++a)  // stable insertion sort, score descending
      for (int b = a; b > 0 && fin_score[s * beam + order[b]] > fin_score[s * beam + order[b - 1]]; --b) {
        const int t = order[b];
        order[b] = order[b - 1];
        order[b - 1] = t;
      }
    // gap between the returned hypothesis and the runner-up (normalised-score units)
    if (margins && cnt >= 2) margins[2 * s + 1] = fin_score[s * beam + order[0]] - fin_score[s * beam + order[1]];
  }
  __syncthreads();
  for (int h = 0; h < beam; ++h) {
    int32_t* dst = out_tok + ((size_t)s * beam + h) * out_stride;
    if (h < cnt) {
      const int src = order[h];
      const int len = fin_len[s * beam + src];
      for (int i = threadIdx.x; i < out_stride; i += blockDim.x)
        dst[i] = i < len ? fin_tok[((size_t)s * beam + src) * stride + i] : -1;
      if (threadIdx.x == 0) {
        out_len[s * beam + h] = len;
        out_score[s * beam + h] = fin_score[s * beam + src];
      }
    } else {
      for (int i = threadIdx.x; i < out_stride; i += blockDim.x) dst[i] = -1;
      if (threadIdx.x == 0) {
        out_len[s * beam + h] = 0;
        out_score[s * beam + h] = -INFINITY;
      }
    }
  }
}

hipError_t launch_beam_output(const int32_t* fin_tok, const int32_t* fin_len, const float* fin_score,
                              const int32_t* fin_count, int n, int beam, int stride, int out_stride,
                              int32_t* out_tok, int32_t* out_len, float* out_score, float* margins,
                              hipStream_t stream) {
  hipLaunchKernelGGL(beam_output_kernel, dim3(n), dim3(128), 0, stream, fin_tok, fin_len, fin_score,
                     fin_count, beam, stride, out_stride, out_tok, out_len, out_score, margins);
  return hipGetLastError();
}

// tok[r] = (int32) src[r * src_stride + col]   (teacher forcing in smi_text_decoder_logits)
__global__ void gather_tokens_kernel(const int64_t* __restrict__ src, int src_stride, int col,
                                     int32_t* __restrict__ tok, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) tok[i] = (int32_t)src[(size_t)i * src_stride + col];
}

hipError_t launch_gather_tokens(const int64_t* src, int src_stride, int col, int32_t* tok, int rows,
                                hipStream_t stream) {
  hipLaunchKernelGGL(gather_tokens_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, src,
                     src_stride, col, tok, rows);
  return hipGetLastError();
}

}  // namespace smi
