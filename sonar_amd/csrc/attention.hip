// Multi-head self-attention over PACKED (pad-free) rows, head_dim = 64.
// Reference: StandardMultiheadAttention + SDPA with key-padding mask,
// wired at sonar/models/sonar_text/factory.py:130-141; scores * Dh^-0.5,
// softmax in fp32, P.V.  With packed rows the key-padding mask becomes the
// per-sentence key range [0, len).
//
// gfx950 mapping: one workgroup = 128 queries of one (sentence, head); each of
// the 4 waves owns 32 queries.  Scores are computed TRANSPOSED
// (S^T = K.Q^T on v_mfma_f32_32x32x16_f16), so a lane holds 16 scores of ONE
// query: the online softmax is lane-local (+1 cross-half shuffle), and the
// fp32 scores convert in-register straight into the B operand of
// O^T = V^T.P^T -- the 16-wide MFMA K-slot order is a free permutation as long
// as V^T is fetched in the same order, so P never moves between lanes or
// through LDS.  K is staged row-major with the 16-B XOR swizzle (conflict-free
// ds_read_b128), V is staged transposed with an 8-B granule swizzle
// (conflict-free ds_read_b64 / ds_write_b16).  S <= 514 in SONAR, so the
// kernel is HBM-bound (~64 flop/B); the score matrix never leaves registers.
#include "common.hpp"
#include "kernels.hpp"

namespace smi {

constexpr int AT_QB = 128;  // queries per workgroup
constexpr int AT_KB = 64;   // keys per K/V tile

// TM: ctx is written in the tile-major GEMM operand layout (common.hpp) -- a wave then stores
// 2 KiB runs (32 rows x 64 B of one k-block) instead of 8-B pieces one row stride apart.
// QTM: qkv is READ tile-major ([T, 3d] as K = 3d blocks, the QKV GEMM's register-direct output):
// the 64-key x 64-B halves of a K or V tile are then contiguous 4 KiB runs.
template <bool TM, bool QTM>
__global__ __launch_bounds__(256) void attention_kernel(const f16* __restrict__ qkv,
                                                        const int32_t* __restrict__ cu,
                                                        f16* __restrict__ ctx, int d, float sl2e) {
  __shared__ __attribute__((aligned(16))) char lds[2 * AT_KB * 128];
  char* Ks = lds;                // [64 keys][128 B], 16-B chunk c of key r at slot c ^ ((r>>1)&7)
  char* Vt = lds + AT_KB * 128;  // [64 dims][128 B], 8-B key granule g of dim r at slot g ^ ((r>>1)&15)

  const int n = blockIdx.x, h = blockIdx.y;
  const int start = cu[n];
  const int len = cu[n + 1] - start;
  const int q0 = blockIdx.z * AT_QB;
  if (q0 >= len) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const size_t ld = (size_t)3 * d;

  const int qi = q0 + wave * 32 + l31;
  const int K3 = 3 * d;
  // element (row r of this sentence, column c of the [q | k | v] row)
  auto at = [&](int r, int c) -> const f16* {
    if constexpr (QTM) return qkv + tm_offset(start + r, c, K3);
    return qkv + (size_t)(start + r) * ld + c;
  };
  half8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const half8*)at(min(qi, len - 1), h * 64 + (ks * 2 + hi) * 8);

  float m = -1e30f, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

  for (int kv0 = 0; kv0 < len; kv0 += AT_KB) {
    __syncthreads();
    // ---- stage K (row-major, swizzled) ----
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int c = tid + 256 * x;
      const int key = c >> 3, slot = c & 7;
      const int chunk = slot ^ ((key >> 1) & 7);
      const int krow = min(kv0 + key, len - 1);
      *(half8*)(Ks + c * 16) = *(const half8*)at(krow, d + h * 64 + chunk * 8);
    }
    // ---- stage V transposed ----
    {
      const int keyl = wave * 16 + (lane & 15);
      const int vrow = min(kv0 + keyl, len - 1);
      const int kg = keyl >> 2, kw = (keyl & 3) * 2;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int dc = (lane >> 4) + 4 * x;
        const half8 v = *(const half8*)at(vrow, 2 * d + h * 64 + dc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int dd = dc * 8 + e;
          *(f16*)(Vt + dd * 128 + ((kg ^ ((dd >> 1) & 15)) << 3) + kw) = v[e];
        }
      }
    }
    __syncthreads();

    // ---- S^T = K . Q^T for 2 blocks of 32 keys ----
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
      const int krow = kb * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 kf = *(const half8*)(Ks + krow * 128 + (((ks * 2 + hi) ^ ((krow >> 1) & 7)) << 4));
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }

    // ---- online softmax (per query = per lane, halves joined by one shuffle) ----
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float v = key < len ? s[kb][r] * sl2e : -INFINITY;
        s[kb][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    m = m_new;
    float psum = 0.f;
    half8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
        psum += p;
        pf[kb][r >> 3][r & 7] = (f16)p;
      }
    lsum = lsum * alpha + psum;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int dd = db * 32 + l31;
      const char* vrow = Vt + dd * 128;
      const int sw = (dd >> 1) & 15;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int g0 = kb * 8 + 4 * u + hi;
          const half4 a0 = *(const half4*)(vrow + ((g0 ^ sw) << 3));
          const half4 a1 = *(const half4*)(vrow + (((g0 + 2) ^ sw) << 3));
          half8 vf;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vf[e] = a0[e];
            vf[e + 4] = a1[e];
          }
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][u], o[db], 0, 0, 0);
        }
    }
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
        if constexpr (TM)
          *(half4*)(ctx + tm_offset(start + qi, h * 64 + db * 32 + 8 * q + 4 * hi, d)) = v;
        else
          *(half4*)(op + db * 32 + 8 * q + 4 * hi) = v;
      }
  }
}

hipError_t launch_attention(const f16* qkv, const int32_t* cu, f16* ctx, int N, int max_len, int d,
                            int heads, hipStream_t stream, int ctx_tm) {
  if (heads <= 0 || d != heads * 64 || N <= 0 || max_len <= 0) return hipErrorInvalidValue;
  const float sl2e = 0.125f * 1.4426950408889634f;  // Dh^-0.5 * log2(e)
  dim3 grid(N, heads, (max_len + AT_QB - 1) / AT_QB);
  // ctx_tm: bit 0 = ctx written tile-major, bit 1 = qkv read tile-major
  switch (ctx_tm & 3) {
    case 0: hipLaunchKernelGGL((attention_kernel<false, false>), grid, dim3(256), 0, stream, qkv, cu, ctx, d, sl2e); break;
    case 1: hipLaunchKernelGGL((attention_kernel<true, false>), grid, dim3(256), 0, stream, qkv, cu, ctx, d, sl2e); break;
    case 2: hipLaunchKernelGGL((attention_kernel<false, true>), grid, dim3(256), 0, stream, qkv, cu, ctx, d, sl2e); break;
    default: hipLaunchKernelGGL((attention_kernel<true, true>), grid, dim3(256), 0, stream, qkv, cu, ctx, d, sl2e); break;
  }
  return hipGetLastError();
}

}  // namespace smi
