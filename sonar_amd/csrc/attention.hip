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
// ds_read_b128), V row-major too and read through the LDS transpose read
// (ds_read_b64_tr_b16); both arrive by double-buffered global->LDS DMA.  S <= 514 in SONAR, so the
// kernel is HBM-bound (~64 flop/B); the score matrix never leaves registers.
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace smi {

constexpr int AT_QB = 128;  // queries per workgroup
constexpr int AT_KB = 64;   // keys per K/V tile

// ds_read_b64_tr_b16 as inline asm (see speech.hip: for the builtin hipcc's LDS-DMA alias tracking waits
// for the next tile's DMA in front of the first read); results are waited for with an lgkmcnt(0) that
// carries them as operands.
template <int OFF>
__device__ __forceinline__ half4 at_tr_read(unsigned lds_addr) {
  half4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
  return v;
}

// TM: ctx is written in the tile-major GEMM operand layout (common.hpp) -- a wave then stores
// 2 KiB runs (32 rows x 64 B of one k-block) instead of 8-B pieces one row stride apart.
// QTM: qkv is READ tile-major ([T, 3d] as K = 3d blocks, the QKV GEMM's register-direct output):
// the 64-key x 64-B halves of a K or V tile are then contiguous 4 KiB runs.
//
// K and V tiles (64 keys x 128 B) go global -> LDS by DMA into two buffers: tile t+1 is in flight while
// tile t is consumed, one barrier per tile, no staging registers.  Both are row-major [key][128 B] with
// 16-B chunk c of key r at slot c ^ swz(r): K swz = (r>>1)&7 (conflict-free ds_read_b128 of the K
// fragments), V swz = ((r>>1)&1)<<2, read with ds_read_b64_tr_b16 -- a 16-lane group fetches a
// [4 keys][16 dims] block and every lane receives the 4 keys of its dim, i.e. the V^T fragment without
// a transposed copy (keys r, r+2 of a group sit in different halves of the 128-B row: all 64 banks).
// NTL: the K / V tiles are loaded non-temporally (launcher: every sentence fits ONE 128-query block, so each qkv line
// is read by exactly one workgroup, once)
template <bool TM, bool QTM, bool NTL = false>
__global__ __launch_bounds__(256, 4) void attention_kernel(const f16* __restrict__ qkv,
                                                        const int32_t* __restrict__ cu,
                                                        f16* __restrict__ ctx, int d, float sl2e, int order) {
  constexpr int TILE = AT_KB * 128;  // 8 KiB
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE];  // [buffer][K | V]

  // order 0: sentence = blockIdx.x, head = blockIdx.y (dispatch order: all sentences of head 0, then head 1, ...).
  // order 1: the heads of a sentence are dispatched together, so the whole 6 KB qkv rows of a sentence are read at
  // about the same time (each head's piece is one 128-B line of the row) instead of in 16 passes over the matrix.
  int n = blockIdx.x, h = blockIdx.y;
  if (order) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    h = lin % (int)gridDim.y;
    n = lin / (int)gridDim.y;
  }
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

  // DMA: instruction x of wave w fills LDS bytes [x*4096 + w*1024, +1024) of a tile: thread ->
  // (key = c>>3, slot = c&7), c = x*256 + tid, fetches the chunk that belongs in that slot
  auto stage = [&](int kv0, int buf) {
    char* kt = lds + buf * 2 * TILE;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int c = tid + 256 * x;
      const int key = c >> 3, slot = c & 7;
      const int row = min(kv0 + key, len - 1);
      if constexpr (NTL) {
        glds16_nt(at(row, d + h * 64 + (slot ^ ((key >> 1) & 7)) * 8), kt + x * 4096 + wave * 1024);
        glds16_nt(at(row, 2 * d + h * 64 + (slot ^ (((key >> 1) & 1) << 2)) * 8), kt + TILE + x * 4096 + wave * 1024);
      } else {
        glds16(at(row, d + h * 64 + (slot ^ ((key >> 1) & 7)) * 8), kt + x * 4096 + wave * 1024);
        glds16(at(row, 2 * d + h * 64 + (slot ^ (((key >> 1) & 1) << 2)) * 8), kt + TILE + x * 4096 + wave * 1024);
      }
    }
  };
  // V^T fragments: lane p of a 16-lane group feeds row (p>>2) of its [4 keys][16 dims] block and gets
  // the 4 keys of dim p.  MFMA (db, kb, u), half `part`: keys 32 kb + 16 u + 4 hi + 8 part + (0..3),
  // dims 32 db + 16 (l31>>4) + (0..15); 32 kb + 16 u + 8 part never changes the swizzle bit, so it is
  // an immediate offset.  (k slot e of the MFMA <-> key 16u + 4hi + 8(e>>2) + (e&3): the S^T layout.)
  unsigned vaddr[2];
  {
    const int p16 = lane & 15, g16 = (lane >> 4) & 1;
    const int key = 4 * hi + (p16 >> 2);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = db * 4 + g16 * 2 + ((p16 & 3) >> 1);
      vaddr[db] = (unsigned)(size_t)(lds + TILE + key * 128 + ((chunk ^ (((key >> 1) & 1) << 2)) << 4) + (p16 & 1) * 8);
    }
  }

  float m = -1e30f, lsum = 0.f;  // m: running reference of the SCALED scores (log2 domain)
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

  stage(0, 0);
  for (int kv0 = 0, t = 0; kv0 < len; kv0 += AT_KB, ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile t has landed for everyone; everyone is done with the other buffer
    if (kv0 + AT_KB < len) stage(kv0 + AT_KB, (t + 1) & 1);
    const char* Ks = lds + (t & 1) * 2 * TILE;

    // The tile's two 32-key blocks are taken one after the other, each with its own online-softmax step: only
    // 16 scores, 8 P values and 16 V^T registers are live at a time (the 64-key version needed 131 VGPRs =
    // 3 workgroups per CU; the kernel is a short latency chain per workgroup, so resident workgroups are what pays).
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // ---- S^T = K . Q^T, 32 keys ----
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
      const int krow = kb * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 kf = *(const half8*)(Ks + krow * 128 + (((ks * 2 + hi) ^ ((krow >> 1) & 7)) << 4));
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sc, 0, 0, 0);
      }
      // ---- online softmax (per query = per lane, halves joined by one shuffle) ----
      if (kv0 + AT_KB > len) {  // only the last tile of a sentence has keys to mask
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= len) sc[r] = -INFINITY;
      }
      float mx = sc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
      {  // the other half of the keys sits 32 lanes away: v_permlane32_swap (VALU) instead of __shfl_xor (an LDS round trip in
         // the middle of the softmax chain, once per key block)
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * sl2e;  // sl2e > 0: the scaled maximum
      }
      // Lazy rescale: the reference m only moves when some query's block maximum exceeds it by more than
      // 2^8; until then p = exp2(x - m) <= 256 (fine in fp32 and for the fp16 P operand) and the 32
      // output accumulators are left alone.
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
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], sl2e, -m));
        psum += p;
        pf[r >> 3][r & 7] = (f16)p;
      }
      lsum += psum;

      // ---- O^T += V^T . P^T ----
      half4 va[2][2][2];  // [db][u][part]
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const unsigned a = vaddr[db] + (t & 1) * 2 * TILE + kb * 4096;
        va[db][0][0] = at_tr_read<0>(a);
        va[db][0][1] = at_tr_read<1024>(a);
        va[db][1][0] = at_tr_read<2048>(a);
        va[db][1][1] = at_tr_read<3072>(a);
      }
      // the wait carries the results as operands: the MFMAs below depend on IT, not just on the reads
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(va[0][0][0]), "+v"(va[0][0][1]), "+v"(va[0][1][0]), "+v"(va[0][1][1]),
                     "+v"(va[1][0][0]), "+v"(va[1][0][1]), "+v"(va[1][1][0]), "+v"(va[1][1][1])
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
            vf[e + 4] = va[db][u][1][e];
          }
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u], o[db], 0, 0, 0);
        }
    }
  }

  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = 1.0f / ltot;
  if constexpr (TM) {
    // Tile-major context: a lane holds 8-B pieces of ONE row, so a direct store scatters 16-B pieces over 16 lines per
    // instruction.  The wave parks its 32 rows x 128 B in LDS (the K / V buffers, free after the barrier) and writes
    // them back as 16 rows x 64 B per instruction = one linear 1 KiB run of a tile-major block, 16 B per lane.
    __syncthreads();  // every wave is done with the K / V tiles
    char* st = lds + wave * 4096;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        half4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (f16)(o[db][q * 4 + e] * inv);
        *(half4*)(st + l31 * 128 + (((db * 4 + q) ^ (l31 & 7)) << 4) + hi * 8) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-local hand-over: the readers below are lanes of this wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hb = i >> 1, row = (i & 1) * 16 + (lane >> 2), slot = lane & 3;
      const int qr = q0 + wave * 32 + row;
      const int rr = (start + qr) & 255;
      const int c4 = slot ^ tm_swz(rr);
      const f32x4 v = *(const f32x4*)(st + row * 128 + (((hb * 4 + c4) ^ (row & 7)) << 4));
      // whole 64-B row segments, read next by the attention-output GEMM on other XCDs: non-temporal (common.hpp)
      if (qr < len) store_nt((f32x4*)(ctx + tm_offset(start + qr, h * 64 + hb * 32 + c4 * 8, d)), v);
    }
  } else if (qi < len) {
    f16* op = ctx + (size_t)(start + qi) * d + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        half4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (f16)(o[db][q * 4 + e] * inv);
        *(half4*)(op + db * 32 + 8 * q + 4 * hi) = v;
      }
  }
}

hipError_t launch_attention(const f16* qkv, const int32_t* cu, f16* ctx, int N, int max_len, int d,
                            int heads, hipStream_t stream, int ctx_tm) {
  if (heads <= 0 || d != heads * 64 || N <= 0 || max_len <= 0) return hipErrorInvalidValue;
  const float sl2e = 0.125f * 1.4426950408889634f;  // Dh^-0.5 * log2(e)
  dim3 grid(N, heads, (max_len + AT_QB - 1) / AT_QB);
  const int order = tune(TUNE_ATT_ORDER, 1);  // ATT_ORDER=0: head-major dispatch (A/B measurements)
  // ctx_tm: bit 0 = ctx written tile-major, bit 1 = qkv read tile-major
  switch (ctx_tm & 3) {
    case 0: hipLaunchKernelGGL((attention_kernel<false, false>), grid, dim3(256), 0, stream, qkv, cu, ctx, d, sl2e, order); break;
    case 1: hipLaunchKernelGGL((attention_kernel<true, false>), grid, dim3(256), 0, stream, qkv, cu, ctx, d, sl2e, order); break;
    case 2: hipLaunchKernelGGL((attention_kernel<false, true>), grid, dim3(256), 0, stream, qkv, cu, ctx, d, sl2e, order); break;
    default:
      if (max_len <= AT_QB)
        hipLaunchKernelGGL((attention_kernel<true, true, true>), grid, dim3(256), 0, stream, qkv, cu, ctx, d, sl2e, order);
      else
        hipLaunchKernelGGL((attention_kernel<true, true, false>), grid, dim3(256), 0, stream, qkv, cu, ctx, d, sl2e, order);
      break;
  }
  return hipGetLastError();
}

}  // namespace smi
