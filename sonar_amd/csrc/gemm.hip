// Dense projection GEMMs of the SONAR text encoder/decoder and conformer layers
// (q/k/v/out projections, FFNs, pointwise convolutions: reference wiring at
// sonar/models/sonar_text/factory.py:130-153 and, for the conformer, fairseq2's
// ConformerBlock built by sonar/models/sonar_speech/factory.py:64-71), fp16 in, fp32
// accumulate on MFMA (v_mfma_f32_16x16x32_f16 in the 256x256 engine, 32x32x16 in the 128x128 one),
// with the epilogues fused.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "gemm_epi.hpp"
#include "gemm_lone.hpp"
#include "gemm_lone16.hpp"
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
// EPI_TANH_F16      : out_h[m][n]  = f16(tanh(acc + bias[n]))
// EPI_RESID_F16     : resid_h[m][n] = f16(float(resid_h[m][n]) + acc + bias[n])   (fp16 residual stream)
// EPI_RESID_HALF_F16: resid_h[m][n] = f16(float(resid_h[m][n]) + 0.5 * (acc + bias[n]))
// EPI_GLU_F16       : out_h[m][g*32+c] = f16(a * sigmoid(b)), a/b = columns g*64+c / g*64+32+c
//                     (W rows interleaved in 32-channel groups at pack time), out width N/2
// bias may be null for every epilogue.
// Epilogue of the 128x128-family engines (gemm_tile.hpp, gemm_lone.hpp): a wave holds MI x NI blocks of 32x32, block
// (ni, mi) element r is C[row0 + mi*32][col0 + ni*32 + 8*(r>>2) + (r&3)] with row0 = tile row + wave row + (lane&31),
// col0 = tile column + wave column + 4*(lane>>5).  b[ni][q]: the bias of the lane's 4-column runs, loaded by the caller
// (before the K loop where possible: a load issued here would sit between the stores -- vmcnt does not tell loads from
// stores, so every wait for a bias value also waited for all the stores before it; the read-modify-write epilogues
// likewise read ALL old values first).
template <int EPI, int LAYOUT, int MI, int NI>
__device__ __forceinline__ void gt_epilogue(const f32x16 (&acc)[NI][MI], const f32x4 (&b)[NI][4], void* __restrict__ out,
                                            int row0, int col0, int glu_col0, int N, int ldo) {
  if constexpr (EPI == EPI_GLU_F16) {
    static_assert(NI == 2, "GLU pairs the two column blocks of a wave");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oc = glu_col0 + 8 * q;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = row0 + mi * 32;
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          h[e] = (f16)((acc[0][mi][q * 4 + e] + b[0][q][e]) * sigmoid_f(acc[1][mi][q * 4 + e] + b[1][q][e]));
        *(half4*)((f16*)out + (size_t)m * ldo + oc) = h;
      }
    }
  } else if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_RESID_HALF_F32) {
    f32x4 old[NI][4][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          old[ni][q][mi] = *(const f32x4*)((const float*)out + (size_t)(row0 + mi * 32) * ldo + col0 + ni * 32 + 8 * q);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][q * 4 + e] + b[ni][q][e];
          if constexpr (EPI == EPI_RESID_HALF_F32) v = v * 0.5f;
          *(f32x4*)((float*)out + (size_t)(row0 + mi * 32) * ldo + col0 + ni * 32 + 8 * q) = old[ni][q][mi] + v;
        }
  } else if constexpr (EPI == EPI_RESID_F16 || EPI == EPI_RESID_HALF_F16) {
    half4 old[NI][4][MI];
    auto at = [&](int ni, int q, int mi) {
      const int m = row0 + mi * 32, n = col0 + ni * 32 + 8 * q;
      return LAYOUT == 3 ? (f16*)out + tm_offset(m, n, N) : (f16*)out + (size_t)m * ldo + n;
    };
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) old[ni][q][mi] = *(const half4*)at(ni, q, mi);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          half4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[ni][mi][q * 4 + e] + b[ni][q][e];
            if constexpr (EPI == EPI_RESID_HALF_F16) v *= 0.5f;
            h[e] = (f16)((float)old[ni][q][mi][e] + v);
          }
          *(half4*)at(ni, q, mi) = h;
        }
  } else {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int m = row0 + mi * 32, n = col0 + ni * 32 + 8 * q;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][q * 4 + e] + b[ni][q][e];
          if constexpr (EPI == EPI_STORE_F32) {
            *(f32x4*)((float*)out + (size_t)m * ldo + n) = v;
          } else {
            const half4 h = epi_act_pack<EPI>(v);
            if constexpr (LAYOUT == 2)
              *(half4*)((f16*)out + tm_offset(m, n, N)) = h;
            else
              *(half4*)((f16*)out + (size_t)m * ldo + n) = h;
          }
        }
  }
}

// the lane's bias values: 4-column runs at col0 + ni*32 + 8*q (zeros without a bias)
template <int NI>
__device__ __forceinline__ void gt_load_bias(f32x4 (&b)[NI][4], const float* __restrict__ bias, int col0) {
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      b[ni][q] = bias ? *(const f32x4*)(bias + col0 + ni * 32 + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
}

// LAYOUT: 0 = row-major operands and output; 1 = tile-major X and W (common.hpp), row-major
// output; 2 = tile-major X, W and fp16 output (the output is the next GEMM's X, its K = N);
// 3 = tile-major X, W and a tile-major fp16 RESIDUAL STREAM that is read-modified-written
// (EPI_RESID_F16 only: the text encoder's x, so that the residual epilogue needs no LDS staging).
// RING: 0 = the two-stage loop, 2 workgroups per CU (launches with more tiles than CUs); 4 = the counted-wait
// ring of gemm_tile.hpp with that many stages, 1 workgroup per CU (round 3's lone-tile path, kept for A/B runs:
// SMI_LONE=0; the default for launches whose tiles all fit on the chip at once is gemm_lone_kernel below).
template <int EPI, int LAYOUT = 0, int RING = 0>
__global__ __launch_bounds__(GT_THREADS, RING ? 1 : 2) void gemm_tn_kernel(const f16* __restrict__ X,
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
  if constexpr (EPI == EPI_BIAS_F16) {
    if (ksplit > 1) fp16_saturate_on();  // fp16 split-K partial sums saturate instead of overflowing to inf
  }
  if constexpr (RING)
    gt_mainloop_ring<(LAYOUT > 0), RING>(acc, X, W, K, m0, n0, smem, kz * klen, klen);
  else
    gt_mainloop<(LAYOUT > 0)>(acc, X, W, K, m0, n0, smem, kz * klen, klen);
  if (kz > 0) {
    bias = nullptr;
    out = (char*)out + (size_t)kz * part_stride;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int col0 = n0 + wn * 64 + 4 * hi;
  f32x4 b[2][4];
  gt_load_bias<2>(b, bias, col0);
  gt_epilogue<EPI, LAYOUT, 2, 2>(acc.v, b, out, m0 + wm * 64 + (lane & 31), col0, n0 / 2 + wn * 32 + 4 * hi, N, ldo);
}

// The lone-tile engine (gemm_lone.hpp): unit shape BM x BN (64x64 is the one the launcher uses).  Same arguments and
// epilogues as gemm_tn_kernel (EPI_GLU_F16: BN = 128 only).
template <int EPI, int LAYOUT, int BM, int BN>
__global__ __launch_bounds__(GT_THREADS, (LoneShape<BM, BN>::WG_PER_CU)) void gemm_lone_kernel(const f16* __restrict__ X, const f16* __restrict__ W,
                                                                  const float* __restrict__ bias, void* __restrict__ out,
                                                                  int M, int N, int K, int ldo, int ksplit,
                                                                  size_t part_stride) {
  using S = LoneShape<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LONE_TRACE(5);  // kernel entry
  int tile_m, tile_n;
  lone_tile_coords(M / BM, N / BN, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz = blockIdx.y;
  const int klen = K / ksplit;
  if constexpr (EPI == EPI_BIAS_F16) {
    if (ksplit > 1) fp16_saturate_on();  // fp16 split-K partial sums saturate instead of overflowing to inf
  }
  if (kz > 0) {
    bias = nullptr;
    out = (char*)out + (size_t)kz * part_stride;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int col0 = n0 + wn * (BN / 2) + 4 * hi;
  f32x4 b[S::NI][4];
  gt_load_bias<S::NI>(b, bias, col0);  // in flight under the pipeline fill
  f32x16 acc[S::NI][S::MI];
  lone_mainloop<(LAYOUT > 0), BM, BN, S::NI, S::MI>(acc, X, W, K, m0, n0, smem, kz * klen, klen);
  gt_epilogue<EPI, LAYOUT, S::MI, S::NI>(acc, b, out, m0 + wm * (BM / 2) + (lane & 31), col0,
                                         n0 / 2 + wn * 32 + 4 * hi, N, ldo);
  LONE_TRACE(4);  // stores issued
}

// The k-sliced lone-tile unit (gemm_lone16.hpp): tile-major X and W, 64x64 units, no LDS in the K loop.  EPI_BIAS_F16 /
// EPI_RELU_F16 (fp16 output, tile-major when OUT_TM, else row-major with ldo) and EPI_STORE_F32 (fp32 row-major: split-K
// slabs).  blockIdx.y = K part (EPI_STORE_F32), each part K / ksplit = 128 * NKB columns.
template <int EPI, bool OUT_TM, int NKB>
__global__ __launch_bounds__(L16_THREADS, 2) void gemm_lone16_kernel(const f16* __restrict__ X, const f16* __restrict__ W,
                                                                     const float* __restrict__ bias, void* __restrict__ out,
                                                                     int M, int N, int K, int ldo, size_t part_stride) {
  static_assert(EPI == EPI_BIAS_F16 || EPI == EPI_RELU_F16 || EPI == EPI_STORE_F32, "epilogues of the k-sliced unit");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile_m, tile_n;
  lone_tile_coords(M / L16_BM, N / L16_BN, tile_m, tile_n);
  const int m0 = tile_m * L16_BM, n0 = tile_n * L16_BN;
  const int kz = blockIdx.y;
  if constexpr (EPI == EPI_BIAS_F16) {
    if (gridDim.y > 1) fp16_saturate_on();  // fp16 split-K partial sums saturate instead of overflowing to inf
  }
  if (kz > 0) {
    bias = nullptr;
    out = (char*)out + (size_t)kz * part_stride;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  // the lane's bias values (its 4-column runs), requested in front of the operand stream through a pointer that falls back to
  // a valid address: a conditional load is a branch, and hipcc waits for the loads inside it (vmcnt(0)) before it goes on
  const float* bp = bias ? bias + n0 + 4 * kg : (const float*)W + 4 * kg;
  f32x4 b[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) b[ni] = *(const f32x4*)(bp + ni * 16);
  f32x4 acc[4][4];
  lone16_mainloop<NKB>(acc, X, W, K, m0, n0, kz * 128 * NKB);
  if (!bias) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 v[4];
  lone16_reduce(acc, v, smem);
  const int row = m0 + wave * 16 + l15;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + ni * 16 + 4 * kg;
    const f32x4 o = v[ni] + b[ni];
    if constexpr (EPI == EPI_STORE_F32) {
      *(f32x4*)((float*)out + (size_t)row * ldo + n) = o;
    } else {
      const half4 h = epi_act_pack<EPI>(o);
      if constexpr (OUT_TM)
        *(half4*)((f16*)out + tm_offset(row, n, N)) = h;
      else
        *(half4*)((f16*)out + (size_t)row * ldo + n) = h;
    }
  }
}

// LDS accesses the compiler must NOT see as LDS accesses: after the next tile's LDS-DMA pipeline fill has been issued,
// hipcc guards every ordinary LDS load and store with `s_waitcnt vmcnt(0)` (its alias tracking cannot tell the ring slots
// the fill is landing in from the staging buffers / constant area next to them), which drains the fill and serialises the
// epilogue's global stores.  The staging buffers are ring slot 3 and the LDS above the ring -- never a target of the fill
// -- so nothing has to be waited for.  Stores complete at the lgkmcnt(0) of SMI_LGKM0_BARRIER, which every staged pass
// runs before anyone reads; loads are issued AND waited for in one asm statement (lds_read_stage).
__device__ __forceinline__ void lds_write_f4_asm(void* p, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v) : "memory");
}
template <typename T8>
__device__ __forceinline__ void lds_write_b64_asm(void* p, T8 v) {
  static_assert(sizeof(T8) == 8, "8-byte value");
  asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v) : "memory");
}
// N 16-B reads and their wait in ONE statement (issue and wait must not be separable: between them the compiler may
// copy a destination register that the LDS has not written yet -- r03 experiment 4, attempt 4).  Used by the staged
// epilogues, whose ordinary LDS loads each drew a vmcnt(0): the staging buffers are ring slot 3 and the LDS above the
// ring, never a target of the next tile's fill, so nothing has to be waited for.
template <int N>
__device__ __forceinline__ void lds_read_stage(const char* const (&p)[N], f32x4 (&v)[N]) {
  static_assert(N == 2 || N == 4, "2 or 4 reads");
  if constexpr (N == 4)
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"((unsigned)(size_t)p[0]), "v"((unsigned)(size_t)p[1]), "v"((unsigned)(size_t)p[2]), "v"((unsigned)(size_t)p[3])
                 : "memory");
  else
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1])
                 : "v"((unsigned)(size_t)p[0]), "v"((unsigned)(size_t)p[1])
                 : "memory");
}

// Same epilogues on the 256x256 ping-pong tile engine (gemm_tile256.hpp).
#ifdef SMI_GEMM_TRACE
// development aid (-DSMI_GEMM_TRACE): per-tile phase timestamps (100 MHz wall clock) of thread 0
// of the first 16 workgroups; read back with smi_debug_gemm_trace()
__device__ unsigned long long g2_trace_buf[16 * 64 * 8];
#define G2_TRACE(slot)                                                               \
  if (threadIdx.x == 0 && blockIdx.x < 16 && trace_i < 64)                           \
  g2_trace_buf[(blockIdx.x * 64 + trace_i) * 8 + (slot)] = wall_clock64()
#else
#define G2_TRACE(slot)
#endif

// PERSISTENT: the grid is one workgroup per CU (160 KiB of LDS) and every workgroup walks tiles
// id = round * gridDim + xcd_remap(block).  The pipeline fill of tile i+1 is issued before the
// epilogue of tile i, and the epilogue leaves through two small staging buffers outside ring slots
// 0..2 (gemm_tile256.hpp), so the ~2 us fill latency of the 32-slice (K = 1024) tiles is hidden.
// -DSMI_G2_STORE_OVERLAP=1: per-tile constants by LDS-DMA + the previous tile's stores left in flight under the first two K slices
// (round 5 experiment 1: QKV -1.2 %, FFN-inner +2 %, C2 step +0.65 % -- NEGATIVE, the default stays 0; profiles/r05_experiments.txt)
#ifndef SMI_G2_STORE_OVERLAP
#define SMI_G2_STORE_OVERLAP 0
#endif
constexpr bool G2_STORE_OVERLAP = SMI_G2_STORE_OVERLAP != 0;
// -DSMI_G2_OVERLAP_BIAS_TM=1: the same for the <EPI_BIAS_F16, tile-major out> instantiation only (the encoder's QKV projection,
// which gained 1.4 % in experiment 1, and the decoder's logits GEMM)
#ifndef SMI_G2_OVERLAP_BIAS_TM
#define SMI_G2_OVERLAP_BIAS_TM 0
#endif
constexpr bool G2_OVERLAP_BIAS_TM = SMI_G2_OVERLAP_BIAS_TM != 0;
// -DSMI_STATS_FUSED=0: the round-4 statistics epilogue of the fp16 logits GEMM (separate statistics and store passes) -- A/B builds
#ifndef SMI_STATS_FUSED
#define SMI_STATS_FUSED 1
#endif
constexpr bool G2_STATS_FUSED = SMI_STATS_FUSED != 0;

template <int EPI, int LAYOUT = 0>
__global__ __launch_bounds__(G2_THREADS) void gemm_tn256_kernel(const f16* __restrict__ X,
                                                                const f16* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                void* __restrict__ out_, int M, int N,
                                                                int K, int ldo, GemmTileStats stats, int ksplit,
                                                                size_t part_stride, int raster, GemmLnFold fold) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool TM = LAYOUT > 0;
  // LayerNorm fold (kernels.hpp: GemmLnFold).  CONSUMER: fp16 tile-major epilogues whose X operand is the residual stream
  // itself; PRODUCER: the tile-major residual epilogue, which leaves the row sums of the stream it writes.
  // (round 4: also the SiLU epilogue, and the ROW-MAJOR fp16 outputs of tile-major operands -- LAYOUT 1, bias and GLU: the
  // conformer's fused QKV and pointwise_conv1 feed per-clip kernels that read rows -- in the centred variant only)
  constexpr bool FOLD_CONSUMER_ST = LAYOUT == 1 && (EPI == EPI_BIAS_F16 || EPI == EPI_GLU_F16);  // staged epilogues
  constexpr bool FOLD_CONSUMER = (LAYOUT == 2 && (EPI == EPI_BIAS_F16 || EPI == EPI_RELU_F16 || EPI == EPI_SILU_F16)) ||
                                 FOLD_CONSUMER_ST;
  constexpr bool FOLD_PRODUCER = LAYOUT == 3;
  // the exact variant (epilogue term "- mean * c1") exists for the text encoder's bias / relu consumers only: the later
  // consumers are centred-only, which spares them the c1 slice's registers (the SiLU kernel spilled with it)
  constexpr bool FOLD_EXACT = LAYOUT == 2 && (EPI == EPI_BIAS_F16 || EPI == EPI_RELU_F16);
  const bool folded = FOLD_CONSUMER && fold.part_in != nullptr;
  // work unit = (tile, K part kz): split-K (EPI_STORE_F32 only) gives each part its own fp32 output slab
  // at out + kz * part_stride bytes; the bias goes into part 0; the consumer sums the slabs.
  // K part kz of ksplit covers the K slices [S kz / ksplit, S (kz + 1) / ksplit), S = K / 32: the parts need not be
  // equal (the decoder's FFN output projection runs 12 parts of 21-22 slices: 240 units on 256 CUs)
  const int nslices = K / G2_BK;
  auto kpart = [&](int kz_, int& s0, int& ns) {
    s0 = (int)((long long)nslices * kz_ / ksplit);
    ns = (int)((long long)nslices * (kz_ + 1) / ksplit) - s0;
  };
  const int ntm = M / G2_BM, ntn = N / G2_BN, ntiles = ntm * ntn * ksplit;
  void* out = out_;
  if constexpr (EPI == EPI_BIAS_F16 && LAYOUT <= 1) {
    if (ksplit > 1) fp16_saturate_on();  // fp16 split-K partial sums saturate instead of overflowing to inf
  }
#ifdef SMI_GEMM_TRACE
  int trace_i = 0;
  G2_TRACE(5);  // kernel entry
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // accumulator layout (gemm_tile256.hpp): lane -> row l15 of a 16-row block, 4 consecutive columns
  // at 4*kg of a 16-column block
  const int l15 = lane & 15, kg = lane >> 4, wr = wave >> 2, wc = wave & 3;
  const int hi = lane >> 5;  // read-out row parity of the staged epilogues

  // The bias slice of a tile (256 floats) waits in the second staging buffer: it is fetched with
  // the pipeline fill of its tile (no exposed latency, no registers held across the K loop) and
  // read back with ds_reads, which do not touch the vmcnt queue the next tile's fill sits in.
  float* bias_lds = (float*)g2_stage(smem, 1);
  const int tid = threadIdx.x;
  auto fetch_bias = [&](int n0, int kz) {  // one value per thread (tid < 256): ONE register held across the epilogue
    float v = 0.f;
    if constexpr (!((G2_STORE_OVERLAP && (LAYOUT == 2 || LAYOUT == 3)) || (G2_OVERLAP_BIAS_TM && EPI == EPI_BIAS_F16 && LAYOUT == 2))) {
      if (bias && kz == 0 && tid < 256) v = bias[n0 + tid];
    }
    return v;
  };
  // fold consumer: the tile's c1 slice, and (mean, rstd) of its 256 rows from the producers' partial sums -- fetched with
  // the pipeline fill like the bias, parked in LDS next to it (floats 256..511; half sums at 1536..), zeros at 1024.. for the
  // accumulators' start value (the bias slot holds c2, which is added AFTER the row scaling)
  float* c1_lds = bias_lds + 256;
  // Round 5 -- register-direct epilogues (LAYOUT 2 / 3): the per-tile constants travel by LDS-DMA with the tile's pipeline fill
  // (DMA_CONST) instead of through registers.  A value handed over in a VGPR (`bias_lds[tid] = bias_next` at the tile start)
  // made hipcc drain the vector-memory queue there (s_waitcnt vmcnt(0): at a loop head its count of younger operations is
  // pessimistic), and with it the previous tile's 16 output stores -- 2.7 us of a 26.6 us K = 1024 tile (round-2
  // experiment 26) that can run under the next tile's first K slices instead (gemm_tile256.hpp: pend16).  The constants
  // are then READ with asm loads (issue + wait in one statement), which hipcc's LDS-DMA alias tracking does not guard.
  constexpr bool DMA_CONST = (G2_STORE_OVERLAP && (LAYOUT == 2 || LAYOUT == 3)) ||
                             (G2_OVERLAP_BIAS_TM && EPI == EPI_BIAS_F16 && LAYOUT == 2);
  float* zero_lds = bias_lds + (DMA_CONST ? 7936 : 1024);   // DMA_CONST: clear of the statistics / row-sum scratch below
  float* statraw_lds = bias_lds + 4096;                     // DMA_CONST fold consumer: [4 partials][256 rows] (sum, sum of squares)
  float2* rowsum_lds = (float2*)(bias_lds + 1536);  // fold producer: [4 column waves][256 rows]
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue_consts = [&](int m0n, int n0n, int kzn) {
    if constexpr (DMA_CONST) {
      if (bias && kzn == 0 && wave_u < 4) glds4(bias + n0n + tid, bias_lds + wave_u * 64);
      if (folded) {
        const float* ps = (const float*)fold.part_in + (size_t)m0n * 2 + tid;   // 512 floats per partial, one per thread
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
          if (pp < fold.nparts) glds4(ps + (size_t)pp * M * 2, statraw_lds + pp * 512 + wave_u * 64);
      }
    }
  };
  auto fetch_c1 = [&](int n0) {
    float v = 0.f;
    if constexpr (FOLD_EXACT) {
      if (folded && !fold.centered && tid < 256) v = fold.c1[n0 + tid];
    }
    return v;
  };
  // Raw partial sums only -- NO arithmetic here: anything computed from the loads would put their s_waitcnt in front of
  // the previous tile's epilogue and expose a full memory latency per tile (measured: +8..11 % on the consuming GEMMs).
  // All 512 threads take part: thread t fetches partials (t >> 8), (t >> 8) + 2 of row t & 255 (nparts <= 4).
  struct RawStat { float2 a, b; };
  auto fetch_rowstat = [&](int m0) {
    RawStat r{{0.f, 0.f}, {0.f, 0.f}};
    if (!DMA_CONST && folded) {
      const int p = tid >> 8, row = m0 + (tid & 255);
      if (p < fold.nparts) r.a = fold.part_in[(size_t)p * M + row];
      if (p + 2 < fold.nparts) r.b = fold.part_in[(size_t)(p + 2) * M + row];
    }
    return r;
  };
  float2* stat_lds = (float2*)(bias_lds + 1536);  // [2][256] half sums, joined after the barrier of g2_begin
  if ((FOLD_CONSUMER || DMA_CONST) && tid < 256) zero_lds[tid] = 0.f;  // published by the barrier of the first g2_begin

  // unit id = kz * (ntm * ntn) + output tile: neighbouring ids share operand panels of one K part
  const int nout = ntm * ntn;
  // raster 0: grouped 8(m) x ntn super-tiles in id order (gemm_tile256.hpp).
  // raster 2 (full 256-workgroup grids, N >= 4 n-quads): XCD c OWNS the m-groups c, c + 8, ... (8 X panels each) and
  // walks the n-quads of a group in consecutive rounds, in an order rotated by c.  With raster 0 and N = 8192 the
  // 8 XCDs work on the SAME 8 X panels in every round (each on its own 4 W panels): every X panel is pulled across
  // the fabric by all 8 XCDs at the same moment (PMC: 3.2 GB fetched per launch for 0.29 GB of operands).  With
  // XCD-owned m-groups an X panel is fetched by one XCD only, and the rotation keeps the XCDs on different W panels.
  // Measured (profiles/r02_experiments.txt, experiments 6-7): FFN inner 1.82-1.87 -> 1.75-1.76 ms
  // (1175-1210 -> 1249-1260 TFLOP/s), 44.0 -> 41.9 ms per C2 step; no effect at N = 3072 (X is shared by 3 XCDs
  // there), so it is used from 16 n tiles up.  (raster 1 = the same without the rotation.)
  // Virtual ids t = 256 q + 32 c + j (round q, XCD c, slot j); the last m-group may be partial: its surplus slots,
  // and XCDs that own one group fewer, skip the id.
  const int nq = ntn / 4;
  const int nvirt = raster ? ((ntm + 63) / 64) * nq * 256 : ntiles;
  auto coords = [&](int t, int& tm_, int& tn_) -> bool {
    if (raster == 0) {
      g2_tile_coords_of(t % nout, ntm, ntn, tm_, tn_);
      return true;
    }
    const int q = t / 256, c = (t % 256) / 32, j = t % 32;
    tm_ = (c + 8 * (q / nq)) * 8 + j % 8;
    tn_ = ((q + (raster == 2 ? c : 0)) % nq) * 4 + j / 8;
    return tm_ < ntm;
  };
  int tile_m = 0, tile_n = 0;
  // first valid id of this workgroup at or after t (stride = grid size)
  auto seek = [&](int t) {
    while (t < nvirt && !coords(t, tile_m, tile_n)) t += gridDim.x;
    return t;
  };
  int tile = seek(xcd_remap(blockIdx.x, gridDim.x));
  if (tile >= nvirt) return;
  int kz = raster ? 0 : tile / nout;
  int ks0, nt;
  kpart(kz, ks0, nt);
  G2Src src = g2_make_src<TM, TM>(X, W, K, tile_m * G2_BM, tile_n * G2_BN, ks0 * G2_BK);
  // The small per-tile fetches are issued BEFORE the pipeline fill: their destination registers carried the previous
  // tile's values, and hipcc guards the overwrite (zero-init + conditional load) with `s_waitcnt vmcnt(0)` -- which, placed
  // after the LDS-DMA issue, waited for the whole fill in front of the epilogue (+1.3 us per K = 1024 tile, measured).
  float bias_next = fetch_bias(tile_n * G2_BN, kz);
  float c1_next = fetch_c1(tile_n * G2_BN);
  RawStat stat_next = fetch_rowstat(tile_m * G2_BM);
  issue_consts(tile_m * G2_BM, tile_n * G2_BN, kz);
  g2_prefetch(src, nt, smem);

  bool first_tile = true;  // no stores of a previous tile in the queue
  while (tile < nvirt) {
    const int m0 = tile_m * G2_BM, n0 = tile_n * G2_BN;
    const int tile_n_cur = tile_n;
    (void)tile_n_cur;
    out = (char*)out_ + (size_t)kz * part_stride;
    G2_TRACE(0);
    if constexpr (!DMA_CONST) {
      if (tid < 256) bias_lds[tid] = bias_next;
    }
    if constexpr (FOLD_CONSUMER_ST) {
      // the staged epilogue of the previous tile went through this staging buffer: the accumulators' zero start value
      // has to be written again (the register-direct tile-major epilogues never touch it)
      if (folded && tid < 256) zero_lds[tid] = 0.f;
    }
    if (folded) {
      if constexpr (FOLD_EXACT) {
        if (!fold.centered && tid < 256) c1_lds[tid] = c1_next;
      }
      if constexpr (!DMA_CONST) stat_lds[tid] = float2{stat_next.a.x + stat_next.b.x, stat_next.a.y + stat_next.b.y};
    }
    GemmTile256Acc acc;
    // register-direct epilogues (tile-major fp16 store, tile-major residual read-modify-write): 16 stores per lane and tile
    // (plus, on some waves, a row-sum / statistics store), no LDS staging -- they may overlap the next tile's first two K slices
    // (gemm_tile256.hpp: pend16).  Needs >= 5 K slices per unit (the counted waits of iterations 0, 1 assume that slices 3, 4
    // are issued there) and a previous tile of this workgroup.
    const bool pend16 = DMA_CONST && nt >= 5 && !first_tile;
    if constexpr (DMA_CONST) {
      // g2_begin with the start values read by an asm load: kz > 0 parts of a split-K launch and bias-less GEMMs start from
      // zeros, a fold consumer too (its bias slot holds c2, added after the row scaling)
      if (pend16)
        SMI_WAIT_VMCNT(16);
      else
        SMI_WAIT_VMCNT(0);
      SMI_LGKM0_BARRIER();  // slices 0..2 and the constants complete for everyone
      const float* isrc = (folded || !bias || kz != 0) ? zero_lds : bias_lds;
      const char* ip[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) ip[ni] = (const char*)(isrc + wc * 64 + ni * 16 + 4 * kg);
      f32x4 iv[4];
      lds_read_stage<4>(ip, iv);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) acc.v[ni][mi] = iv[ni];
    } else {
      g2_begin(acc, folded ? zero_lds : bias_lds, false);
    }
    // Fold consumer: what the epilogue needs stays in REGISTERS, spread over the wave, and is fetched there with
    // ds_bpermute (a lane crossbar, no LDS memory access): after the next tile's LDS-DMA fill has been issued, hipcc puts
    // `s_waitcnt vmcnt(0)` in front of every LDS load (its alias tracking cannot tell the ring from the constant area),
    // which drained the fill and serialised the epilogue's stores (+9 % on the FFN-inner GEMM).  Lane L of a wave keeps
    // (rstd, -rstd * mean) of the rows wr*128 + L and + 64 + L of the tile, and c1 / c2 of column wc*64 + L.
    // Only the LDS reads happen here (nothing may read LDS once the next fill is in flight); the arithmetic on them
    // waits for the epilogue, where VALU work hides behind the stores -- here it would sit, serial, in front of the K loop.
    float2 row_sq[2] = {{0.f, 0.f}, {0.f, 0.f}};  // (sum, sum of squares) of the lane's two rows
    float col_c1 = 0.f, col_c2 = 0.f;
    if (folded) {
      if constexpr (DMA_CONST) {
        // the raw partials of the lane's two rows, summed in the order of the register path: (p0 + p2) + (p1 + p3)
        float2 pr[2][4];
        float c2v;
        asm volatile(
            "ds_read_b64 %0, %9\n\tds_read_b64 %1, %9 offset:2048\n\tds_read_b64 %2, %9 offset:4096\n\t"
            "ds_read_b64 %3, %9 offset:6144\n\tds_read_b64 %4, %9 offset:512\n\tds_read_b64 %5, %9 offset:2560\n\t"
            "ds_read_b64 %6, %9 offset:4608\n\tds_read_b64 %7, %9 offset:6656\n\tds_read_b32 %8, %10\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(pr[0][0]), "=&v"(pr[0][1]), "=&v"(pr[0][2]), "=&v"(pr[0][3]), "=&v"(pr[1][0]), "=&v"(pr[1][1]),
              "=&v"(pr[1][2]), "=&v"(pr[1][3]), "=&v"(c2v)
            : "v"((unsigned)(size_t)(statraw_lds + (wr * 128 + lane) * 2)), "v"((unsigned)(size_t)(bias_lds + wc * 64 + lane))
            : "memory");
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float2 z = {0.f, 0.f};
          const float2 p0 = pr[u][0], p1 = fold.nparts > 1 ? pr[u][1] : z, p2 = fold.nparts > 2 ? pr[u][2] : z,
                       p3 = fold.nparts > 3 ? pr[u][3] : z;
          row_sq[u] = float2{(p0.x + p2.x) + (p1.x + p3.x), (p0.y + p2.y) + (p1.y + p3.y)};
        }
        col_c2 = c2v;
        if constexpr (FOLD_EXACT) {
          if (!fold.centered) col_c1 = c1_lds[wc * 64 + lane];
        }
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = wr * 128 + u * 64 + lane;
          const float2 h0 = stat_lds[row], h1 = stat_lds[256 + row];
          row_sq[u] = float2{h0.x + h1.x, h0.y + h1.y};
        }
        if constexpr (FOLD_EXACT) col_c1 = c1_lds[wc * 64 + lane];
        col_c2 = bias_lds[wc * 64 + lane];
      }
    }
    G2_TRACE(1);
    g2_mainloop(acc, src, nt, smem, pend16);
    first_tile = false;
    G2_TRACE(2);
    tile = seek(tile + (int)gridDim.x);  // (tile_m, tile_n) now name the NEXT tile; m0 / n0 keep this one
    if (tile < nvirt) {  // fill for the next tile, behind this tile's epilogue
      kz = raster ? 0 : tile / nout;
      kpart(kz, ks0, nt);
      src = g2_make_src<TM, TM>(X, W, K, tile_m * G2_BM, tile_n * G2_BN, ks0 * G2_BK);
      bias_next = fetch_bias(tile_n * G2_BN, kz);  // before the fill: see above
      c1_next = fetch_c1(tile_n * G2_BN);
      stat_next = fetch_rowstat(tile_m * G2_BM);
      issue_consts(tile_m * G2_BM, tile_n * G2_BN, kz);
      g2_prefetch(src, nt, smem);
    }
    G2_TRACE(3);
    // epilogues that leave through the fp32 staging passes (fp32 outputs and the fp16 residual stream,
    // whose read-modify-write adds in fp32 and rounds once)
    constexpr bool F16_RESID = EPI == EPI_RESID_F16 || EPI == EPI_RESID_HALF_F16;
    constexpr bool HALF_STEP = EPI == EPI_RESID_HALF_F32 || EPI == EPI_RESID_HALF_F16;
    constexpr bool F32_OUT = EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32 || EPI == EPI_RESID_HALF_F32 || F16_RESID;
    // staged epilogues: pass p holds tile rows wr*128 + p*32 + (0..31) of both row groups =
    // accumulator blocks mi = 2p, 2p+1; a lane writes staging row lr_w(mi)
    auto lr_w = [&](int mi) { return wr * 32 + (mi & 1) * 16 + l15; };
    // Fold consumer with a STAGED epilogue (row-major fp16 output): the centred LayerNorm fold applied to the accumulators in
    // place, before the activation and the staging passes -- out = rstd(row) * acc + c2(column).  Like the tile-major store
    // (below) it takes everything from registers through ds_bpermute: lane L holds (sum, sum of squares) of rows wr*128 + L and
    // + 64 + L and c2 of column wc*64 + L.  128 v_pk_fma_f32 per lane and tile.
    auto fold_affine_st = [&](GemmTile256Acc& a) {
      float row_rs[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float mean = row_sq[u].x * fold.inv_k;
        const float var = fmaxf(row_sq[u].y * fold.inv_k - mean * mean, 0.f);
        row_rs[u] = __builtin_amdgcn_rsqf(var + fold.eps);
      }
      float rsall[8];
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
        rsall[mi] = __int_as_float(__builtin_amdgcn_ds_bpermute(((mi & 3) * 16 + l15) * 4, __float_as_int(row_rs[mi >> 2])));
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        f32x4 c2v;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          c2v[r] = __int_as_float(__builtin_amdgcn_ds_bpermute((ni * 16 + 4 * kg + r) * 4, __float_as_int(col_c2)));
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const f32x2 rs2 = {rsall[mi], rsall[mi]};
          f32x4 v = a.v[ni][mi];
#pragma unroll
          for (int hp2 = 0; hp2 < 2; ++hp2) {
            const f32x2 c2p = {c2v[2 * hp2], c2v[2 * hp2 + 1]};
            f32x2 vp = {v[2 * hp2], v[2 * hp2 + 1]};
            vp = __builtin_elementwise_fma(rs2, vp, c2p);
            v[2 * hp2] = vp[0];
            v[2 * hp2 + 1] = vp[1];
          }
          a.v[ni][mi] = v;
        }
      }
    };
    (void)fold_affine_st;

    // (round 4: also with the tile-major fp16 store -- an fp16 model's logits are fp16 in the reference, the statistics are
    // then taken of the ROUNDED values, i.e. of exactly what the selection kernels will read)
    constexpr bool STATS_F16 = EPI == EPI_BIAS_F16 && LAYOUT == 2;
    bool stats_stored = false;  // the fused statistics + store pass below has written the tile
    (void)stats_stored;
    if constexpr (STATS_F16 && G2_STATS_FUSED) {
      // Round 5: statistics of the fp16 logits and their tile-major store in ONE pass over the accumulators, every tile but the
      // last column tile (whose padding columns need masking: the general code below).  Counted in the ISA, the round-4
      // statistics cost ~12 VALU issue slots per element (128 elements per lane and tile): two conversions (round to fp16 and
      // back), a multiply by scale * log2 e, TWO v_cndmask of the valid-column mask (the runtime `!full` test had been turned
      // into selects on every element of every tile), max, subtract, v_exp_f32 (quarter rate), add -- and the store then
      // converted every value again.  Here: ONE packed conversion shared by the store and the statistics (v_cvt_pk_f16_f32),
      // the unpack, the maximum of the RAW rounded values (scale > 0: max and scaling commute, bit for bit), the scaled
      // exponent as one v_pk_fma_f32 per two values, packed adds: ~8 slots per element, and the 16 stores of a wave leave
      // spread over the pass instead of as one burst behind it.
      // (stats.scale > 0 is what lets the maximum be taken before the scaling; any other scale takes the general code below)
      if (stats.tile_max && n0 + G2_BN <= stats.valid_n && !folded && stats.scale > 0.f) {
        stats_stored = true;
        float2* red = (float2*)(bias_lds + 256);  // [256 rows][4 column waves]
        const float sc2 = stats.scale * 1.4426950408889634f;
        const int cidx = (kg & 1) * 2 + (kg >> 1);
        const int sw = tm_swz(l15);
        f16* lane0 = (f16*)out + ((size_t)(m0 >> 8) * (N >> 5) + (n0 >> 5) + wc * 2) * TM_BLOCK +
                     (wr * 128 + l15) * 32 + ((cidx ^ sw) << 3);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          uint32_t h[4][2];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const uint2 hp = __builtin_bit_cast(uint2, epi_act_pack<EPI>(acc.v[ni][mi]));
            h[ni][0] = hp.x;
            h[ni][1] = hp.y;
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {  // the store of store_tile (MODE 0) below
            const auto s0 = __builtin_amdgcn_permlane16_swap(h[2 * j][0], h[2 * j + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(h[2 * j][1], h[2 * j + 1][1], false, false);
            const u32x4 chunk = {s0[0], s1[0], s0[1], s1[1]};
#ifndef SMI_PROBE_NOSTORE  // probe build (wrong results on purpose): what ANY scheme that never stores the logits could save
            store_nt((u32x4*)(lane0 + (size_t)j * TM_BLOCK + mi * (16 * 32)), chunk);
#else
            if (chunk[0] == 0x12345678u && stats.valid_n < 0) store_nt((u32x4*)(lane0 + (size_t)j * TM_BLOCK + mi * (16 * 32)), chunk);
#endif
          }
          f32x2 t[8];  // the lane's 16 ROUNDED values of the row
          float mx = -INFINITY;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const half2v hv = __builtin_bit_cast(half2v, h[ni][q]);
              t[ni * 2 + q] = f32x2{(float)hv[0], (float)hv[1]};
              mx = fmaxf(mx, fmaxf(t[ni * 2 + q][0], t[ni * 2 + q][1]));
            }
          {
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
            const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
          }
          const float mxs = mx * sc2;                        // the row maximum in the log2 domain, as the general code has it
          const float nb = mxs == -INFINITY ? 0.f : -mxs;
          const f32x2 sc22 = {sc2, sc2}, nb2 = {nb, nb};
          f32x2 se2 = {0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const f32x2 a = __builtin_elementwise_fma(t[e], sc22, nb2);
            se2 += f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
          }
          float se = se2[0] + se2[1];
          {
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(se), __float_as_uint(se), false, false);
            se = __uint_as_float(a[0]) + __uint_as_float(a[1]);
            const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(se), __float_as_uint(se), false, false);
            se = __uint_as_float(b[0]) + __uint_as_float(b[1]);
          }
          if (kg == 0) lds_write_b64_asm(&red[(wr * 128 + mi * 16 + l15) * 4 + wc], float2{mxs, se});
        }
        SMI_LGKM0_BARRIER();
        if (tid < 256) {
          f32x4 r01, r23;
          {
            const char* rp[2] = {(const char*)&red[tid * 4], (const char*)&red[tid * 4 + 2]};
            f32x4 rv[2];
            lds_read_stage<2>(rp, rv);
            r01 = rv[0];
            r23 = rv[1];
          }
          const float2 a0 = {r01[0], r01[1]}, a1 = {r01[2], r01[3]}, a2 = {r23[0], r23[1]}, a3 = {r23[2], r23[3]};
          const float m = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
          const float ms = m == -INFINITY ? 0.f : m;
          const float sum = a0.y * __builtin_amdgcn_exp2f(a0.x - ms) + a1.y * __builtin_amdgcn_exp2f(a1.x - ms) +
                            a2.y * __builtin_amdgcn_exp2f(a2.x - ms) + a3.y * __builtin_amdgcn_exp2f(a3.x - ms);
          const size_t o = (size_t)tile_n_cur * M + m0 + tid;
          stats.tile_max[o] = m * 0.6931471805599453f;
          stats.tile_sum[o] = sum;
        }
      }
    }
    if constexpr (EPI == EPI_STORE_F32 || STATS_F16) {
      if (stats.tile_max && !stats_stored) {
        // softmax statistics of this tile's 256 columns for each of its 256 rows (the decoder's logits
        // GEMM, no bias): branch-free and lane-local over the lane's 16 values of a row in the log2
        // domain (t = v * scale * log2 e, one v_exp_f32 per element), joined across the 4 lane groups
        // by two shuffles and across the 4 column waves through LDS (above the bias slice).
        float2* red = (float2*)(bias_lds + 256);  // [256 rows][4 column waves]
        const float sc2 = stats.scale * 1.4426950408889634f;
        const bool full = n0 + G2_BN <= stats.valid_n;  // every tile but the last one
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          float t[4][4];
          float mx = -INFINITY;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc.v[ni][mi][r];
              if constexpr (STATS_F16) v = (float)(f16)v;
              v *= sc2;
              if (!full) v = (n0 + wc * 64 + ni * 16 + 4 * kg + r) < stats.valid_n ? v : -INFINITY;
              t[ni][r] = v;
              mx = fmaxf(mx, v);
            }
          // lanes 16 / 32 apart joined by v_permlane16/32_swap (VALU rate; __shfl_xor is an LDS round trip per step)
          {
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
            const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
          }
          const float ms = mx == -INFINITY ? 0.f : mx;
          float se = 0.f;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) se += __builtin_amdgcn_exp2f(t[ni][r] - ms);
          {
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(se), __float_as_uint(se), false, false);
            se = __uint_as_float(a[0]) + __uint_as_float(a[1]);
            const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(se), __float_as_uint(se), false, false);
            se = __uint_as_float(b[0]) + __uint_as_float(b[1]);
          }
          if (kg == 0) lds_write_b64_asm(&red[(wr * 128 + mi * 16 + l15) * 4 + wc], float2{mx, se});
        }
        SMI_LGKM0_BARRIER();
        if (tid < 256) {
          f32x4 r01, r23;  // (max, sum) of column waves 0, 1 and 2, 3: two 16-B reads through the asm helper
          {
            const char* rp[2] = {(const char*)&red[tid * 4], (const char*)&red[tid * 4 + 2]};
            f32x4 rv[2];
            lds_read_stage<2>(rp, rv);
            r01 = rv[0];
            r23 = rv[1];
          }
          const float2 a0 = {r01[0], r01[1]}, a1 = {r01[2], r01[3]}, a2 = {r23[0], r23[1]}, a3 = {r23[2], r23[3]};
          const float m = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
          const float ms = m == -INFINITY ? 0.f : m;
          const float sum = a0.y * __builtin_amdgcn_exp2f(a0.x - ms) + a1.y * __builtin_amdgcn_exp2f(a1.x - ms) +
                            a2.y * __builtin_amdgcn_exp2f(a2.x - ms) + a3.y * __builtin_amdgcn_exp2f(a3.x - ms);
          const size_t o = (size_t)tile_n_cur * M + m0 + tid;  // [tile][row]: 1 KiB coalesced per tile
          stats.tile_max[o] = m * 0.6931471805599453f;         // back to natural units
          stats.tile_sum[o] = sum;
        }
      }
    }
    if constexpr (LAYOUT == 3) {
      // fp16 residual stream in the tile-major layout: read-modify-write straight from the accumulators (no LDS
      // staging, no barriers).  As in the LAYOUT == 2 store, a 32-column k-block is the accumulator-block pair
      // A = 2j, B = 2j + 1 and v_permlane16_swap joins lane groups so that every lane owns one whole 16-B chunk --
      // here on the fp32 values (4 swaps per pair instead of 2), because the residual add is one fp32 add rounded
      // once.  The 16 old chunks of a wave are requested before the first add.
      static_assert(EPI == EPI_RESID_F16 || EPI == EPI_RESID_HALF_F16, "tile-major residual: fp16 stream epilogues only");
      constexpr float RSTEP = EPI == EPI_RESID_HALF_F16 ? 0.5f : 1.0f;  // x += 0.5 * (...): the conformer's macaron FFNs
      const int cidx = (kg & 1) * 2 + (kg >> 1);
      const int sw = tm_swz(l15);
      f16* lane0 = (f16*)out + ((size_t)(m0 >> 8) * (N >> 5) + (n0 >> 5) + wc * 2) * TM_BLOCK +
                   (wr * 128 + l15) * 32 + ((cidx ^ sw) << 3);
      half8 oldv[8][2];
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) oldv[mi][j] = *(const half8*)(lane0 + (size_t)j * TM_BLOCK + mi * (16 * 32));
      const bool emit = FOLD_PRODUCER && fold.part_out != nullptr;
      float rs_sum[8], rs_sq[8];  // fold producer: sums of the lane's 16 NEW values of row (mi, l15)
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        rs_sum[mi] = 0.f;
        rs_sq[mi] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 a = acc.v[2 * j][mi], b = acc.v[2 * j + 1][mi];
          float c8[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const auto sp = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(b[i]), false, false);
            c8[i] = __uint_as_float(sp[0]);
            c8[4 + i] = __uint_as_float(sp[1]);
          }
          half8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (f16)__builtin_fmaf(c8[i], RSTEP, (float)oldv[mi][j][i]);
          store_nt((half8*)(lane0 + (size_t)j * TM_BLOCK + mi * (16 * 32)), o);
          if (emit) {
            // of the ROUNDED values (what the consuming GEMM will read), two per v_dot2_f32_f16: exact fp16 products,
            // fp32 accumulation -- 8 instructions per chunk instead of 8 conversions + 8 adds + 8 FMAs
            const half2v ones = {(f16)1.f, (f16)1.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const half2v t2 = {o[2 * i], o[2 * i + 1]};
              rs_sum[mi] = __builtin_amdgcn_fdot2(t2, ones, rs_sum[mi], false);
              rs_sq[mi] = __builtin_amdgcn_fdot2(t2, t2, rs_sq[mi], false);
            }
          }
        }
      }
      if (emit) {
        // join the 4 lane groups of a row (lanes 16 apart), then the 4 column waves through LDS, then one coalesced
        // 2 KiB store of the tile's 256 partial (sum, sum of squares) pairs
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          float v0 = rs_sum[mi], v1 = rs_sq[mi];
          auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v0), false, false);
          v0 = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
          auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v0), false, false);
          v0 = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
          s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v1), __float_as_uint(v1), false, false);
          v1 = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
          s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v1), __float_as_uint(v1), false, false);
          v1 = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
          if (kg == 0) lds_write_b64_asm(&rowsum_lds[wc * 256 + wr * 128 + mi * 16 + l15], float2{v0, v1});
        }
        SMI_LGKM0_BARRIER();
        if (tid < 256) {
          float2 a0, a1, a2, a3;  // issue and wait in ONE statement (see lds_read_stage)
          asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:2048\n\tds_read_b64 %2, %4 offset:4096\n\t"
                       "ds_read_b64 %3, %4 offset:6144\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                       : "v"((unsigned)(size_t)(rowsum_lds + tid))
                       : "memory");
          fold.part_out[(size_t)tile_n_cur * M + m0 + tid] = float2{(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y)};
        }
      }
    } else
    if constexpr (F32_OUT) {
      // fp32 outputs: 8 sub-passes (row pass p, column half nh) of 64 rows x 128 columns (the waves'
      // blocks ni = 2nh, 2nh+1).  The residual values of sub-pass sp+1 are requested before the barrier
      // of sub-pass sp (the barrier's memory clobber would otherwise pin every load behind it and
      // expose one HBM latency per sub-pass).
      const int c = lane & 31;
      auto load_old = [&](int sp, f32x4 (&o)[4]) {
        const int p = sp >> 1, nh = sp & 1;
        const int gcol = n0 + (c >> 3) * 64 + nh * 32 + (c & 7) * 4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int lr = wave * 8 + it * 2 + hi;
          const int row = m0 + (lr >> 5) * 128 + p * 32 + (lr & 31);
          if constexpr (EPI == EPI_STORE_F32) {
            o[it] = f32x4{0.f, 0.f, 0.f, 0.f};
          } else if constexpr (F16_RESID) {
            const half4 hv = *(const half4*)((const f16*)out + (size_t)row * ldo + gcol);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[it][e] = (float)hv[e];
          } else {
            o[it] = *(const f32x4*)((const float*)out + (size_t)row * ldo + gcol);
          }
        }
      };
      f32x4 old[2][4];
      load_old(0, old[0]);
#pragma unroll
      for (int sp = 0; sp < 8; ++sp) {
        const int p = sp >> 1, nh = sp & 1;
        char* st = g2_stage(smem, sp);
        const int gcol = n0 + (c >> 3) * 64 + nh * 32 + (c & 7) * 4;
        if (sp + 1 < 8) load_old(sp + 1, old[(sp + 1) & 1]);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int nl = 0; nl < 2; ++nl) {
            const int mi = 2 * p + mh, ni = 2 * nh + nl;
            f32x4 v = acc.v[ni][mi];
            if constexpr (HALF_STEP) v = v * 0.5f;
            const int lr = lr_w(mi);
            // 16-B chunk (4 floats) of the 128 staged columns: wc*8 + nl*4 + kg
            lds_write_f4_asm(st + lr * 512 + (((wc * 8 + nl * 4 + kg) ^ g2_stage_swz(lr)) << 4), v);
          }
        SMI_LGKM0_BARRIER();
        f32x4 sv[4];
        {
          const char* sp_[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int lr = wave * 8 + it * 2 + hi;
            sp_[it] = st + lr * 512 + ((c ^ g2_stage_swz(lr)) << 4);
          }
          lds_read_stage<4>(sp_, sv);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int lr = wave * 8 + it * 2 + hi;
          const int row = m0 + (lr >> 5) * 128 + p * 32 + (lr & 31);
          const f32x4 v = sv[it];
          if constexpr (F16_RESID) {
            const f32x4 sum = old[sp & 1][it] + v;
            half4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (f16)sum[e];
            *(half4*)((f16*)out + (size_t)row * ldo + gcol) = hv;
          } else {
            *(f32x4*)((float*)out + (size_t)row * ldo + gcol) = old[sp & 1][it] + v;
          }
        }
      }
    } else if constexpr (EPI == EPI_GLU_F16) {
      // 128 output channels per tile; a wave's 64 columns are [32 values | 32 gates] (W rows interleaved
      // in 32-channel groups at pack time): channel wc*32 + nl*16 + 4kg + r from blocks ni = nl, nl + 2
      if constexpr (FOLD_CONSUMER_ST) {
        if (folded) fold_affine_st(acc);  // acc <- rstd(row) * acc + c2(column), the LayerNorm in front of this GEMM
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        char* st = g2_stage(smem, p);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int nl = 0; nl < 2; ++nl) {
            const int mi = 2 * p + mh;
            half4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              h[e] = (f16)(acc.v[nl][mi][e] * sigmoid_f(acc.v[nl + 2][mi][e]));
            const int lr = lr_w(mi);
            // 16-B chunk (8 channels) of the 128 staged channels: wc*4 + nl*2 + (kg>>1), half kg&1
            lds_write_b64_asm(st + lr * 512 + (((wc * 4 + nl * 2 + (kg >> 1)) ^ g2_stage_swz(lr)) << 4) + (kg & 1) * 8, h);
          }
        SMI_LGKM0_BARRIER();
        const int c = lane & 15;  // 16 lanes x 16 B = one 256-B output row
        f32x4 sv[2];
        {
          const char* sp_[2];
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int lr = wave * 8 + it * 4 + (lane >> 4);
            sp_[it] = st + lr * 512 + ((c ^ g2_stage_swz(lr)) << 4);
          }
          lds_read_stage<2>(sp_, sv);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int lr = wave * 8 + it * 4 + (lane >> 4);
          const int row = m0 + (lr >> 5) * 128 + p * 32 + (lr & 31);
          const f32x4 v = sv[it];
          *(f32x4*)((f16*)out + (size_t)row * ldo + n0 / 2 + c * 8) = v;
        }
      }
    } else if constexpr (LAYOUT == 2) {
      // fp16 tile-major output straight from the accumulators, no LDS pass, no barriers.  A lane holds
      // 4 consecutive columns (8 B) of row l15 per 16-column block; a 32-column k-block is the block
      // pair A = 2j, B = 2j+1.  v_permlane16_swap moves lane group 1's piece of A against group 0's
      // piece of B (and 3 against 2), after which every lane holds a whole 16-B chunk: group kg owns
      // chunk (kg&1)*2 + (kg>>1) of the k-block.  A wave instruction then stores 16 rows x 64 B = one
      // dense, contiguous 1 KiB run of a tile-major block.
      const int cidx = (kg & 1) * 2 + (kg >> 1);
      const int sw = tm_swz(l15);
      f16* lane0 = (f16*)out + ((size_t)(m0 >> 8) * (N >> 5) + (n0 >> 5) + wc * 2) * TM_BLOCK +
                   (wr * 128 + l15) * 32 + ((cidx ^ sw) << 3);
      // LayerNorm fold: acc holds x . (W (.) g)^T; out = rstd * (acc - mean * c1) + c2 per (row, column) -- or, with
      // row-centred weights, rstd * acc + c2.  The store loop exists in three straight-line copies (MODE 0 plain, 1 exact
      // fold, 2 centred fold) chosen by ONE wave-uniform branch per tile: with the mode tested inside the unrolled loop
      // every iteration carried two or three scalar branches, the compiler could not overlap iterations across them, and
      // the consuming GEMMs ran 6-9 % slower than the unfolded kernel (r03 experiments).  The loop runs k-block (j)
      // outermost so that only the two 16-column blocks of a k-block have their c1 / c2 in registers.
      auto store_tile = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        float rsall[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        float row_rs[2] = {1.f, 1.f}, row_nm[2] = {0.f, 0.f};
        if constexpr (MODE != 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const float mean = row_sq[u].x * fold.inv_k;
            const float var = fmaxf(row_sq[u].y * fold.inv_k - mean * mean, 0.f);
            row_rs[u] = __builtin_amdgcn_rsqf(var + fold.eps);  // 1 ulp; the result is rounded to fp16 a few steps later
            row_nm[u] = -row_rs[u] * mean;
          }
        }
        if constexpr (MODE == 2) {  // row mi * 16 + l15 of the wave's 128: lane (mi * 16 + l15) & 63, register mi >> 2
#pragma unroll
          for (int mi = 0; mi < 8; ++mi)
            rsall[mi] = __int_as_float(__builtin_amdgcn_ds_bpermute(((mi & 3) * 16 + l15) * 4, __float_as_int(row_rs[mi >> 2])));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 c1v[2] = {}, c2v[2] = {};
          if constexpr (MODE != 0) {  // columns (2j + nl) * 16 + 4 kg + r of the wave's 64: held by that lane
#pragma unroll
            for (int nl = 0; nl < 2; ++nl)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int src = ((2 * j + nl) * 16 + 4 * kg + r) * 4;
                if constexpr (MODE == 1) c1v[nl][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(col_c1)));
                c2v[nl][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(col_c2)));
              }
          }
#pragma unroll
          for (int mi = 0; mi < 8; ++mi) {
            float2 aff = {1.f, 0.f};
            if constexpr (MODE == 2) aff.x = rsall[mi];
            if constexpr (MODE == 1) {
              const int src = ((mi & 3) * 16 + l15) * 4;
              aff.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(row_rs[mi >> 2])));
              aff.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(row_nm[mi >> 2])));
            }
            uint32_t h[2][2];
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) {
              f32x4 v = acc.v[2 * j + nl][mi];
              if constexpr (MODE != 0) {
                // v_pk_fma_f32: two values per instruction (a wave64 VALU instruction takes 4 cycles)
                const f32x2 rs2 = {aff.x, aff.x}, nm2 = {aff.y, aff.y};
#pragma unroll
                for (int hp2 = 0; hp2 < 2; ++hp2) {
                  const f32x2 c2p = {c2v[nl][2 * hp2], c2v[nl][2 * hp2 + 1]};
                  f32x2 vp = {v[2 * hp2], v[2 * hp2 + 1]};
                  if constexpr (MODE == 1) {
                    const f32x2 c1p = {c1v[nl][2 * hp2], c1v[nl][2 * hp2 + 1]};
                    vp = __builtin_elementwise_fma(rs2, vp, __builtin_elementwise_fma(nm2, c1p, c2p));
                  } else {
                    vp = __builtin_elementwise_fma(rs2, vp, c2p);
                  }
                  v[2 * hp2] = vp[0];
                  v[2 * hp2 + 1] = vp[1];
                }
              }
              const uint2 hp = __builtin_bit_cast(uint2, epi_act_pack<EPI>(v));
              h[nl][0] = hp.x;
              h[nl][1] = hp.y;
            }
            // rows 16..31 / 48..63 of h[0] <-> rows 0..15 / 32..47 of h[1]
            const auto s0 = __builtin_amdgcn_permlane16_swap(h[0][0], h[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(h[0][1], h[1][1], false, false);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 chunk = {s0[0], s1[0], s0[1], s1[1]};
            store_nt((u32x4*)(lane0 + (size_t)j * TM_BLOCK + mi * (16 * 32)), chunk);
          }
        }
      };
      if constexpr (FOLD_CONSUMER) {
        if (stats_stored) {
          // written by the fused statistics + store pass
        } else if (!folded)
          store_tile(std::integral_constant<int, 0>{});
        else if (fold.centered || !FOLD_EXACT)
          store_tile(std::integral_constant<int, 2>{});
        else if constexpr (FOLD_EXACT)
          store_tile(std::integral_constant<int, 1>{});
      } else {
        store_tile(std::integral_constant<int, 0>{});
      }
    } else {
      if constexpr (FOLD_CONSUMER_ST) {
        if (folded) fold_affine_st(acc);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        char* st = g2_stage(smem, p);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const int mi = 2 * p + mh;
            f32x4 v = acc.v[ni][mi];
            const half4 h = epi_act_pack<EPI>(v);
            const int lr = lr_w(mi);
            // 16-B chunk (8 columns) of the 256 staged columns: wc*8 + ni*2 + (kg>>1), half kg&1
            lds_write_b64_asm(st + lr * 512 + (((wc * 8 + ni * 2 + (kg >> 1)) ^ g2_stage_swz(lr)) << 4) + (kg & 1) * 8, h);
          }
        SMI_LGKM0_BARRIER();
        const int c = lane & 31;
        f32x4 sv[4];
        {
          const char* sp_[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int lr = wave * 8 + it * 2 + hi;
            sp_[it] = st + lr * 512 + ((c ^ g2_stage_swz(lr)) << 4);
          }
          lds_read_stage<4>(sp_, sv);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int lr = wave * 8 + it * 2 + hi;
          const int row = m0 + (lr >> 5) * 128 + p * 32 + (lr & 31);
          const f32x4 v = sv[it];
          *(f32x4*)((f16*)out + (size_t)row * ldo + n0 + c * 8) = v;
        }
      }
    }
    // staging reads retired by every wave before the next tile's loop refills slot 3
    SMI_LGKM0_BARRIER();
    G2_TRACE(4);
#ifdef SMI_GEMM_TRACE
    ++trace_i;
#endif
  }
}

#ifdef SMI_GEMM_TRACE
extern "C" int smi_debug_gemm_trace(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g2_trace_buf), sizeof(g2_trace_buf));
}
#endif

// Persistent-grid cap of the calling thread's next 256x256 launches (0 = none): a decode chain that shares the chip with
// other chains leaves CUs to them during its one throughput-bound launch (the logits GEMM)
static thread_local int g2_grid_cap = 0;
void set_gemm_grid_cap(int workgroups) { g2_grid_cap = workgroups; }

int num_cus() {
  static std::atomic<int> cached[64];
  const int dev = DeviceOnce::dev();
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

template <int EPI, int LAYOUT>
static hipError_t launch_one256(const f16* X, const f16* W, const float* bias, void* out, int M,
                                int N, int K, int ldo, hipStream_t stream, const GemmTileStats* stats = nullptr,
                                int ksplit = 1, size_t part_stride = 0, const GemmLnFold* fold = nullptr) {
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256_kernel<EPI, LAYOUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G2_KERNEL_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done.set();
  }
  int grid = std::min((M / G2_BM) * (N / G2_BN) * ksplit, num_cus());
  if (g2_grid_cap > 0) grid = std::min(grid, g2_grid_cap);
  const int want_raster = tune(TUNE_G2_RASTER, 2);  // G2_RASTER=0 restores the id-order raster everywhere (A/B measurements)
  const int ntm = M / G2_BM, ntn = N / G2_BN;
  // XCD-owned m-groups (see the kernel): whole chip, >= 4 n-quads, and a number of m-groups (8 row tiles each) that
  // deals evenly to the 8 XCDs -- otherwise the id-order raster balances better
  const int raster = (want_raster && ksplit == 1 && grid == 256 && ntn % 4 == 0 && ntn >= 16 && ((ntm + 7) / 8) % 8 == 0)
                         ? want_raster : 0;
  hipLaunchKernelGGL((gemm_tn256_kernel<EPI, LAYOUT>), dim3(grid), dim3(G2_THREADS), G2_KERNEL_LDS_BYTES,
                     stream, X, W, bias, out, M, N, K, ldo, stats ? *stats : GemmTileStats{nullptr, nullptr, 1.f, 0}, ksplit,
                     part_stride, raster, fold ? *fold : GemmLnFold{nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0});
  return hipGetLastError();
}

// SMI_GT_RING: stages of round 3's lone-tile ring (0 = never use it, 4; default 4) -- A/B switch, used with SMI_LONE=0
static int gt_ring_stages() {
  return tune(TUNE_GT_RING, 4) == 4 ? 4 : 0;
}

// SMI_LONE: 0 = round 3's ring for every lone-tile launch (A/B runs; read per launch: decode-time paths switch it per
// call), otherwise 64x64 units of the lone-tile engine (gemm_lone.hpp) when a launch is small enough for them.
static bool lone_enabled() {
  return tune(TUNE_LONE, 1) != 0;
}
// 64x64 units, two workgroups per CU (64 KiB of LDS each): used while all units are resident at once.  Measured
// (profiles/r04_experiments.txt, experiment 11): at M = 256 / 512 every projection of the encoder is 25-35 % faster than on
// 128x128 tiles (more CUs stream operands, a unit has a quarter of the MFMAs and half the LDS traffic); past ~2 units
// per CU (M = 1280 x N = 3072: 960 units) the 128x128 ring wins again -- a 64x64 unit moves twice the operand bytes per
// flop through L2.
static bool lone_fits(int M, int N, int ksplit) {
  return (int64_t)(M / 64) * (N / 64) * ksplit <= 2 * (int64_t)num_cus();
}
static bool ring_fits(int M, int N, int ksplit) {
  return (int64_t)(M / GT_BM) * (N / GT_BN) * ksplit <= num_cus();
}

template <int EPI, int LAYOUT, int RING>
static hipError_t launch_one_ring(const f16* X, const f16* W, const float* bias, void* out, int M, int N,
                                  int K, int ldo, hipStream_t stream, int ksplit, size_t part_stride) {
  constexpr int lds = (RING ? RING : 2) * GT_STAGE_BYTES;
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<EPI, LAYOUT, RING>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done.set();
  }
  const int grid = (M / GT_BM) * (N / GT_BN);
  hipLaunchKernelGGL((gemm_tn_kernel<EPI, LAYOUT, RING>), dim3(grid, ksplit), dim3(GT_THREADS), lds,
                     stream, X, W, bias, out, M, N, K, ldo, ksplit, part_stride);
  return hipGetLastError();
}

template <int EPI, int LAYOUT, int BM, int BN>
static hipError_t launch_lone(const f16* X, const f16* W, const float* bias, void* out, int M, int N, int K, int ldo,
                              hipStream_t stream, int ksplit, size_t part_stride) {
  constexpr int lds = LoneShape<BM, BN>::LDS_BYTES;
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_lone_kernel<EPI, LAYOUT, BM, BN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done.set();
  }
  const int grid = (M / BM) * (N / BN);
  hipLaunchKernelGGL((gemm_lone_kernel<EPI, LAYOUT, BM, BN>), dim3(grid, ksplit), dim3(GT_THREADS), lds, stream, X, W,
                     bias, out, M, N, K, ldo, ksplit, part_stride);
  return hipGetLastError();
}

// the k-sliced unit (gemm_lone16.hpp): tile-major operands, K per unit 256 / 512 / 1024.  SMI_LONE16=0: the LDS-ring unit
// for every lone launch (A/B runs; its results are bit-identical to the other 128x128-family engines, these are not)
static bool lone16_enabled() {
  return tune(TUNE_LONE16, 1) != 0;
}
template <int EPI, bool OUT_TM>
static hipError_t launch_lone16(const f16* X, const f16* W, const float* bias, void* out, int M, int N, int K, int ldo,
                                hipStream_t stream, int ksplit, size_t part_stride) {
  const int grid = (M / L16_BM) * (N / L16_BN);
  const int nkb = K / ksplit / 128;
#define SMI_L16(NKB)                                                                                                      \
  {                                                                                                                       \
    static DeviceOnce attr_done;                                                                                          \
    if (!attr_done.done()) {                                                                                              \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_lone16_kernel<EPI, OUT_TM, NKB>,                               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, L16_LDS_BYTES);                      \
      if (e != hipSuccess) return e;                                                                                      \
      attr_done.set();                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL((gemm_lone16_kernel<EPI, OUT_TM, NKB>), dim3(grid, ksplit), dim3(L16_THREADS), L16_LDS_BYTES,     \
                       stream, X, W, bias, out, M, N, K, ldo, part_stride);                                               \
  }
  if (nkb == 8) SMI_L16(8) else if (nkb == 4) SMI_L16(4) else SMI_L16(2)
#undef SMI_L16
  return hipGetLastError();
}

template <int EPI, int LAYOUT = 0>
static hipError_t launch_one(const f16* X, const f16* W, const float* bias, void* out, int M, int N,
                             int K, int ldo, hipStream_t stream, int ksplit = 1, size_t part_stride = 0) {
  if constexpr ((EPI == EPI_BIAS_F16 || EPI == EPI_RELU_F16 || EPI == EPI_STORE_F32) && (LAYOUT == 1 || LAYOUT == 2)) {
    const int klen = K / ksplit;
    if (lone_enabled() && lone16_enabled() && lone_fits(M, N, ksplit) && K % ksplit == 0 &&
        (klen == 256 || klen == 512 || klen == 1024) && (EPI != EPI_STORE_F32 || LAYOUT == 1))
      return launch_lone16<EPI, LAYOUT == 2>(X, W, bias, out, M, N, K, ldo, stream, ksplit, part_stride);
  }
  if constexpr (EPI != EPI_GLU_F16)  // GLU pairs two 32-column blocks of a wave: 128-column tiles only
    if (lone_enabled() && lone_fits(M, N, ksplit))
      return launch_lone<EPI, LAYOUT, 64, 64>(X, W, bias, out, M, N, K, ldo, stream, ksplit, part_stride);
  // every workgroup gets a CU of its own: hide the DMA latency with a deeper ring instead of a second workgroup
  if (ring_fits(M, N, ksplit) && gt_ring_stages() == 4)
    return launch_one_ring<EPI, LAYOUT, 4>(X, W, bias, out, M, N, K, ldo, stream, ksplit, part_stride);
  return launch_one_ring<EPI, LAYOUT, 0>(X, W, bias, out, M, N, K, ldo, stream, ksplit, part_stride);
}

hipError_t launch_gemm_tn(int epi_sel, const f16* X, const f16* W, const float* bias, void* out,
                          int M, int N, int K, int ldo, hipStream_t stream, const GemmTileStats* stats,
                          const GemmLnFold* fold) {
  // epi_sel = epilogue | (engine << 8) | layout flags: engine 0 auto, 1 force 128x128, 2 force
  // 256x256; GEMM_IN_TM = X and W tile-major, GEMM_OUT_TM = fp16 output tile-major (needs IN_TM)
  const int epi = epi_sel & 0xff, sel = (epi_sel >> 8) & 0xf;
  const bool in_tm = epi_sel & GEMM_IN_TM, out_tm = epi_sel & GEMM_OUT_TM;
  if (M % GT_BM || N % GT_BN || K % GT_BK || M <= 0) return hipErrorInvalidValue;
  if (in_tm && (M % TM_ROWS || N % TM_ROWS)) return hipErrorInvalidValue;
  if (out_tm && (!in_tm || ldo != (epi == EPI_GLU_F16 ? N / 2 : N))) return hipErrorInvalidValue;
  const bool can256 = M % G2_BM == 0 && N % G2_BN == 0;
  if (sel == 2 && !can256) return hipErrorInvalidValue;
  // the 256x256 ping-pong engine is ~1.5x more efficient per CU than the 128x128 one but has a 21 us
  // floor for a K = 1024 tile and one workgroup per CU; measured crossover (tools/probe_engines.py):
  // 128 tiles tie, 160 tiles win; round 4 (tools/probe_engines_mid.py, M = 1024 x N = 8192 = 128 tiles, tile-major operands): 25.8 vs
  // 28.6 us hot, 31.8 vs 33.8 us on cold weights -> use it from 128 tiles (half the CUs) up
  // (SMI_G2_AUTO_MIN overrides the threshold, read per launch: tests that compare runs of different row counts bit for bit
  // pin the engine family with it)
  const int64_t auto_min = tune(TUNE_G2_AUTO_MIN, 128);
  const bool use256 = sel == 2 || (sel == 0 && can256 && (int64_t)(M / G2_BM) * (N / G2_BN) >= auto_min);
  if (fold) {  // LayerNorm fold: 256x256 engine, tile-major stream
    if (!can256 || sel == 1 || !in_tm || stats) return hipErrorInvalidValue;
    if (fold->part_in) {  // consumer: tile-major outputs (bias / relu / silu) or row-major ones (bias / GLU; centred weights)
      if (!fold->c1 || fold->nparts < 1 || fold->nparts > 4) return hipErrorInvalidValue;
      if (out_tm) {
        // the 4-wave engine (gemm_v2.hip) where it applies: >= 24 K slices, several persistent rounds
        if (gemm_v2_fits(epi, M, N, K, bias, fold)) return launch_gemm_v2(epi, X, W, bias, (f16*)out, M, N, K, stream, fold);
        if (epi == EPI_BIAS_F16) return launch_one256<EPI_BIAS_F16, 2>(X, W, bias, out, M, N, K, ldo, stream, nullptr, 1, 0, fold);
        if (epi == EPI_RELU_F16) return launch_one256<EPI_RELU_F16, 2>(X, W, bias, out, M, N, K, ldo, stream, nullptr, 1, 0, fold);
        if (epi == EPI_SILU_F16 && fold->centered)
          return launch_one256<EPI_SILU_F16, 2>(X, W, bias, out, M, N, K, ldo, stream, nullptr, 1, 0, fold);
        return hipErrorInvalidValue;
      }
      if (!fold->centered) return hipErrorInvalidValue;
      if (epi == EPI_BIAS_F16) return launch_one256<EPI_BIAS_F16, 1>(X, W, bias, out, M, N, K, ldo, stream, nullptr, 1, 0, fold);
      if (epi == EPI_GLU_F16) return launch_one256<EPI_GLU_F16, 1>(X, W, bias, out, M, N, K, ldo, stream, nullptr, 1, 0, fold);
      return hipErrorInvalidValue;
    }
    // producer: the tile-major residual epilogue (part_out may be null: plain read-modify-write of the stream)
    if (!out_tm) return hipErrorInvalidValue;
    if ((epi == EPI_RESID_F16 || epi == EPI_RESID_HALF_F16) && gemm_v2_fits(epi, M, N, K, bias, fold))
      return launch_gemm_v2(epi, X, W, bias, (f16*)out, M, N, K, stream, fold);
    if (epi == EPI_RESID_F16) return launch_one256<EPI_RESID_F16, 3>(X, W, bias, out, M, N, K, ldo, stream, nullptr, 1, 0, fold);
    if (epi == EPI_RESID_HALF_F16)
      return launch_one256<EPI_RESID_HALF_F16, 3>(X, W, bias, out, M, N, K, ldo, stream, nullptr, 1, 0, fold);
    return hipErrorInvalidValue;
  }
  if (stats) {  // tile statistics: the 256x256 engine's fp32-store epilogue or its tile-major fp16 store, without a bias
    if (epi == EPI_BIAS_F16 && out_tm && in_tm && can256 && sel != 1 && !bias) {
      if (gemm_v2_stats_fits(M, N, K, stats)) return launch_gemm_v2_stats(X, W, (f16*)out, M, N, K, stream, stats, g2_grid_cap);
      return launch_one256<EPI_BIAS_F16, 2>(X, W, bias, out, M, N, K, ldo, stream, stats);
    }
    if (epi != EPI_STORE_F32 || out_tm || !can256 || sel == 1 || bias) return hipErrorInvalidValue;
    return in_tm ? launch_one256<EPI_STORE_F32, 1>(X, W, bias, out, M, N, K, ldo, stream, stats)
                 : launch_one256<EPI_STORE_F32, 0>(X, W, bias, out, M, N, K, ldo, stream, stats);
  }
#define SMI_EPI_CASE(E, L)                                                     \
  case E:                                                                      \
    return use256 ? launch_one256<E, L>(X, W, bias, out, M, N, K, ldo, stream) \
                  : launch_one<E, L>(X, W, bias, out, M, N, K, ldo, stream);
  if (out_tm) {  // fp16 outputs that feed the next GEMM; EPI_RESID_F16: the tile-major residual stream
    // a decode step's FFN-inner projection (M = 1280 rows): 256 lone units of 160 x 256 instead of 160 of 256 x 256
    // (and the small-batch encoder's fused QKV projection: bias, tile-major out)
    if ((epi == EPI_RELU_F16 || epi == EPI_BIAS_F16) && sel != 1 && gemm_v2_lone_fits(M, N, K, 1))
      return launch_gemm_v2_lone(epi == EPI_RELU_F16 ? 1 : 2, X, W, bias, out, M, N, K, 1, stream);
    if (use256 && gemm_v2_fits(epi, M, N, K, bias, nullptr))
      return launch_gemm_v2(epi, X, W, bias, (f16*)out, M, N, K, stream, nullptr);
    switch (epi) {
      SMI_EPI_CASE(EPI_BIAS_F16, 2)
      SMI_EPI_CASE(EPI_RELU_F16, 2)
      SMI_EPI_CASE(EPI_SILU_F16, 2)
      SMI_EPI_CASE(EPI_RESID_F16, 3)
      SMI_EPI_CASE(EPI_RESID_HALF_F16, 3)
    }
    return hipErrorInvalidValue;
  }
  if (in_tm) {
    switch (epi) {
      SMI_EPI_CASE(EPI_BIAS_F16, 1)
      SMI_EPI_CASE(EPI_RESID_F32, 1)
      SMI_EPI_CASE(EPI_STORE_F32, 1)
      SMI_EPI_CASE(EPI_RESID_HALF_F32, 1)
      SMI_EPI_CASE(EPI_RESID_F16, 1)
      SMI_EPI_CASE(EPI_RESID_HALF_F16, 1)
      SMI_EPI_CASE(EPI_GLU_F16, 1)
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
    SMI_EPI_CASE(EPI_TANH_F16, 0)
    SMI_EPI_CASE(EPI_RESID_F16, 0)
    SMI_EPI_CASE(EPI_RESID_HALF_F16, 0)
  }
#undef SMI_EPI_CASE
  return hipErrorInvalidValue;
}

// Split-K GEMM into `ksplit` slabs: parts[z][m][n] = X[:, Kz] . W[:, Kz]^T (+ bias for z = 0); the consumer
// (launch_sum_layernorm / launch_fold_residual) adds the slabs to the residual stream.  slab_f16: the slabs are fp16 (an fp16
// model rounds every sublayer output to fp16 in the reference; here each PARTIAL is rounded once and the sum is formed in
// fp32: half the slab traffic between the two kernels), else fp32.
hipError_t launch_gemm_tn_splitk(const f16* X, const f16* W, const float* bias, void* parts, int M,
                                 int N, int K, int ksplit, hipStream_t stream, int in_tm, int slab_f16) {
  if (M % GT_BM || N % GT_BN || ksplit < 1 || K % GT_BK || M <= 0) return hipErrorInvalidValue;
  if (in_tm && (M % TM_ROWS || N % TM_ROWS)) return hipErrorInvalidValue;
  // the 256x256 ping-pong engine is far more efficient per CU than the 128x128 one (decoder FFN inner:
  // 160 tiles on 256 CUs still beat 640 small tiles); use it when the units roughly fill the chip once
  // and every unit has a real K loop (its K parts may be unequal)
  const int units256 = (M / G2_BM) * (N / G2_BN) * ksplit;
  const int min_units = tune(TUNE_G2_SPLITK_MIN, 96);  // A/B switch (an atomic load per launch)
  const size_t ps = (size_t)M * N * (slab_f16 ? 2 : 4);
  const bool big = M % G2_BM == 0 && N % G2_BN == 0 && (K / G2_BK) / ksplit >= 16 && units256 >= min_units && units256 <= num_cus();
  if (!big && K % (GT_BK * ksplit)) return hipErrorInvalidValue;  // the 128x128 engine splits K evenly
  if (slab_f16) {
    // ... and its FFN-output projection: 8 x 4 tiles x 8 K parts = 256 lone units (gemm_v2_lone.hip)
    if (in_tm && gemm_v2_lone_fits(M, N, K, ksplit)) return launch_gemm_v2_lone(0, X, W, bias, parts, M, N, K, ksplit, stream);
    if (big)
      return in_tm ? launch_one256<EPI_BIAS_F16, 1>(X, W, bias, parts, M, N, K, N, stream, nullptr, ksplit, ps)
                   : launch_one256<EPI_BIAS_F16, 0>(X, W, bias, parts, M, N, K, N, stream, nullptr, ksplit, ps);
    return in_tm ? launch_one<EPI_BIAS_F16, 1>(X, W, bias, parts, M, N, K, N, stream, ksplit, ps)
                 : launch_one<EPI_BIAS_F16, 0>(X, W, bias, parts, M, N, K, N, stream, ksplit, ps);
  }
  if (big)
    return in_tm ? launch_one256<EPI_STORE_F32, 1>(X, W, bias, parts, M, N, K, N, stream, nullptr, ksplit, ps)
                 : launch_one256<EPI_STORE_F32, 0>(X, W, bias, parts, M, N, K, N, stream, nullptr, ksplit, ps);
  return in_tm ? launch_one<EPI_STORE_F32, 1>(X, W, bias, parts, M, N, K, N, stream, ksplit, ps)
               : launch_one<EPI_STORE_F32, 0>(X, W, bias, parts, M, N, K, N, stream, ksplit, ps);
}

// How many K parts launch_gemm_tn_splitk should be given for a decode-time projection (M = beam x batch rows,
// N = model_dim): as many as keep EVERY unit on a CU of its own -- one round of lone tiles is the fastest a
// latency-bound launch gets -- without starving a unit of K loop.  <= max_parts (the slab buffer).
int gemm_splitk_parts(int M, int N, int K, int max_parts) {
  if (M % G2_BM == 0 && N % G2_BN == 0) {
    const int tiles = (M / G2_BM) * (N / G2_BN);
    const int ks = std::min(std::min(max_parts, num_cus() / std::max(tiles, 1)), (K / G2_BK) / 16);
    if (ks >= 1 && tiles * ks >= 96) return ks;
  }
  if (lone_enabled()) {
    // Lone-tile units: the part count with the cheapest launch by a two-term model -- K tiles per unit x the time of one
    // K tile (measured: 0.15 us for a 64x64 unit with a CU of its own, 0.25 us with two per CU, 0.41 us for a 128x128
    // ring unit), plus what every part adds around the launch (its slab is written here and read by the consumer:
    // 8 bytes per output element at ~20 TB/s, it is L2 / Infinity-Cache traffic); a unit keeps at least 4 K tiles.
    // SMI_LONE_KS overrides (A/B runs).
    if (const int v = tune(TUNE_LONE_KS, 0)) {
      if (v >= 1 && v <= max_parts && K % (GT_BK * v) == 0 && (lone_fits(M, N, v) || ring_fits(M, N, v))) return v;
    }
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= max_parts; ks *= 2) {
      if (K % (GT_BK * ks) || (ks > 1 && K / ks < 4 * GT_BK)) break;
      double t_tile;
      if (lone_fits(M, N, ks))
        t_tile = (int64_t)(M / 64) * (N / 64) * ks <= num_cus() ? 0.15 : 0.25;
      else if (ring_fits(M, N, ks))
        t_tile = 0.41;
      else
        break;
      const double cost = (K / ks / GT_BK) * t_tile + ks * ((double)M * N * 8.0 / 20e6);
      if (cost < best_cost) {
        best_cost = cost;
        best = ks;
      }
    }
    return best;
  }
  const int tiles = (M / GT_BM) * (N / GT_BN);
  int ks = 1;
  while (ks * 2 <= max_parts && tiles * ks * 2 <= num_cus() && K % (GT_BK * ks * 2) == 0 && K / (ks * 2) >= 2 * GT_BK)
    ks *= 2;
  return ks;
}

}  // namespace smi

#ifdef SMI_GEMM_TRACE
extern "C" int smi_debug_gemm_splitk(const void* x, const void* w, const float* bias, float* parts, int m, int n, int k,
                                     int ksplit, int in_tm, void* stream) {
  return (int)smi::launch_gemm_tn_splitk((const smi::f16*)x, (const smi::f16*)w, bias, parts, m, n, k, ksplit,
                                         (hipStream_t)stream, in_tm, 0);
}
#endif
