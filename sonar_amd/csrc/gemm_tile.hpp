// 128x128x64 fp16 MFMA tile engine shared by the dense GEMMs and the xsim
// mining kernel.  C[m][n] = sum_k X[m][k] * W[n][k]  (both operands K-major,
// i.e. the nn.Linear layout the reference's checkpoints use:
// sonar/models/sonar_text/factory.py:130-153 builds Linear(weight[out,in])).
//
// Mapping to CDNA4:
//  * 256 threads = 4 waves in a 2(m) x 2(n) grid, each wave owns 64x64 of C as
//    2x2 v_mfma_f32_32x32x16_f16 blocks (64 accumulator VGPRs).
//  * The MFMA "A" operand is the W tile and the "B" operand the X tile, so the
//    accumulator of a lane holds ONE output row m = lane&31 and 4-element
//    contiguous runs along n: epilogues store 8/16 B per lane along the
//    contiguous dimension instead of 2 B.
//  * Operand tiles go HBM -> LDS with 16-byte global_load_lds DMA (no VGPR
//    round trip).  The DMA writes LDS lane-linearly, so the bank swizzle is
//    applied to the per-lane SOURCE address and undone by the ds_read_b128
//    address: 16-B chunk c of tile row r lives at slot c ^ ((r>>1)&7).  With
//    128-B rows this makes every 16-lane ds_read_b128 group hit 16 distinct
//    16-B slots of the 256-B bank row (conflict free).
//  * 2 LDS stages (64 KiB) -> 2 workgroups per CU; one barrier per K tile, the
//    DMA for tile t+1 is in flight while tile t is multiplied.
#pragma once
#include "common.hpp"

namespace smi {

constexpr int GT_BM = 128;
constexpr int GT_BN = 128;
constexpr int GT_BK = 64;
constexpr int GT_THREADS = 256;
constexpr int GT_STAGE_BYTES = (GT_BM + GT_BN) * GT_BK * 2;  // 32 KiB
constexpr int GT_LDS_BYTES = 2 * GT_STAGE_BYTES;             // 64 KiB

struct GemmTileAcc {
  f32x16 v[2][2];  // [ni][mi]
};

// Issue the DMA for one K tile.  xg/wg: per-lane source pointers for the four
// 1-KiB wave chunks this wave owns in each operand tile (already swizzled).
__device__ __forceinline__ void gt_issue(const f16* const (&xg)[4], const f16* const (&wg)[4],
                                         int koff, char* stage, int wave) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    glds16(xg[q] + koff, stage + (wave * 4 + q) * 1024);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    glds16(wg[q] + koff, stage + GT_BM * GT_BK * 2 + (wave * 4 + q) * 1024);
  }
}

// Multiply one resident K tile.
__device__ __forceinline__ void gt_compute(GemmTileAcc& acc, const char* stage, int xrow_off,
                                           int wrow_off, int t_sw) {
  const char* xs = stage + xrow_off;
  const char* ws = stage + GT_BM * GT_BK * 2 + wrow_off;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int coff = (t_sw ^ (ks << 5));  // ((ks*2+hi) ^ swz) << 4
    half8 fw0 = *(const half8*)(ws + coff);
    half8 fw1 = *(const half8*)(ws + 32 * 128 + coff);
    half8 fx0 = *(const half8*)(xs + coff);
    half8 fx1 = *(const half8*)(xs + 32 * 128 + coff);
    acc.v[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw0, fx0, acc.v[0][0], 0, 0, 0);
    acc.v[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw0, fx1, acc.v[0][1], 0, 0, 0);
    acc.v[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1, fx0, acc.v[1][0], 0, 0, 0);
    acc.v[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1, fx1, acc.v[1][1], 0, 0, 0);
  }
}

// K loop for the 128x128 tile at (m0, n0) over columns [k0, k0 + klen) of X: [*, K] and
// W: [*, K] (row-major, row stride K; or both in the tile-major layout of common.hpp when TM:
// the per-lane DMA source address absorbs the layout, the LDS image is the same).  Rows
// m0..m0+127 of X and n0..n0+127 of W must be readable.  k0 % 64 == 0, klen % 64 == 0
// (klen < K: one slice of a split-K GEMM).
template <bool TM = false>
__device__ __forceinline__ void gt_mainloop(GemmTileAcc& acc, const f16* __restrict__ X,
                                            const f16* __restrict__ W, int K, int m0, int n0,
                                            char* smem, int k0 = 0, int klen = -1) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // DMA source pointers: chunk c = wave*4+q covers tile rows c*8..c*8+7;
  // lane -> row c*8 + (lane>>3), LDS slot lane&7 holds global chunk slot^f(row).
  const f16* xg[4];
  const f16* wg[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = wave * 4 + q;
    const int row = c * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if constexpr (TM) {
      xg[q] = X + tm_offset(m0 + row, k0 + chunk * 8, K);
      wg[q] = W + tm_offset(n0 + row, k0 + chunk * 8, K);
    } else {
      xg[q] = X + (size_t)(m0 + row) * K + k0 + chunk * 8;
      wg[q] = W + (size_t)(n0 + row) * K + k0 + chunk * 8;
    }
  }
  constexpr int kstep = TM ? 2 * TM_BLOCK : GT_BK;  // elements per K tile along the source

  // Fragment read offsets: row = w*64 + blk*32 + (lane&31); f(row) = ((lane&31)>>1)&7.
  const int l31 = lane & 31, hi = lane >> 5;
  const int t_sw = (hi ^ ((l31 >> 1) & 7)) << 4;
  const int xrow_off = (wm * 64 + l31) * 128;
  const int wrow_off = (wn * 64 + l31) * 128;

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc.v[i][j][r] = 0.f;

  const int nt = (klen < 0 ? K : klen) / GT_BK;
  gt_issue(xg, wg, 0, smem, wave);
  for (int t = 0; t < nt; ++t) {
    // The compiler drains the outstanding LDS DMA (vmcnt(0)) ahead of this
    // barrier; after it tile t is visible to every wave and every wave has
    // finished reading the other stage.
    __syncthreads();
    if (t + 1 < nt) gt_issue(xg, wg, (t + 1) * kstep, smem + ((t + 1) & 1) * GT_STAGE_BYTES, wave);
    gt_compute(acc, smem + (t & 1) * GT_STAGE_BYTES, xrow_off, wrow_off, t_sw);
  }
}

// ---- latency-hiding variant for LONE tiles (round 3).  When a launch has no more tiles than CUs -- the decoder's
// M = beam x batch projections, the encoder at small batches, the poolers -- a workgroup has the CU to itself and
// the two-stage loop above is a chain of exposed DMA latencies: one K tile takes ~1.2 us (measured: 19.6 us for the
// 16 K tiles of the decoder's fused QKV projection) against 0.2 us of MFMA work.  Here the ring holds ST stages of
// K = 64 (ST * 32 KiB, one workgroup per CU), ST - 1 tiles are in flight, waits are COUNTED (vmcnt(8) per tile
// still allowed in flight: a wave issues 8 DMA instructions per tile) and barriers are raw s_barrier, as in
// gemm_tile256.hpp.  One interval per tile:
//   fragments of tile t -> VGPRs | DMA of tile t+ST-1 into the stage tile t-1 was read from | counted wait for
//   MY pieces of tile t+1 | lgkmcnt(0) + s_barrier | MFMAs of tile t
// RAW: a wave passes vmcnt for its pieces of tile t+1 before the barrier that ends interval t, so after that barrier
// every piece of tile t+1 is in LDS.  WAR: the stage refilled in interval t was last read in interval t-1, and every
// wave retired those reads (lgkmcnt(0)) before the barrier that ended interval t-1.
#define SMI_GT_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <bool TM, int ST>
__device__ __forceinline__ void gt_mainloop_ring(GemmTileAcc& acc, const f16* __restrict__ X,
                                                 const f16* __restrict__ W, int K, int m0, int n0,
                                                 char* smem, int k0 = 0, int klen = -1) {
  static_assert(ST == 3 || ST == 4, "ring of 3 or 4 stages");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const f16* xg[4];
  const f16* wg[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = wave * 4 + q;
    const int row = c * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if constexpr (TM) {
      xg[q] = X + tm_offset(m0 + row, k0 + chunk * 8, K);
      wg[q] = W + tm_offset(n0 + row, k0 + chunk * 8, K);
    } else {
      xg[q] = X + (size_t)(m0 + row) * K + k0 + chunk * 8;
      wg[q] = W + (size_t)(n0 + row) * K + k0 + chunk * 8;
    }
  }
  constexpr int kstep = TM ? 2 * TM_BLOCK : GT_BK;
  const int l31 = lane & 31, hi = lane >> 5;
  const int t_sw = (hi ^ ((l31 >> 1) & 7)) << 4;
  const int xrow_off = (wm * 64 + l31) * 128;
  const int wrow_off = (wn * 64 + l31) * 128;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc.v[i][j][r] = 0.f;

  const int nt = (klen < 0 ? K : klen) / GT_BK;
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (s < nt) gt_issue(xg, wg, s * kstep, smem + s * GT_STAGE_BYTES, wave);
  // tile 0 landed (mine), the rest of the fill stays in flight
  if (nt >= ST - 1) {
    if constexpr (ST == 4) SMI_GT_WAIT_VMCNT(16); else SMI_GT_WAIT_VMCNT(8);
  } else if (nt == 2) {
    SMI_GT_WAIT_VMCNT(8);
  } else {
    SMI_GT_WAIT_VMCNT(0);
  }
  asm volatile("s_barrier" ::: "memory");
  for (int t = 0; t < nt; ++t) {
    const char* stage = smem + (t % ST) * GT_STAGE_BYTES;
    const char* xs = stage + xrow_off;
    const char* ws = stage + GT_BM * GT_BK * 2 + wrow_off;
    half8 fw[4][2], fx[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = (t_sw ^ (ks << 5));
      fw[ks][0] = *(const half8*)(ws + coff);
      fw[ks][1] = *(const half8*)(ws + 32 * 128 + coff);
      fx[ks][0] = *(const half8*)(xs + coff);
      fx[ks][1] = *(const half8*)(xs + 32 * 128 + coff);
    }
    const int ahead = nt - 1 - t;  // tiles after this one
    if (ahead >= ST - 1) {
      gt_issue(xg, wg, (t + ST - 1) * kstep, smem + ((t + ST - 1) % ST) * GT_STAGE_BYTES, wave);
      if constexpr (ST == 4) SMI_GT_WAIT_VMCNT(16); else SMI_GT_WAIT_VMCNT(8);
    } else if (ahead == 2 && ST == 4) {
      SMI_GT_WAIT_VMCNT(8);
    } else {
      SMI_GT_WAIT_VMCNT(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc.v[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks][0], fx[ks][0], acc.v[0][0], 0, 0, 0);
      acc.v[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks][0], fx[ks][1], acc.v[0][1], 0, 0, 0);
      acc.v[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks][1], fx[ks][0], acc.v[1][0], 0, 0, 0);
      acc.v[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks][1], fx[ks][1], acc.v[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Accumulator coordinates: acc.v[ni][mi][r] is C[m][n] with
//   m = m0 + wm*64 + mi*32 + (lane&31)
//   n = n0 + wn*64 + ni*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
__device__ __forceinline__ int gt_row(int m0, int mi) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return m0 + (wave >> 1) * 64 + mi * 32 + (lane & 31);
}
__device__ __forceinline__ int gt_col(int n0, int ni, int quad) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return n0 + (wave & 1) * 64 + ni * 32 + 8 * quad + 4 * (lane >> 5);
}

// Grouped, XCD-aware raster: logical id -> (tile_m, tile_n).  8 tile rows are
// walked column-major so the ~64 workgroups resident on one XCD cover an
// 8x8 super-tile whose operand panels fit that XCD's 4 MiB L2.
__device__ __forceinline__ void gt_tile_coords(int ntm, int ntn, int& tile_m, int& tile_n) {
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int GM = 8;
  const int per_group = GM * ntn;
  const int group = id / per_group;
  const int first_m = group * GM;
  const int gsz = min(GM, ntm - first_m);
  const int in_group = id - group * per_group;
  tile_m = first_m + in_group % gsz;
  tile_n = in_group / gsz;
}

}  // namespace smi
