// 256x256 fp16 MFMA tile engine, 8 waves, 4-slot K=32 LDS ring, ping-pong wave
// groups.  C[m][n] = sum_k X[m][k] * W[n][k], both operands K-major.
//
// MFMA shape: v_mfma_f32_16x16x32_f16 (one instruction spans the whole K=32 slice of a 16x16 block).
// The chip is power-limited on real data, and tools/micro/mfma_power.hip (no memory traffic, 8 waves/CU,
// 128 accumulator registers) measures 2038 TFLOP/s for 16x16x32 against 1732 for 32x32x16 on uniform
// random operands (both 2478 on zeros): same peak rate, ~18 % less energy per flop.  Same 12 ds_read_b128
// per wave and slice, same 128 accumulator registers; a lane's accumulators are still 4-wide runs along n.
//
// Why this shape on MI355X (measurements in DESIGN.md 3.1 / profiles/):
//  * a 128x128 tile pulls 512 KiB through L2 per 33.5 MFLOP; 256x256 halves it;
//  * one workgroup of 8 waves owns the CU (128 KiB LDS): 2 waves per SIMD.  The
//    waves of a SIMD are put in different GROUPS (waves 0-3 / 4-7) that run the
//    same loop one barrier interval apart: while one wave of the SIMD issues its
//    12 ds_read_b128 + 4 global_load_lds for a K slice, its partner runs the 32
//    MFMAs of the previous slice;
//  * the LDS ring holds 4 K=32 slices; DMA for slice t+3 is issued while slice t
//    is consumed and is waited for with a COUNTED s_waitcnt vmcnt(8) (never 0 in
//    steady state).  Barriers are raw s_barrier: __syncthreads() would drain the
//    DMA queue (vmcnt(0)) every interval.  A 5-slot ring (all 160 KiB) measured
//    no faster.  The chip is power-limited under this loop (the shader clock drops
//    from 2.4 to 1.6-2.0 GHz); with the tile-major layout, the 16x16x32 MFMA and the
//    conflict-free slot swizzle the loop is no longer bound by the operand stream
//    (halving the L2->LDS bytes buys 3-9 %): 8192^3 runs at 1473 TFLOP/s on uniform
//    random operands and 2209 on zeros.
//
// Operand layouts (per operand, template flags XTM / WTM):
//  * row-major  A[r][k]: a slice is 256 rows x 64 B, rows K*2 bytes apart;
//  * TILE-MAJOR: block (r/256, k/32) is 16 KiB contiguous and already holds the LDS image
//    (row rr at rr*64, 16-B chunk c at slot c ^ tm_swz(rr), common.hpp), so a slice is ONE contiguous
//    16 KiB burst and every DMA instruction copies 1 KiB linearly.  Measured +26 % on the bare
//    operand stream and +7 % on the full GEMM versus row-major: 64-B pieces
//    2-16 KiB apart are a poor DRAM-page / L2-channel pattern.  Every producer in the
//    encoder (LayerNorm, attention, the FFN-inner epilogue, weight packing) emits it directly.
//
// LDS slice layout: X rows [256][64 B] then W rows [256][64 B]; 16-B chunk c of
// row r sits at slot c ^ tm_swz(r) (common.hpp) -> conflict-free ds_read_b128 for the
// 16x16x32 operand fragments (lane -> row l&15, chunk l>>4).
//
// Hazard bookkeeping (intervals are the spans between consecutive barriers;
// group 0 reads slice t in interval 2t and multiplies it in 2t+1, group 1 one
// interval later):
//  RAW  a wave passes `vmcnt` for ITS part of slice t+1 before the barrier that
//       closes its read segment of slice t -> every part of slice t+1 has landed
//       before interval 2t+2 at the latest for group-1 waves (closing 2t+1).
//  WAR  slot (t+3)&3 held slice t-1, last read by group 1 in interval 2t-1; all
//       ds_reads are retired (lgkmcnt(0)) BEFORE the closing barrier, and the DMA
//       into that slot is issued in intervals >= 2t.
#pragma once
#include "common.hpp"

namespace smi {

constexpr int G2_BM = 256;
constexpr int G2_BN = 256;
constexpr int G2_BK = 32;
constexpr int G2_THREADS = 512;
constexpr int G2_SLOT_BYTES = (G2_BM + G2_BN) * G2_BK * 2;  // 32 KiB
constexpr int G2_LDS_BYTES = 4 * G2_SLOT_BYTES;             // 128 KiB
constexpr int G2_KERNEL_LDS_BYTES = G2_LDS_BYTES + 32 * 1024;  // ring + second epilogue staging buffer = 160 KiB

struct GemmTile256Acc {
  f32x4 v[4][8];  // [ni][mi]: 16x16 blocks of the wave's 64(n) x 128(m) tile
};

#define SMI_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// retire this wave's LDS reads, then workgroup barrier; "memory" keeps every LDS
// access on its side of the barrier.
#define SMI_LGKM0_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SMI_BARRIER() asm volatile("s_barrier" ::: "memory")

// Per-wave DMA source of one tile: pointers of the two 1-KiB pieces this wave copies per operand
// and slice, and the element step from one K slice to the next.
struct G2Src {
  const f16* xg[2];
  const f16* wg[2];
  int xstep, wstep;
};

template <bool TM>
__device__ __forceinline__ void g2_src(const f16* __restrict__ A, int K, int row0, int k0, int wave, int lane,
                                       const f16* (&ag)[2], int& kstep) {
  if constexpr (TM) {
    // block (row0/256, kb) starts at ((row0/256)*(K/32) + kb) * 8192 elements; piece i at i*512
    const f16* base = A + ((size_t)(row0 >> 8) * (K >> 5) + (k0 >> 5)) * TM_BLOCK;
#pragma unroll
    for (int q = 0; q < 2; ++q) ag[q] = base + (wave * 2 + q) * 512 + lane * 8;
    kstep = TM_BLOCK;
  } else {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = (wave * 2 + q) * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ tm_swz(row);
      ag[q] = A + (size_t)(row0 + row) * K + k0 + chunk * 8;
    }
    kstep = G2_BK;
  }
}

// X: [*, K], W: [*, K] (row-major or tile-major per XTM / WTM); rows m0..m0+255 / n0..n0+255
// readable; K % 32 == 0; the K loop starts at column k0 (k0 % 32 == 0; split-K units).
template <bool XTM, bool WTM>
__device__ __forceinline__ G2Src g2_make_src(const f16* __restrict__ X, const f16* __restrict__ W, int K,
                                             int m0, int n0, int k0 = 0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  G2Src s;
  g2_src<XTM>(X, K, m0, k0, wave, lane, s.xg, s.xstep);
  g2_src<WTM>(W, K, n0, k0, wave, lane, s.wg, s.wstep);
  return s;
}

__device__ __forceinline__ void g2_issue(const G2Src& s, int t, char* smem, int wave) {
  char* slot = smem + (t & 3) * G2_SLOT_BYTES + wave * 2048;
  const int xo = t * s.xstep, wo = t * s.wstep;
  glds16(s.xg[0] + xo, slot);
  glds16(s.xg[1] + xo, slot + 1024);
  glds16(s.wg[0] + wo, slot + G2_BM * G2_BK * 2);
  glds16(s.wg[1] + wo, slot + G2_BM * G2_BK * 2 + 1024);
}

// Pipeline fill: DMA for slices 0..2 into ring slots 0..2.  Issued BEFORE the previous tile's
// epilogue (which stages through slot 3 and the LDS above the ring), so the fill latency of a
// tile is hidden behind the epilogue of the one before it.
__device__ __forceinline__ void g2_prefetch(const G2Src& s, int nt, char* smem) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  g2_issue(s, 0, smem, wave);
  if (nt > 1) g2_issue(s, 1, smem, wave);
  if (nt > 2) g2_issue(s, 2, smem, wave);
}

// Tile start: retire the pipeline fill issued by g2_prefetch together with the previous epilogue's
// stores (vmcnt does not tell loads from stores) and the bias slice written to LDS by the caller; after
// this only counted waits.  The accumulators start from the bias of their columns (bias_lds: the
// tile's 256 bias values, zeros if there is none), so no epilogue has to add it.
// pend16 (round 5, register-direct epilogues only): the previous tile's >= 16 vector-memory operations per wave that were
// issued AFTER the fill -- its 16 output stores -- may stay in flight: vmcnt retires vector-memory operations in issue order
// on this ISA, so the fill is complete when at most 16 operations are outstanding.
__device__ __forceinline__ void g2_begin(GemmTile256Acc& acc, const float* bias_lds, bool pend16 = false) {
  if (pend16)
    SMI_WAIT_VMCNT(16);
  else
    SMI_WAIT_VMCNT(0);
  SMI_LGKM0_BARRIER();  // slices 0..2 and the bias slice complete for everyone
  const int lane = threadIdx.x & 63, wc = (threadIdx.x >> 6) & 3;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const f32x4 b = *(const f32x4*)(bias_lds + wc * 64 + ni * 16 + 4 * (lane >> 4));
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) acc.v[ni][mi] = b;
  }
}

// K loop of one tile (after g2_begin).  pend16: see g2_begin -- the stores sit between the fill (slices 0..2) and the DMA of
// slices 3, 4 in the queue, so the waits of iterations 0 and 1 ("my part of slice t + 1 has landed", t + 1 <= 2) tolerate
// them; iteration 2 needs slice 3, which is younger than the stores, and so retires them: the store burst of a tile (16 x
// 1 KiB per wave, ~2.7 us at the CU's store rate, round-2 experiment 26) overlaps the first two K slices of the next tile.
__device__ __forceinline__ void g2_mainloop(GemmTile256Acc& acc, const G2Src& src, int nt, char* smem, bool pend16 = false) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;  // wr doubles as the ping-pong group

  // fragment of a 16-row block: lane -> row l&15, 16-B chunk l>>4 (the 8 k values of its MFMA slot)
  const int l15 = lane & 15, kg = lane >> 4;
  const int t_sw = (kg ^ tm_swz(l15)) << 4;
  const int xoff = (wr * 128 + l15) * 64 + t_sw;
  const int woff = G2_BM * G2_BK * 2 + (wc * 64 + l15) * 64 + t_sw;

  if (wr == 1) SMI_BARRIER();  // group 1 runs one interval behind

  for (int t = 0; t < nt; ++t) {
    // ---- read segment: fragments of slice t -> VGPRs, DMA for slice t+3 ----
    const char* slot = smem + (t & 3) * G2_SLOT_BYTES;
    half8 fx[8], fw[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fw[ni] = *(const half8*)(slot + woff + ni * 1024);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) fx[mi] = *(const half8*)(slot + xoff + mi * 1024);
    if (t + 3 < nt) {
      g2_issue(src, t + 3, smem, wave);
      if (pend16 && t < 2)
        SMI_WAIT_VMCNT(24);  // ... and the previous tile's >= 16 stores, which are older than slices 3, 4 but younger than 0..2
      else
        SMI_WAIT_VMCNT(8);  // my part of slice t+1 has landed; t+2, t+3 stay in flight
    } else if (t + 2 < nt) {
      SMI_WAIT_VMCNT(4);
    } else {
      SMI_WAIT_VMCNT(0);
    }
    SMI_LGKM0_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
    // ---- multiply segment: W is the MFMA A operand, X the B operand ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
        acc.v[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ni], fx[mi], acc.v[ni][mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    SMI_BARRIER();
  }
  if (wr == 0) SMI_BARRIER();  // balance group 1's extra barrier
}

// ---- epilogue staging: the C tile leaves through LDS in 4 passes of 64 rows (pass p = the waves'
// mi block: tile rows wr*128 + p*32 + (0..31)), two 32 KiB buffers used alternately, placed in
// ring slot 3 and the 32 KiB above the ring so slots 0..2 can already take the next tile's fill.
// A staging row is 512 B; 16-B chunk c of local row lr sits at c ^ g2_stage_swz(lr): the 8/16-B
// accumulator writes (32 rows, one chunk), the whole-row reads and the 16-row x 64-B tile-major
// reads are all bank-conflict free.
constexpr int G2_STAGE0 = 3 * G2_SLOT_BYTES;         // 96 KiB
constexpr int G2_STAGE_BYTES = 64 * 512;             // 32 KiB
__device__ __forceinline__ int g2_stage_swz(int lr) { return ((lr & 3) << 2) | ((lr >> 2) & 3) | (lr & 16); }
__device__ __forceinline__ char* g2_stage(char* smem, int buf) { return smem + G2_STAGE0 + (buf & 1) * G2_STAGE_BYTES; }

// acc.v[ni][mi][r] is C[m][n] with
//   m = m0 + wr*128 + mi*16 + (lane&15)
//   n = n0 + wc*64 + ni*16 + 4*(lane>>4) + r
__device__ __forceinline__ int g2_row(int m0, int mi) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return m0 + (wave >> 2) * 128 + mi * 16 + (lane & 15);
}
__device__ __forceinline__ int g2_col(int n0, int ni) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return n0 + (wave & 3) * 64 + ni * 16 + 4 * (lane >> 4);
}

// XCD-aware grouped raster over 256x256 tiles: the 32 workgroups resident on one XCD (1 per CU)
// cover an 8(m) x 4(n) super-tile, and with N/256 = 32 column tiles a persistent round is exactly one
// group, so an XCD keeps the SAME 4 W panels (2 MiB of its 4 MiB L2) round after round while the X
// panels stream through (measured against 4 x 8: FFN-inner 45.3 -> 44.7 ms per step).
__device__ __forceinline__ void g2_tile_coords_of(int id, int ntm, int ntn, int& tile_m, int& tile_n) {
  constexpr int GM = 8;
  const int per_group = GM * ntn;
  const int group = id / per_group;
  const int first_m = group * GM;
  const int gsz = min(GM, ntm - first_m);
  const int in_group = id - group * per_group;
  tile_m = first_m + in_group % gsz;
  tile_n = in_group / gsz;
}
__device__ __forceinline__ void g2_tile_coords(int ntm, int ntn, int& tile_m, int& tile_n) {
  g2_tile_coords_of(xcd_remap(blockIdx.x, gridDim.x), ntm, ntn, tile_m, tile_n);
}

}  // namespace smi
