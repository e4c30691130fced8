// 256x256 fp16 MFMA tile engine, 8 waves, 4-slot K=32 LDS ring, ping-pong wave
// groups.  C[m][n] = sum_k X[m][k] * W[n][k], both operands K-major.
//
// Why this shape on MI355X (numbers from profiles/ and MI355X_MICROARCH.md):
//  * a 128x128 tile pulls 512 KiB through L2 per 33.5 MFLOP: at >1 PFLOP/s that
//    is >60% of the 34 TB/s aggregate L2 bandwidth; 256x256 halves it;
//  * one workgroup of 8 waves owns the CU (128 KiB LDS): 2 waves per SIMD.  The
//    waves of a SIMD are put in different GROUPS (waves 0-3 / 4-7) that run the
//    same loop one barrier interval apart: while one wave of the SIMD issues its
//    12 ds_read_b128 + 4 global_load_lds for a K slice, its partner runs the 16
//    MFMAs of the previous slice, so the matrix pipe never waits for LDS;
//  * the LDS ring holds 4 K=32 slices; DMA for slice t+3 is issued while slice t
//    is consumed and is waited for with a COUNTED s_waitcnt vmcnt(8) (never 0 in
//    steady state), so ~4 barrier intervals (~2k cycles) of L2/HBM latency are
//    covered.  Barriers are raw s_barrier: __syncthreads() would drain the DMA
//    queue (vmcnt(0)) every interval.
//
// LDS slice layout: X rows [256][64 B] then W rows [256][64 B]; 16-B chunk c of
// row r sits at slot c ^ ((r>>2)&3) (swizzle applied on the DMA source address,
// undone by the ds_read address) -> conflict-free ds_read_b128 for the
// 32x32x16 operand fragments.
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
#ifndef G2_RING
#define G2_RING 4  // 5 (all 160 KiB of LDS) measured no faster: buffering is not the limiter
#endif
constexpr int G2_LDS_BYTES = G2_RING * G2_SLOT_BYTES;       // 128 KiB (ring of 4) / 160 KiB (5)
constexpr int G2_CSTRIDE = 528;                            // epilogue C-tile row stride in LDS
constexpr int G2_KERNEL_LDS_BYTES = G2_LDS_BYTES > G2_BM * G2_CSTRIDE ? G2_LDS_BYTES : G2_BM * G2_CSTRIDE;

struct GemmTile256Acc {
  f32x16 v[2][4];  // [ni][mi]
};

#define SMI_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// retire this wave's LDS reads, then workgroup barrier; "memory" keeps every LDS
// access on its side of the barrier.
#define SMI_LGKM0_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SMI_BARRIER() asm volatile("s_barrier" ::: "memory")

template <int PART = 3>
__device__ __forceinline__ void g2_issue(const f16* const (&xg)[2], const f16* const (&wg)[2], int t,
                                         char* smem, int wave, int kt = -1) {
  char* slot = smem + (G2_RING == 4 ? (t & 3) : (t % G2_RING)) * G2_SLOT_BYTES + wave * 2048;
  const int koff = (kt < 0 ? t : kt) * G2_BK;
  if (PART & 1) {
    glds16(xg[0] + koff, slot);
    glds16(xg[1] + koff, slot + 1024);
  }
  if (PART & 2) {
    glds16(wg[0] + koff, slot + G2_BM * G2_BK * 2);
    glds16(wg[1] + koff, slot + G2_BM * G2_BK * 2 + 1024);
  }
}

// X: [*, K] row-major, W: [*, K]; rows m0..m0+255 / n0..n0+255 readable; K % 32 == 0.
// VAR is a timing-ablation switch (0 = product; others give WRONG results):
//   1 no in-loop DMA, 2 DMA reads full 128-B lines (wrong rows), 3 no MFMA,
//   4 no DMA + fragments read once, 6 DMA issued but never waited for,
//   7 half the DMA, 8 DMA always re-reads slice 0 (cache-hot), 10 DMA issued inside the
//   multiply segment (CORRECT results), 11 half the fragment reads (ks=1 reuses ks=0),
//   12 no MFMA + full-line DMA, 13 DMA only (no MFMA, no fragment reads, no barriers),
//   14 as 13 but cache-hot addresses, 16 as 13 but plain global_load_dwordx4 to VGPRs (no LDS),
//   17 as 13 with tile-major (contiguous 16 KiB per slice) source addresses, 18 full kernel with
//   tile-major source addresses (wrong data, right traffic pattern).
template <int VAR = 0>
__device__ __forceinline__ void g2_mainloop(GemmTile256Acc& acc, const f16* __restrict__ X,
                                            const f16* __restrict__ W, int K, int m0, int n0,
                                            char* smem) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;  // wr doubles as the ping-pong group

  const f16* xg[2];
  const f16* wg[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int row = (wave * 2 + q) * 16 + (lane >> 2);
    int chunk = (lane & 3) ^ ((row >> 2) & 3);
    if constexpr (VAR == 2 || VAR == 12) {
      row = (wave * 2 + q) * 8 + (lane >> 3);
      chunk = lane & 7;
    }
    xg[q] = X + (size_t)(m0 + row) * K + chunk * 8;
    wg[q] = W + (size_t)(n0 + row) * K + chunk * 8;
    if constexpr (VAR == 17 || VAR == 18) {
      // block (rb, kb) of 256 rows x 32 k is 16 KiB contiguous; slice t advances by 8192 halfs.
      // (koff = t*32 is added later: scale it to t*8192 by pre-multiplying the base and using
      // a 256x stride: emulate with pointer arithmetic below)
      xg[q] = X + (size_t)(m0 / 256) * (size_t)K * 256 + (wave * 2 + q) * 512 + lane * 8;
      wg[q] = W + (size_t)(n0 / 256) * (size_t)K * 256 + (wave * 2 + q) * 512 + lane * 8;
    }
  }
  constexpr int KSTEP = (VAR == 17 || VAR == 18) ? 256 : 1;  // slice stride multiplier

  const int l31 = lane & 31, hi = lane >> 5;
  const int t_sw = (hi ^ ((l31 >> 2) & 3)) << 4;
  const int xoff = (wr * 128 + l31) * 64 + t_sw;
  const int woff = G2_BM * G2_BK * 2 + (wc * 64 + l31) * 64 + t_sw;

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc.v[i][j][r] = 0.f;

  const int nt = K / G2_BK;
  g2_issue(xg, wg, 0, smem, wave, 0);
  if (nt > 1) g2_issue(xg, wg, 1, smem, wave, 1 * KSTEP);
  if (nt > 2) g2_issue(xg, wg, 2, smem, wave, 2 * KSTEP);
  if (G2_RING == 5 && nt > 3) g2_issue(xg, wg, 3, smem, wave, 3 * KSTEP);
  if (G2_RING == 5 && nt > 3) SMI_WAIT_VMCNT(12);
  else if (nt > 2) SMI_WAIT_VMCNT(8);
  else if (nt > 1) SMI_WAIT_VMCNT(4);
  else SMI_WAIT_VMCNT(0);
  SMI_BARRIER();            // slice 0 complete for everyone
  if (wr == 1) SMI_BARRIER();  // group 1 runs one interval behind

  for (int t = 0; t < nt; ++t) {
    // ---- read segment: fragments of slice t -> VGPRs, DMA for slice t+3 ----
    const char* slot = smem + (G2_RING == 4 ? (t & 3) : (t % G2_RING)) * G2_SLOT_BYTES;
    half8 fx[2][4], fw[2][2];
    if ((VAR != 4 && VAR != 13 && VAR != 14 && VAR != 16 && VAR != 17) || t == 0)
#pragma unroll
    for (int ks = 0; ks < (VAR == 11 ? 1 : 2); ++ks) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fw[ks][ni] = *(const half8*)(slot + ((woff + ni * 2048) ^ (ks << 5)));
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fx[ks][mi] = *(const half8*)(slot + ((xoff + mi * 2048) ^ (ks << 5)));
    }
    if constexpr (VAR == 11) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fw[1][ni] = fw[0][ni];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fx[1][mi] = fx[0][mi];
    }
    if (VAR == 1 || VAR == 4) {
    } else if (VAR == 10) {
      // slice t+3 is issued later, inside the multiply segment: only t+1, t+2 outstanding here
      if (t + 2 < nt) SMI_WAIT_VMCNT(4);
      else SMI_WAIT_VMCNT(0);
    } else if (G2_RING == 5 && VAR == 0) {
      // ring of 5: slice t+4 goes into the slot slice t-1 just vacated; t+2..t+4 stay in flight
      if (t + 4 < nt) {
        g2_issue(xg, wg, t + 4, smem, wave, (t + 4) * KSTEP);
        SMI_WAIT_VMCNT(12);
      } else if (t + 3 < nt) {
        SMI_WAIT_VMCNT(8);
      } else if (t + 2 < nt) {
        SMI_WAIT_VMCNT(4);
      } else {
        SMI_WAIT_VMCNT(0);
      }
    } else if (t + 3 < nt) {
      if (VAR == 7) g2_issue<1>(xg, wg, t + 3, smem, wave);
      else if (VAR == 8 || VAR == 14) g2_issue(xg, wg, t + 3, smem, wave, 0);
      else if (VAR == 16) {
        const int koff = (t + 3) * G2_BK;
        half8 a0 = *(const half8*)(xg[0] + koff), a1 = *(const half8*)(xg[1] + koff);
        half8 a2 = *(const half8*)(wg[0] + koff), a3 = *(const half8*)(wg[1] + koff);
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3));
      }
      else g2_issue(xg, wg, t + 3, smem, wave, (t + 3) * KSTEP);
      if (VAR == 7) SMI_WAIT_VMCNT(4);
      else if (VAR != 6) SMI_WAIT_VMCNT(8);  // my part of slice t+1 has landed; t+2, t+3 stay in flight
    } else if (t + 2 < nt) {
      SMI_WAIT_VMCNT(4);
    } else {
      SMI_WAIT_VMCNT(0);
    }
    if (VAR != 13 && VAR != 14 && VAR != 16 && VAR != 17) SMI_LGKM0_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
    // ---- multiply segment ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
        {
          if constexpr (VAR == 3 || VAR == 12 || VAR == 13 || VAR == 14 || VAR == 16 || VAR == 17) {
            asm volatile("" ::"v"(fw[ks][ni]), "v"(fx[ks][mi]));
          } else {
            acc.v[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks][ni], fx[ks][mi], acc.v[ni][mi], 0, 0, 0);
          }
          if constexpr (VAR == 10) {
            if (ks == 0 && ni == 0 && mi == 3 && t + 3 < nt) g2_issue<1>(xg, wg, t + 3, smem, wave);
            if (ks == 1 && ni == 0 && mi == 3 && t + 3 < nt) g2_issue<2>(xg, wg, t + 3, smem, wave);
          }
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (VAR != 13 && VAR != 14 && VAR != 16 && VAR != 17) SMI_BARRIER();
  }
  if (wr == 0) SMI_BARRIER();  // balance group 1's extra barrier
}

// acc.v[ni][mi][r] is C[m][n] with
//   m = m0 + wr*128 + mi*32 + (lane&31)
//   n = n0 + wc*64 + ni*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
__device__ __forceinline__ int g2_row(int m0, int mi) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return m0 + (wave >> 2) * 128 + mi * 32 + (lane & 31);
}
__device__ __forceinline__ int g2_col(int n0, int ni, int quad) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return n0 + (wave & 3) * 64 + ni * 32 + 8 * quad + 4 * (lane >> 5);
}

// XCD-aware grouped raster over 256x256 tiles: the ~32 workgroups resident on
// one XCD (1 per CU) cover a 4(m) x 8(n) super-tile.
__device__ __forceinline__ void g2_tile_coords(int ntm, int ntn, int& tile_m, int& tile_n) {
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int GM = 4;
  const int per_group = GM * ntn;
  const int group = id / per_group;
  const int first_m = group * GM;
  const int gsz = min(GM, ntm - first_m);
  const int in_group = id - group * per_group;
  tile_m = first_m + in_group % gsz;
  tile_n = in_group / gsz;
}

}  // namespace smi
