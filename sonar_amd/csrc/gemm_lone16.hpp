// LONE-TILE engine, second form (round 4, experiment 17): 64x64 units on TILE-MAJOR operands with NO LDS in the K loop.
//
// The phase trace of gemm_lone.hpp's unit (tools/lone_trace.py, profiles/r04w_lone_phase_trace.log): a 64x64x1024 unit with a CU
// to itself spends 5.3 of its 7.1 in-kernel us in the K loop, and per K = 64 step ~335 of ~585 cycles waiting for its own
// fragment reads -- 48 KiB of LDS traffic per step (the 2 x 2 wave grid reads every operand row twice, the DMA writes it once)
// through a 128 B/clk LDS, all four waves in the same phase.  The bytes a unit NEEDS are 16 KiB per step, and the tile-major
// layout (common.hpp) already stores them as the v_mfma_f32_16x16x32_f16 fragment image: a 16-row x 32-column block is one
// contiguous 1 KiB run in which lane l's 16 bytes sit at (row l & 15, slot (l >> 4) ^ swz(row)).  So:
//
//  * a wave loads its fragments straight from global memory into registers (16 B per lane, every instruction a dense
//    1 KiB run: the TA's 64 B/clk is the only limit, 16 KiB per 64 columns of K = 256 cycles per CU);
//  * the four waves split K, not the tile: wave w owns the 32-column k-blocks w, w + 4, ... and accumulates the WHOLE
//    64x64 tile for them (4 x 4 blocks of 16x16 = 64 accumulator registers), so no operand byte is read twice and
//    there is no barrier in the loop -- the waves drift apart and overlap each other's loads and MFMAs;
//  * 4 k-blocks of loads (32 registers each) are in flight per wave = 128 KiB per CU, from the first instruction;
//  * the loop is fully unrolled (NKB = k-blocks per wave: 2, 4 or 8, i.e. K per unit 256 / 512 / 1024): straight-line
//    code, for which hipcc's vmcnt bookkeeping is exact;
//  * after the loop the four partial tiles meet in LDS (64 KiB, the only LDS the kernel uses: two workgroups per CU):
//    wave w adds the four partials of row strip w in the fixed order 0..3 and runs the epilogue for 16 rows x 64 columns.
//
// The fp32 summation order over K differs from the other 128x128-family engines (k-blocks w, w+4, ... per partial, then
// four partials): results are equal to them within fp32 rounding, not bit for bit.
#pragma once
#include "common.hpp"

namespace smi {

constexpr int L16_BM = 64, L16_BN = 64, L16_THREADS = 256, L16_LDS_BYTES = 64 * 1024;

// acc[ni][mi][r] is C[m][n] with  m = m0 + mi*16 + (lane & 15),  n = n0 + ni*16 + 4*(lane >> 4) + r
// X, W tile-major [*, K]; rows m0..m0+63 / n0..n0+63; this unit's K range is [k0, k0 + 128 * NKB).
template <int NKB>
__device__ __forceinline__ void lone16_mainloop(f32x4 (&acc)[4][4], const f16* __restrict__ X, const f16* __restrict__ W,
                                                int K, int m0, int n0, int k0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, kg = lane >> 4;
  // block (row-block, kb) of a tile-major matrix starts at ((row >> 8) * (K >> 5) + kb) * TM_BLOCK elements; inside it
  // row rr sits at rr * 32 elements, chunk c at slot c ^ tm_swz(rr)
  const f16* xb = X + ((size_t)(m0 >> 8) * (K >> 5) + (k0 >> 5) + wave) * TM_BLOCK;
  const f16* wb = W + ((size_t)(n0 >> 8) * (K >> 5) + (k0 >> 5) + wave) * TM_BLOCK;
  int xo[4], wo[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int xr = (m0 & 255) + b * 16 + l15, wr = (n0 & 255) + b * 16 + l15;
    xo[b] = xr * 32 + ((kg ^ tm_swz(xr)) << 3);
    wo[b] = wr * 32 + ((kg ^ tm_swz(wr)) << 3);
  }
  // Loads in flight per wave: DW / DX k-blocks of W / X, 4 each (32 registers per k-block, 128 KiB in flight per CU).  Measured
  // against ALL of W up front (248 registers) and against X two blocks ahead: the same 1.32-1.34 ms per batch-of-5 forward
  // (r04 experiment 17) -- the unit is not waiting for operand latency.  Probe builds: -DSMI_L16_DW / -DSMI_L16_DX.
#ifdef SMI_L16_DW
  constexpr int DW = NKB < SMI_L16_DW ? NKB : SMI_L16_DW;
#else
  constexpr int DW = NKB < 4 ? NKB : 4;
#endif
#ifdef SMI_L16_DX
  constexpr int DX = NKB < SMI_L16_DX ? NKB : SMI_L16_DX;
#else
  constexpr int DX = NKB < 4 ? NKB : 4;
#endif
  half8 fx[DX][4], fw[DW][4];
  auto load_w = [&](int i, int slot) {  // this wave's i-th k-block = global k-block wave + 4 * i
#pragma unroll
    for (int b = 0; b < 4; ++b) fw[slot][b] = *(const half8*)(wb + (size_t)i * 4 * TM_BLOCK + wo[b]);
  };
  auto load_x = [&](int i, int slot) {
#pragma unroll
    for (int b = 0; b < 4; ++b) fx[slot][b] = *(const half8*)(xb + (size_t)i * 4 * TM_BLOCK + xo[b]);
  };
  // issue order = consumption order (vmcnt retires in order): W0 X0 W1 X1 ... then the W blocks past DX
#pragma unroll
  for (int i = 0; i < DW; ++i) {
    load_w(i, i);
    if (i < DX) load_x(i, i);
  }
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  // sched_barriers pin the order "loads in flight | MFMAs of block i | next loads | ...": left alone, the machine scheduler
  // minimises registers and turns the loop into load -> vmcnt(0) -> MFMA with one k-block in flight
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NKB; ++i) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i % DW][ni], fx[i % DX][mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (i + DW < NKB) load_w(i + DW, i % DW);
    if (i + DX < NKB) load_x(i + DX, i % DX);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The four waves' partial tiles -> LDS -> wave w keeps row strip w: out[ni] = sum over partials 0..3 of block (ni, w).
// red: 64 KiB, [partial][mi][ni][lane] f32x4.
__device__ __forceinline__ void lone16_reduce(const f32x4 (&acc)[4][4], f32x4 (&out)[4], char* red) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4* r4 = (f32x4*)red;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) r4[((wave * 4 + mi) * 4 + ni) * 64 + lane] = acc[ni][mi];
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    f32x4 s = r4[((0 * 4 + wave) * 4 + ni) * 64 + lane];
#pragma unroll
    for (int p = 1; p < 4; ++p) s += r4[((p * 4 + wave) * 4 + ni) * 64 + lane];
    out[ni] = s;
  }
}

}  // namespace smi
