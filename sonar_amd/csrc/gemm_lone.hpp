// LONE-TILE engine (round 4): GEMM units for launches whose units are all resident on the chip at once -- the encoder at
// small batches (the reference's default predict(batch_size=5): 190-222 tokens = ONE 256-row panel), the decoder's
// narrow projections, the poolers.  Such a launch is a chain of fixed costs (~6 us: launch gap, ramp, the first operand
// latency, epilogue, drain) plus ONE unit's K loop, and the K loop scales with the operand bytes a unit pulls through
// its CU (measured 0.15 / 0.34 / 0.41 us per K tile for 64x64 / 128x64 / 128x128 units: fragment reads + DMA writes
// are 3x the stage bytes of LDS traffic), neither with the ring depth nor with how reads and MFMAs overlap.  What
// shortens it is a SMALLER unit on MORE CUs:
//
//  * unit shape BM x BN over K tiles of 64 (the LDS image and bank swizzle of gemm_tile.hpp: 128-B rows, 16-B chunk c
//    of row r at slot c ^ ((r>>1)&7), v_mfma_f32_32x32x16_f16, W = MFMA A operand, X = B operand); 4 waves as
//    2(m) x 2(n), wave tile BM/2 x BN/2.  The launcher uses 64x64 (gemm.hip: lone_fits): at M = 256 the fused QKV
//    projection becomes 192 units instead of 48, the FFN projections 512 instead of 128, each with a quarter of the
//    MFMAs and half the LDS bytes of a 128x128 unit.
//  * a ring of 4 stages = 64 KiB for 64x64, so TWO workgroups share a CU when a launch has more units than CUs.  Deeper
//    rings were measured and lost (8 stages: +0.4 us per launch; every stage of the fill is issued before the first
//    wait), and so did larger units (r04 experiment 11).
//  * a stage's slot goes back to the DMA as soon as its fragments are in REGISTERS: the fragments of stage t+1 are read
//    while the MFMAs of stage t run (two fragment sets, the loop is unrolled by two).  Measured neutral against
//    retiring the reads first (MFMA issue is asynchronous, so the ring of gemm_tile.hpp overlapped them already); kept
//    because it frees a slot one step earlier.
//
// One interval (step t), all four waves in the same phase:
//     ds_read fragments of stage t+1 -> set B | DMA of stage t+ST into slot t%ST | MFMAs of stage t from set A |
//     counted vmcnt: MY pieces of stage t+2 have landed | lgkmcnt(0) | s_barrier
// RAW  stage t+1 is read in step t: every wave passed the vmcnt for its pieces of stage t+1 before the barrier that
//      ended step t-1 (the prologue for t = 0).
// WAR  slot t%ST is refilled in step t: it held stage t, whose fragment reads were issued in step t-1 and retired
//      (lgkmcnt(0)) by every wave before the barrier that ended step t-1.
// vmcnt retires in issue order: after the issue of step t the stages younger than t+2 are t+3 .. min(nt-1, t+ST),
// CPW DMA instructions per wave each.
#pragma once
#include "gemm_tile.hpp"

namespace smi {

#ifdef SMI_GEMM_TRACE  // development aid: phase timestamps of thread 0 of workgroups 0..15 (gemm.hip: g2_trace_buf)
extern __device__ unsigned long long g2_trace_buf[16 * 64 * 8];
#define LONE_TRACE(slot) \
  if (threadIdx.x == 0 && blockIdx.x < 16 && blockIdx.y == 0) g2_trace_buf[(blockIdx.x * 64) * 8 + (slot)] = wall_clock64()
#else
#define LONE_TRACE(slot)
#endif

template <int BM, int BN>
struct LoneShape {
  static_assert((BM == 128 || BM == 64) && (BN == 128 || BN == 64), "unit shapes: 128x128, 128x64, 64x64");
  static constexpr int MI = BM / 64, NI = BN / 64;             // 32-row MFMA blocks of a wave along m / n
  static constexpr int STAGE_BYTES = (BM + BN) * GT_BK * 2;    // X rows, then W rows, 128 B each
  static constexpr int CPW = (BM + BN) / 32;                   // 1-KiB DMA pieces per wave and stage
#ifdef SMI_LONE_ST  // probe builds: another ring depth
  static constexpr int STAGES = SMI_LONE_ST;
#else
  static constexpr int STAGES = 4;
#endif
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;       // 64 KiB for 64x64
  static constexpr int WG_PER_CU = 160 * 1024 / LDS_BYTES >= 2 ? 2 : 1;
};

template <int N>
__device__ __forceinline__ void lone_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// at most j * CPW of this wave's DMA instructions still in flight (j is wave-uniform, 0 <= j < 8)
template <int CPW>
__device__ __forceinline__ void lone_wait_stages(int j) {
  switch (j) {
    case 0: lone_wait_vm<0>(); break;
    case 1: lone_wait_vm<CPW>(); break;
    case 2: lone_wait_vm<2 * CPW>(); break;
    case 3: lone_wait_vm<3 * CPW>(); break;
    case 4: lone_wait_vm<4 * CPW>(); break;
    case 5: lone_wait_vm<5 * CPW>(); break;
    case 6: lone_wait_vm<6 * CPW>(); break;
    default: lone_wait_vm<7 * CPW>(); break;
  }
}

template <int BM, int BN>
struct LoneFrag {
  half8 w[4][LoneShape<BM, BN>::NI];
  half8 x[4][LoneShape<BM, BN>::MI];
};

// one 16-B LDS read, NOT waited for: the destination is valid after lone_retire
template <int OFF>
__device__ __forceinline__ void lone_ds_read(half8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// Retire this wave's LDS reads, then the workgroup barrier.  Every register of the fragment set is an in/out operand: the
// compiler may neither use nor copy a fragment between its read and this statement's results.
template <int BM, int BN>
__device__ __forceinline__ void lone_retire(LoneFrag<BM, BN>& f) {
  constexpr int MI = LoneShape<BM, BN>::MI, NI = LoneShape<BM, BN>::NI;
#define SMI_LONE_RETIRE "s_waitcnt lgkmcnt(0)\n\ts_barrier"
  if constexpr (MI == 1 && NI == 1)
    asm volatile(SMI_LONE_RETIRE
                 : "+v"(f.w[0][0]), "+v"(f.w[1][0]), "+v"(f.w[2][0]), "+v"(f.w[3][0]), "+v"(f.x[0][0]), "+v"(f.x[1][0]),
                   "+v"(f.x[2][0]), "+v"(f.x[3][0])::"memory");
  else if constexpr (MI == 2 && NI == 1)
    asm volatile(SMI_LONE_RETIRE
                 : "+v"(f.w[0][0]), "+v"(f.w[1][0]), "+v"(f.w[2][0]), "+v"(f.w[3][0]), "+v"(f.x[0][0]), "+v"(f.x[1][0]),
                   "+v"(f.x[2][0]), "+v"(f.x[3][0]), "+v"(f.x[0][1]), "+v"(f.x[1][1]), "+v"(f.x[2][1]), "+v"(f.x[3][1])::"memory");
  else
    asm volatile(SMI_LONE_RETIRE
                 : "+v"(f.w[0][0]), "+v"(f.w[1][0]), "+v"(f.w[2][0]), "+v"(f.w[3][0]), "+v"(f.x[0][0]), "+v"(f.x[1][0]),
                   "+v"(f.x[2][0]), "+v"(f.x[3][0]), "+v"(f.x[0][1]), "+v"(f.x[1][1]), "+v"(f.x[2][1]), "+v"(f.x[3][1]),
                   "+v"(f.w[0][1]), "+v"(f.w[1][1]), "+v"(f.w[2][1]), "+v"(f.w[3][1])::"memory");
#undef SMI_LONE_RETIRE
}

// acc[ni][mi][r] is C[m][n] with  m = m0 + wm*(BM/2) + mi*32 + (lane&31),
//                                 n = n0 + wn*(BN/2) + ni*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
// X: [*, K], W: [*, K] row-major, or both tile-major (common.hpp) when TM; rows m0..m0+BM-1 / n0..n0+BN-1 readable;
// the K loop covers columns [k0, k0 + klen), k0 % 64 == 0, klen % 64 == 0, klen >= 64.
template <bool TM, int BM, int BN, int NI, int MI>
__device__ __forceinline__ void lone_mainloop(f32x16 (&acc)[NI][MI], const f16* __restrict__ X,
                                              const f16* __restrict__ W, int K, int m0, int n0, char* smem, int k0,
                                              int klen) {
  static_assert(NI == LoneShape<BM, BN>::NI && MI == LoneShape<BM, BN>::MI, "accumulator blocks of the unit shape");
  using S = LoneShape<BM, BN>;
  constexpr int ST = S::STAGES, CPW = S::CPW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // DMA sources: piece c = wave*CPW + q covers stage rows c*8 .. c*8+7 (X rows first, then W rows); lane -> row
  // c*8 + (lane>>3), LDS slot lane&7 holds the global chunk slot ^ f(row)
  const f16* src[CPW];
#pragma unroll
  for (int q = 0; q < CPW; ++q) {
    const int c = wave * CPW + q;
    const int row = c * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const bool isx = c < BM / 8;  // wave-uniform
    const f16* base = isx ? X : W;
    const int r = isx ? m0 + row : n0 + row - BM;
    if constexpr (TM)
      src[q] = base + tm_offset(r, k0 + chunk * 8, K);
    else
      src[q] = base + (size_t)r * K + k0 + chunk * 8;
  }
  constexpr int kstep = TM ? 2 * TM_BLOCK : GT_BK;  // elements per K tile along the source
  auto issue = [&](int s) {
    char* stage = smem + (s % ST) * S::STAGE_BYTES + wave * CPW * 1024;
#pragma unroll
    for (int q = 0; q < CPW; ++q) glds16(src[q] + (size_t)s * kstep, stage + q * 1024);
  };

  // Fragment reads are inline asm: hipcc's own lgkmcnt bookkeeping is conservative across the loop's back edge (it put
  // lgkmcnt(0) in front of the MFMAs of stage t, i.e. waited for the reads of stage t+1 issued just before them); the
  // values are handed to the compiler by lone_retire (the wait + barrier statement re-defines every register of the set).
  const int l31 = lane & 31, hi = lane >> 5;
  const int t_sw = (hi ^ ((l31 >> 1) & 7)) << 4;
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned xoff[4], woff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    xoff[ks] = lds0 + (wm * (BM / 2) + l31) * 128 + (t_sw ^ (ks << 5));
    woff[ks] = lds0 + BM * 128 + (wn * (BN / 2) + l31) * 128 + (t_sw ^ (ks << 5));
  }
  auto read_frags = [&](int s, LoneFrag<BM, BN>& f) {
    const unsigned so = (s % ST) * S::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      lone_ds_read<0>(f.w[ks][0], woff[ks] + so);
      if constexpr (NI == 2) lone_ds_read<32 * 128>(f.w[ks][NI - 1], woff[ks] + so);
      lone_ds_read<0>(f.x[ks][0], xoff[ks] + so);
      if constexpr (MI == 2) lone_ds_read<32 * 128>(f.x[ks][MI - 1], xoff[ks] + so);
    }
  };

#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int nt = klen / GT_BK;
  const int filled = min(nt, ST);
  LONE_TRACE(3);  // addresses set up
#pragma unroll
  for (int s = 0; s < ST; ++s)
    if (s < nt) issue(s);
  LONE_TRACE(0);                      // fill issued
  lone_wait_stages<CPW>(filled - 1);  // my pieces of stage 0
  asm volatile("s_barrier" ::: "memory");
  LONE_TRACE(1);                      // stage 0 landed
  LoneFrag<BM, BN> fa, fb;
  read_frags(0, fa);
  if (nt > 1) lone_wait_stages<CPW>(filled - 2);  // my pieces of stage 1
  lone_retire(fa);

#ifdef SMI_GEMM_TRACE
  unsigned long long ph_read = 0, ph_dma = 0, ph_mfma = 0, ph_wait = 0;
#define LONE_PH(acc_, t0_) { const unsigned long long now_ = __builtin_readcyclecounter(); acc_ += now_ - t0_; t0_ = now_; }
#else
#define LONE_PH(acc_, t0_)
#endif
  auto step = [&](int t, const LoneFrag<BM, BN>& cur, LoneFrag<BM, BN>& nxt) {
#ifdef SMI_GEMM_TRACE
    unsigned long long t0 = __builtin_readcyclecounter();
#endif
    if (t + 1 < nt) read_frags(t + 1, nxt);
    LONE_PH(ph_read, t0)
    if (t + ST < nt) issue(t + ST);
    LONE_PH(ph_dma, t0)
#ifdef SMI_LONE_SERIAL  // probe build: fragment reads retired BEFORE the MFMAs (no overlap inside a wave)
    if (t + 2 < nt) lone_wait_stages<CPW>(min(nt - 1, t + ST) - (t + 2));
    lone_retire(nxt);
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.w[ks][ni], cur.x[ks][mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    LONE_PH(ph_mfma, t0)
#ifndef SMI_LONE_SERIAL
    if (t + 2 < nt) lone_wait_stages<CPW>(min(nt - 1, t + ST) - (t + 2));
    lone_retire(nxt);
#endif
    LONE_PH(ph_wait, t0)
  };
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    step(t, fa, fb);
    step(t + 1, fb, fa);
  }
  if (t < nt) step(t, fa, fb);
  LONE_TRACE(2);  // K loop done
#ifdef SMI_GEMM_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 16 && blockIdx.y == 0) {  // shader-clock cycles per phase, summed over the steps
    g2_trace_buf[(blockIdx.x * 64 + 1) * 8 + 0] = ph_read;
    g2_trace_buf[(blockIdx.x * 64 + 1) * 8 + 1] = ph_dma;
    g2_trace_buf[(blockIdx.x * 64 + 1) * 8 + 2] = ph_mfma;
    g2_trace_buf[(blockIdx.x * 64 + 1) * 8 + 3] = ph_wait;
  }
#endif
}

// XCD-aware grouped raster (gt_tile_coords) for any unit shape: consecutive logical ids are the row units of ONE
// column unit, so the units that share a W part sit on the same XCD.
__device__ __forceinline__ void lone_tile_coords(int ntm, int ntn, int& tile_m, int& tile_n) {
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
