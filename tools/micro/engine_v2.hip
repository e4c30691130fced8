// Engine-v2 probe (round 6): K loop of a 256x256 fp16 tile with FOUR waves -- one wave per SIMD, a
// 128(m) x 128(n) wave tile (256 accumulator registers), in-wave software pipelining -- against today's 8-wave ping-pong
// loop (sonar_amd/csrc/gemm_tile256.hpp), same data, same raster, same box, alternating launches.
//
//   v1 (today): 8 waves, 128 x 64 wave tiles, two groups one barrier interval apart; per K = 32 slice and wave:
//               12 ds_read_b128 + 4 LDS-DMA + 32 MFMA, 2 barriers.  LDS reads per slice and CU: 96 KiB.
//   v2 (probe): 4 waves, 128 x 128 wave tiles; per slice and wave 16 ds_read_b128 + 8 LDS-DMA + 64 MFMA, ONE barrier;
//               the fragments of slice s + 1 are read into a second register set under the MFMAs of slice s, the ring
//               streams across tile boundaries (slice s + 4 is issued at the top of slice s: 4 slices of lead instead of 3).
//               LDS reads per slice and CU: 64 KiB (-33 %).
//
// K loop only: the "epilogue" of both is a register sum of the accumulators and one 4-B store per lane and tile (the same
// VALU work per SIMD in both), so the difference is the loop.  -DCHECK builds store the full fp32 tile and compare with a
// host reference (transpose-detecting: random data, M != N).
//
// Build / run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/engine_v2.hip -o /tmp/engine_v2 && /tmp/engine_v2
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "../../sonar_amd/csrc/gemm_tile256.hpp"

using namespace smi;

#define HIP_OK(x)                                                                   \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

// tile i of workgroup b.  raster 2 of the product kernel when the shape allows it (XCD c owns the m-groups c, c + 8, ...),
// plain column-major tile order otherwise (the small CHECK shapes).
__device__ __forceinline__ bool tile_of(int i, int b, int nb, int ntm, int ntn, int& tm, int& tn) {
  if ((ntm % 64) == 0 && (ntn % 4) == 0 && nb == 256) {
    const int nq = ntn / 4, c = b & 7, j = b >> 3, q = i;
    if (q >= (ntm / 64) * nq) return false;
    tm = (c + 8 * (q / nq)) * 8 + j % 8;
    tn = ((q + c) % nq) * 4 + j / 8;
    return true;
  }
  const int t = i * nb + b;
  if (t >= ntm * ntn) return false;
  tm = t % ntm;
  tn = t / ntm;
  return true;
}

// ------------------------------------------------------------------------------------------------ v1: today's loop
template <bool CHECK>
__global__ __launch_bounds__(G2_THREADS) void k_v1(const f16* __restrict__ X, const f16* __restrict__ W,
                                                   float* __restrict__ out, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* zero_lds = (float*)(smem + G2_LDS_BYTES);
  if (threadIdx.x < 256) zero_lds[threadIdx.x] = 0.f;
  const int ntm = M / 256, ntn = N / 256, nt = K / 32;
  int tm, tn, i = 0;
  if (!tile_of(0, blockIdx.x, gridDim.x, ntm, ntn, tm, tn)) return;
  G2Src src = g2_make_src<true, true>(X, W, K, tm * 256, tn * 256, 0);
  g2_prefetch(src, nt, smem);
  bool more = true;
  while (more) {
    const int m0 = tm * 256, n0 = tn * 256;
    GemmTile256Acc acc;
    g2_begin(acc, zero_lds);
    g2_mainloop(acc, src, nt, smem);
    ++i;
    more = tile_of(i, blockIdx.x, gridDim.x, ntm, ntn, tm, tn);
    if (more) {
      src = g2_make_src<true, true>(X, W, K, tm * 256, tn * 256, 0);
      g2_prefetch(src, nt, smem);
    }
    if constexpr (CHECK) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) out[(size_t)g2_row(m0, mi) * N + g2_col(n0, ni) + r] = acc.v[ni][mi][r];
    } else {
      float s = 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) s += (acc.v[ni][mi][0] + acc.v[ni][mi][1]) + (acc.v[ni][mi][2] + acc.v[ni][mi][3]);
      out[(size_t)blockIdx.x * G2_THREADS + threadIdx.x] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------ v2: 4 waves x 128 x 128
#include "engine_v2_asm.inc"
constexpr int V2_THREADS = 256;
constexpr int V2_SLOT = 32768;

struct V2Frag {
  half8 w[8], x[8];
};

// The slice stream of one workgroup: slice s = (tile s / nt, k block s % nt); the issue cursor runs 4 slices ahead of the
// consumer, so the last four steps of a tile issue slices 0..3 of the NEXT tile (the ring never drains at a tile boundary).
// No branches inside a step: the cursor is (this lane's source address in the X block, in the W block) + a byte increment
// (0 once the stream has run out: the last block is fetched again into a slot nobody reads any more).
struct V2Stream {
  const char* xp;
  const char* wp;
  int inc;
};

// Per-wave constants of the ring: LDS-DMA destinations (M0 values: slot base + wave * 4096, W part 16 KiB above) and the
// fragment read addresses of the lane in every slot.
struct V2Ring {
  unsigned m0x[4], m0w[4];
  unsigned xa[4], wa[4];
};

// block J of a step (engine_v2_asm.inc): 16 MFMAs of `cur`, 4 fragment reads of the next slice into `nxt`, 2 LDS-DMA
#define V2_BLOCK_N(NAME, J, cur, nxt, XA, WA, GP, M0V)                                                                      \
  asm volatile(NAME##_STR                                                                                            \
               : [nw0] "=&v"(nxt.w[2 * J]), [nw1] "=&v"(nxt.w[2 * J + 1]), [nx0] "=&v"(nxt.x[2 * J]),                       \
                 [nx1] "=&v"(nxt.x[2 * J + 1])                                                                              \
               : [w0] "v"(cur.w[2 * J]), [w1] "v"(cur.w[2 * J + 1]), [x0] "v"(cur.x[0]), [x1] "v"(cur.x[1]),                \
                 [x2] "v"(cur.x[2]), [x3] "v"(cur.x[3]), [x4] "v"(cur.x[4]), [x5] "v"(cur.x[5]), [x6] "v"(cur.x[6]),        \
                 [x7] "v"(cur.x[7]), [xa] "v"(XA), [wa] "v"(WA), [gp] "v"(GP), [m0v] "s"(M0V)                               \
               : "memory", V2_BLOCK##J##_CLOB)

#define V2_READ(J, nxt, XA, WA)                                                                                             \
  asm volatile(V2_READ##J##_STR                                                                                             \
               : [nw0] "=&v"(nxt.w[2 * J]), [nw1] "=&v"(nxt.w[2 * J + 1]), [nx0] "=&v"(nxt.x[2 * J]),                       \
                 [nx1] "=&v"(nxt.x[2 * J + 1])                                                                              \
               : [xa] "v"(XA), [wa] "v"(WA)                                                                                 \
               : "memory")

// LDS-DMA of a whole slice (pipeline fill only; in the loop the 8 instructions ride inside the blocks)
__device__ __forceinline__ void v2_fill(V2Stream& st, unsigned m0x, unsigned m0w) {
  asm volatile(
      "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
      "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072"
      :
      : "v"(st.xp), "v"(st.wp), "s"(m0x), "s"(m0w)
      : "memory");
  st.xp += st.inc;
  st.wp += st.inc;
}

// one slice: consume `cur` (slice s, ring slot SLOT), read slice s + 1 (slot SLOT + 1) into `nxt`, issue slice s + 4 into
// slot SLOT.  Entry: this wave's part of slice s + 1 may still be in flight.
// VAR 0: the full step.  1: 10 of 16 MFMAs per block (the MFMA count of a 160-row workgroup tile).  2: that + 6 instead of 8
// DMA instructions per wave and slice (24 KiB slices) -- lone-tile probes, wrong results on purpose.
template <int SLOT, int VAR>
__device__ __forceinline__ void v2_step(const V2Frag& cur, V2Frag& nxt, V2Stream& st, const V2Ring& rg) {
  // my pieces of slice s + 1 have landed (s + 2, s + 3 stay in flight); everybody's have, and everybody has finished reading
  // slice s (slot SLOT is free) once the barrier is passed
  if constexpr (VAR == 2)
    asm volatile("s_waitcnt vmcnt(12)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(16)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  constexpr int NS = (SLOT + 1) & 3;
  if constexpr (VAR == 0) {
    V2_BLOCK_N(V2_BLOCK0, 0, cur, nxt, rg.xa[NS], rg.wa[NS], st.xp, rg.m0x[SLOT]);
    V2_BLOCK_N(V2_BLOCK1, 1, cur, nxt, rg.xa[NS], rg.wa[NS], st.xp, rg.m0x[SLOT]);
    V2_BLOCK_N(V2_BLOCK2, 2, cur, nxt, rg.xa[NS], rg.wa[NS], st.wp, rg.m0w[SLOT]);
    V2_BLOCK_N(V2_BLOCK3, 3, cur, nxt, rg.xa[NS], rg.wa[NS], st.wp, rg.m0w[SLOT]);
  } else if constexpr (VAR == 1) {
    V2_BLOCK_N(V2_BLOCK0R, 0, cur, nxt, rg.xa[NS], rg.wa[NS], st.xp, rg.m0x[SLOT]);
    V2_BLOCK_N(V2_BLOCK1R, 1, cur, nxt, rg.xa[NS], rg.wa[NS], st.xp, rg.m0x[SLOT]);
    V2_BLOCK_N(V2_BLOCK2R, 2, cur, nxt, rg.xa[NS], rg.wa[NS], st.wp, rg.m0w[SLOT]);
    V2_BLOCK_N(V2_BLOCK3R, 3, cur, nxt, rg.xa[NS], rg.wa[NS], st.wp, rg.m0w[SLOT]);
  } else {
    V2_BLOCK_N(V2_BLOCK0RD, 0, cur, nxt, rg.xa[NS], rg.wa[NS], st.xp, rg.m0x[SLOT]);
    V2_BLOCK_N(V2_BLOCK1RD, 1, cur, nxt, rg.xa[NS], rg.wa[NS], st.xp, rg.m0x[SLOT]);
    V2_BLOCK_N(V2_BLOCK2RD, 2, cur, nxt, rg.xa[NS], rg.wa[NS], st.wp, rg.m0w[SLOT]);
    V2_BLOCK_N(V2_BLOCK3RD, 3, cur, nxt, rg.xa[NS], rg.wa[NS], st.wp, rg.m0w[SLOT]);
  }
  st.xp += st.inc;
  st.wp += st.inc;
}

__device__ __forceinline__ void v2_init_acc(float v) {
  asm volatile(V2_INIT0_STR ::"v"(v) : V2_BLOCK0_CLOB);
  asm volatile(V2_INIT1_STR ::"v"(v) : V2_BLOCK1_CLOB);
  asm volatile(V2_INIT2_STR ::"v"(v) : V2_BLOCK2_CLOB);
  asm volatile(V2_INIT3_STR ::"v"(v) : V2_BLOCK3_CLOB);
}

template <bool CHECK, int VAR = 0>
__global__ __launch_bounds__(V2_THREADS) void k_v2(const f16* __restrict__ X, const f16* __restrict__ W,
                                                   float* __restrict__ out, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, kg = lane >> 4;
  const int t_sw = (kg ^ tm_swz(l15)) << 4;
  const unsigned lds0 = (unsigned)(size_t)smem;
  V2Ring rg;
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    rg.m0x[sl] = __builtin_amdgcn_readfirstlane(lds0 + sl * V2_SLOT + wave * 4096);
    rg.m0w[sl] = rg.m0x[sl] + 16384;
    rg.xa[sl] = lds0 + sl * V2_SLOT + (wr * 128 + l15) * 64 + t_sw;
    rg.wa[sl] = lds0 + sl * V2_SLOT + 16384 + (wc * 128 + l15) * 64 + t_sw;
  }
  const unsigned voff = wave * 4096 + lane * 16;

  const int ntm = M / 256, ntn = N / 256, nt = K / 32;
  int tm, tn;
  if (!tile_of(0, blockIdx.x, gridDim.x, ntm, ntn, tm, tn)) return;
  V2Stream st;
  st.xp = (const char*)X + (size_t)tm * nt * (TM_BLOCK * 2) + voff;
  st.wp = (const char*)W + (size_t)tn * nt * (TM_BLOCK * 2) + voff;
  st.inc = TM_BLOCK * 2;

  // fill: slices 0..3 of the first tile (nt >= 8)
  v2_fill(st, rg.m0x[0], rg.m0w[0]);
  v2_fill(st, rg.m0x[1], rg.m0w[1]);
  v2_fill(st, rg.m0x[2], rg.m0w[2]);
  v2_fill(st, rg.m0x[3], rg.m0w[3]);
  asm volatile("s_waitcnt vmcnt(24)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  V2Frag fa, fb;
  V2_READ(0, fa, rg.xa[0], rg.wa[0]);
  V2_READ(1, fa, rg.xa[0], rg.wa[0]);
  V2_READ(2, fa, rg.xa[0], rg.wa[0]);
  V2_READ(3, fa, rg.xa[0], rg.wa[0]);

  bool more = true;
  for (int i = 0; more; ++i) {
    const int m0 = tm * 256, n0 = tn * 256;
    v2_init_acc(0.f);
    for (int kb = 0; kb < nt - 4; kb += 4) {
      v2_step<0, VAR>(fa, fb, st, rg);
      v2_step<1, VAR>(fb, fa, st, rg);
      v2_step<2, VAR>(fa, fb, st, rg);
      v2_step<3, VAR>(fb, fa, st, rg);
    }
    // the last four slices of this tile: the cursor moves to the next tile's first block
    more = tile_of(i + 1, blockIdx.x, gridDim.x, ntm, ntn, tm, tn);
    if (more) {
      st.xp = (const char*)X + (size_t)tm * nt * (TM_BLOCK * 2) + voff;
      st.wp = (const char*)W + (size_t)tn * nt * (TM_BLOCK * 2) + voff;
    } else {
      st.xp -= st.inc;
      st.wp -= st.inc;
      st.inc = 0;
    }
    v2_step<0, VAR>(fa, fb, st, rg);
    v2_step<1, VAR>(fb, fa, st, rg);
    v2_step<2, VAR>(fa, fb, st, rg);
    v2_step<3, VAR>(fb, fa, st, rg);
    if constexpr (CHECK) {
      // acc[ni][mi][r] = a[(ni*8+mi)*4 + r] = C[m0 + wr*128 + mi*16 + l15][n0 + wc*128 + ni*16 + 4*kg + r]
      float* o = out + (size_t)(m0 + wr * 128 + l15) * N + n0 + wc * 128 + 4 * kg;
#define V2_OUT(k)                                                                                    \
  {                                                                                                  \
    f32x4 v;                                                                                         \
    asm volatile(V2_RDOUT##k##_STR : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));                \
    *(f32x4*)(o + (size_t)((k) % 8) * 16 * N + ((k) / 8) * 16) = v;                                  \
  }
#define V2_OUT8(b) V2_OUT(b##0) V2_OUT(b##1) V2_OUT(b##2) V2_OUT(b##3) V2_OUT(b##4) V2_OUT(b##5) V2_OUT(b##6) V2_OUT(b##7)
      V2_OUT(0) V2_OUT(1) V2_OUT(2) V2_OUT(3) V2_OUT(4) V2_OUT(5) V2_OUT(6) V2_OUT(7) V2_OUT(8) V2_OUT(9)
      V2_OUT(10) V2_OUT(11) V2_OUT(12) V2_OUT(13) V2_OUT(14) V2_OUT(15) V2_OUT(16) V2_OUT(17) V2_OUT(18) V2_OUT(19)
      V2_OUT(20) V2_OUT(21) V2_OUT(22) V2_OUT(23) V2_OUT(24) V2_OUT(25) V2_OUT(26) V2_OUT(27) V2_OUT(28) V2_OUT(29)
      V2_OUT(30) V2_OUT(31) V2_OUT(32) V2_OUT(33) V2_OUT(34) V2_OUT(35) V2_OUT(36) V2_OUT(37) V2_OUT(38) V2_OUT(39)
      V2_OUT(40) V2_OUT(41) V2_OUT(42) V2_OUT(43) V2_OUT(44) V2_OUT(45) V2_OUT(46) V2_OUT(47) V2_OUT(48) V2_OUT(49)
      V2_OUT(50) V2_OUT(51) V2_OUT(52) V2_OUT(53) V2_OUT(54) V2_OUT(55) V2_OUT(56) V2_OUT(57) V2_OUT(58) V2_OUT(59)
      V2_OUT(60) V2_OUT(61) V2_OUT(62) V2_OUT(63)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------ host
static void fill(std::vector<f16>& v, unsigned seed, bool zeros) {
  unsigned s = seed;
  for (auto& e : v) {
    s = s * 1664525u + 1013904223u;
    e = zeros ? (f16)0.f : (f16)(((s >> 8) & 0xffff) / 32768.f - 1.f);
  }
}

template <typename F>
static float time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  HIP_OK(hipEventRecord(e1));
  HIP_OK(hipEventSynchronize(e1));
  float ms;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  HIP_OK(hipEventDestroy(e0));
  HIP_OK(hipEventDestroy(e1));
  return ms / reps;
}

static int check_shape(int M, int N, int K, int reps = 1) {
  std::vector<f16> hx((size_t)M * K), hw((size_t)N * K);
  fill(hx, 1, false);
  fill(hw, 2, false);
  f16 *dx, *dw;
  float* dout;
  HIP_OK(hipMalloc(&dx, hx.size() * 2));
  HIP_OK(hipMalloc(&dw, hw.size() * 2));
  HIP_OK(hipMalloc(&dout, (size_t)M * N * 4));
  HIP_OK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  // host reference on a sample of rows / all columns, operands interpreted tile-major
  std::vector<float> ref((size_t)M * N, 0.f);
  std::vector<float> xr((size_t)M * K), wr((size_t)N * K);
  for (int r = 0; r < M; ++r)
    for (int k = 0; k < K; ++k) xr[(size_t)r * K + k] = (float)hx[tm_offset(r, k, K)];
  for (int r = 0; r < N; ++r)
    for (int k = 0; k < K; ++k) wr[(size_t)r * K + k] = (float)hw[tm_offset(r, k, K)];
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double a = 0;
      const float* xp = &xr[(size_t)m * K];
      const float* wp = &wr[(size_t)n * K];
      for (int k = 0; k < K; ++k) a += (double)xp[k] * wp[k];
      ref[(size_t)m * N + n] = (float)a;
    }
  int bad_total = 0;
  for (int rw = 0; rw < 2 * reps; ++rw) {
    const int which = rw & 1;
    HIP_OK(hipMemset(dout, 0xff, (size_t)M * N * 4));
    const int grid = std::min(256, (M / 256) * (N / 256));
    if (which == 0) {
      HIP_OK(hipFuncSetAttribute((const void*)k_v1<true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_KERNEL_LDS_BYTES));
      hipLaunchKernelGGL((k_v1<true>), dim3(grid), dim3(G2_THREADS), G2_KERNEL_LDS_BYTES, 0, dx, dw, dout, M, N, K);
    } else {
      HIP_OK(hipFuncSetAttribute((const void*)k_v2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * V2_SLOT));
      hipLaunchKernelGGL((k_v2<true>), dim3(grid), dim3(V2_THREADS), 4 * V2_SLOT, 0, dx, dw, dout, M, N, K);
    }
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> got((size_t)M * N);
    HIP_OK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    int bad = 0;
    for (size_t i = 0; i < got.size(); ++i) {
      const double e = fabs((double)got[i] - ref[i]);
      if (!(e <= 2e-2 + 1e-3 * fabs(ref[i]))) ++bad;
      if (e > maxerr || e != e) maxerr = e;
    }
    printf("check %s M=%d N=%d K=%d: max |err| %.3e, bad %d of %zu\n", which ? "v2" : "v1", M, N, K, maxerr, bad, got.size());
    bad_total += bad;
  }
  HIP_OK(hipFree(dx));
  HIP_OK(hipFree(dw));
  HIP_OK(hipFree(dout));
  return bad_total;
}

int main(int argc, char** argv) {
  int bad = 0;
  // CHECK: several tiles per workgroup (stream across tile boundaries), M != N
  bad += check_shape(512, 256, 128);
  bad += check_shape(768, 512, 1024);
  bad += check_shape(256 * 40, 256 * 8, 128, 4);  // 320 tiles on 256 workgroups, repeated: race screen
  if (bad) {
    printf("CHECK FAILED\n");
    return 1;
  }
  if (argc > 1 && !strcmp(argv[1], "lone")) {
    // LONE TILES: M = 1280 rows (the decode step: 256 sentences x beam 5) x N 8192 x K 1024 = 160 tiles on 256 CUs, one per
    // workgroup; the weights rotate over NW matrices (NW x 16.8 MB > the Infinity Cache) so that every launch streams them
    // from HBM as a decode step does.  What bounds a lone tile: its MFMA count, its operand bytes, or the latency of a slice?
    const int M = 1280, N = 8192, K = 1024, NW = 24;
    std::vector<f16> hx((size_t)M * K), hw((size_t)N * K);
    fill(hx, 21, false);
    fill(hw, 22, false);
    f16 *dx, *dw;
    float* dout;
    HIP_OK(hipMalloc(&dx, hx.size() * 2));
    HIP_OK(hipMalloc(&dw, hw.size() * 2 * NW));
    HIP_OK(hipMalloc(&dout, 256 * 512 * 4));
    HIP_OK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < NW; ++i) HIP_OK(hipMemcpy(dw + (size_t)i * N * K, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipFuncSetAttribute((const void*)k_v1<false>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_KERNEL_LDS_BYTES));
    HIP_OK(hipFuncSetAttribute((const void*)k_v2<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * V2_SLOT));
    HIP_OK(hipFuncSetAttribute((const void*)k_v2<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * V2_SLOT));
    HIP_OK(hipFuncSetAttribute((const void*)k_v2<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * V2_SLOT));
    for (int hot = 0; hot < 2; ++hot) {
      int wi = 0;
      auto wsel = [&] { f16* p = dw + (size_t)(hot ? 0 : (wi++ % NW)) * N * K; return p; };
      auto l1 = [&] { hipLaunchKernelGGL((k_v1<false>), dim3(160), dim3(G2_THREADS), G2_KERNEL_LDS_BYTES, 0, dx, wsel(), dout, M, N, K); };
      auto l20 = [&] { hipLaunchKernelGGL((k_v2<false, 0>), dim3(160), dim3(V2_THREADS), 4 * V2_SLOT, 0, dx, wsel(), dout, M, N, K); };
      auto l21 = [&] { hipLaunchKernelGGL((k_v2<false, 1>), dim3(160), dim3(V2_THREADS), 4 * V2_SLOT, 0, dx, wsel(), dout, M, N, K); };
      auto l22 = [&] { hipLaunchKernelGGL((k_v2<false, 2>), dim3(160), dim3(V2_THREADS), 4 * V2_SLOT, 0, dx, wsel(), dout, M, N, K); };
      std::vector<float> t[4];
      for (int r = 0; r < 6; ++r) {
        t[0].push_back(time_ms(l1, 48));
        t[1].push_back(time_ms(l20, 48));
        t[2].push_back(time_ms(l21, 48));
        t[3].push_back(time_ms(l22, 48));
      }
      const char* names[4] = {"8-wave ping-pong, 256x256 unit", "4-wave, 256x256 unit (64 MFMA / step)", "4-wave, 40 MFMA / step (160-row MFMA count)",
                              "4-wave, 40 MFMA / step + 24 KiB slices"};
      for (int v = 0; v < 4; ++v) {
        std::sort(t[v].begin(), t[v].end());
        printf("lone tiles, %s weights: %-46s %.2f us per launch (min %.2f), back-to-back launches incl. launch overhead\n",
               hot ? "L2-hot " : "HBM-cold", names[v], t[v][t[v].size() / 2] * 1e3, t[v][0] * 1e3);
      }
    }
    return 0;
  }
  const int M = argc > 1 ? atoi(argv[1]) : 131072, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 1024;
  const int rounds = argc > 4 ? atoi(argv[4]) : 7;
  std::vector<f16> hx((size_t)M * K), hw((size_t)N * K);
  f16 *dx, *dw;
  float* dout;
  HIP_OK(hipMalloc(&dx, hx.size() * 2));
  HIP_OK(hipMalloc(&dw, hw.size() * 2));
  HIP_OK(hipMalloc(&dout, 256 * 512 * 4));
  HIP_OK(hipFuncSetAttribute((const void*)k_v1<false>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_KERNEL_LDS_BYTES));
  HIP_OK(hipFuncSetAttribute((const void*)k_v2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * V2_SLOT));
  const double flop = 2.0 * M * N * K;
  for (int zeros = 0; zeros < 2; ++zeros) {
    fill(hx, 11, zeros);
    fill(hw, 12, zeros);
    HIP_OK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    auto l1 = [&] { hipLaunchKernelGGL((k_v1<false>), dim3(256), dim3(G2_THREADS), G2_KERNEL_LDS_BYTES, 0, dx, dw, dout, M, N, K); };
    auto l2 = [&] { hipLaunchKernelGGL((k_v2<false>), dim3(256), dim3(V2_THREADS), 4 * V2_SLOT, 0, dx, dw, dout, M, N, K); };
    time_ms(l1, 3);
    time_ms(l2, 3);
    std::vector<float> t1, t2;
    for (int r = 0; r < rounds; ++r) {
      t1.push_back(time_ms(l1, 10));
      t2.push_back(time_ms(l2, 10));
    }
    std::sort(t1.begin(), t1.end());
    std::sort(t2.begin(), t2.end());
    const float m1 = t1[t1.size() / 2], m2 = t2[t2.size() / 2];
    printf("%s M=%d N=%d K=%d  v1 (8-wave ping-pong): median %.4f ms (min %.4f) = %.0f TFLOP/s | v2 (4 waves 128x128): median %.4f ms "
           "(min %.4f) = %.0f TFLOP/s | v2/v1 = %+.1f %%\n",
           zeros ? "zeros " : "random", M, N, K, m1, t1[0], flop / m1 * 1e-9, m2, t2[0], flop / m2 * 1e-9, (m1 / m2 - 1) * 100);
  }
  return 0;
}
