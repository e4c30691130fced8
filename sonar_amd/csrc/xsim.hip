// xsim cosine-similarity mining: for every row of X find the k most similar
// rows of Y (cosine = dot product of L2-normalised rows).  The reference only
// names xsim (README.md:5); its in-tree form is F.normalize(x) @ F.normalize(y).T
// (tests/integration_tests/test_text_sonar.py:42-53).  Here the Nx x Ny score
// matrix is never materialised: each 128x128 score tile comes out of the
// shared MFMA tile engine straight into a per-lane running top-k kept in
// VGPRs, so HBM traffic stays ~(Nx+Ny)*d*2 B and the kernel is MFMA-bound
// (2*d flop per pair).
//
// Work split: grid = (x tiles) x (y chunks); a workgroup walks the y tiles of
// its chunk.  The grouped XCD raster makes the ~64 workgroups resident on one
// XCD an 8(x) x 8(chunk) block: its 8 X panels stay L2-resident for the whole
// walk and every streamed Y tile is shared by 8 workgroups.
#include <algorithm>
#include <cstdlib>

#include "gemm_tile.hpp"
#include "gemm_tile256.hpp"
#include "kernels.hpp"

namespace smi {

// total order: higher score first, ties -> lower index first (deterministic
// regardless of the order candidates are met).
__device__ __forceinline__ bool better(float s, int i, float s2, int i2) {
  return s > s2 || (s == s2 && i < i2);
}

template <int K>
struct TopK {
  float s[K];
  int i[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      s[j] = -INFINITY;
      i[j] = 0x7fffffff;
    }
  }
  __device__ __forceinline__ void push(float v, int idx) {
    if (better(v, idx, s[K - 1], i[K - 1])) {
      s[K - 1] = v;
      i[K - 1] = idx;
#pragma unroll
      for (int j = K - 1; j > 0; --j) {
        if (better(s[j], i[j], s[j - 1], i[j - 1])) {
          const float ts = s[j];
          s[j] = s[j - 1];
          s[j - 1] = ts;
          const int ti = i[j];
          i[j] = i[j - 1];
          i[j - 1] = ti;
        }
      }
    }
  }
};

// partial results: ps/pi [nchunks][nx_pad][K]
template <int K>
__global__ __launch_bounds__(GT_THREADS, 2) void xsim_tile_kernel(const f16* __restrict__ Xn,
                                                                  const f16* __restrict__ Yn,
                                                                  int d, int ntx, int nty,
                                                                  int nchunks, int tiles_per_chunk,
                                                                  int ny, int nx_pad,
                                                                  float* __restrict__ ps,
                                                                  int* __restrict__ pi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile_m, chunk;
  gt_tile_coords(ntx, nchunks, tile_m, chunk);
  const int m0 = tile_m * GT_BM;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;

  TopK<K> best[2];
  best[0].init();
  best[1].init();

  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = min(nty, t_begin + tiles_per_chunk);
  for (int ty = t_begin; ty < t_end; ++ty) {
    const int n0 = ty * GT_BN;
    GemmTileAcc acc;
    gt_mainloop(acc, Xn, Yn, d, m0, n0, smem);
    __syncthreads();  // all waves done with LDS before the next tile's DMA
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        float vmax = acc.v[ni][mi][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) vmax = fmaxf(vmax, acc.v[ni][mi][r]);
        if (vmax >= best[mi].s[K - 1]) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 64 + ni * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (n < ny) best[mi].push(acc.v[ni][mi][r], n);
          }
        }
      }
    }
  }

  // merge the 4 lists of every x row (2 lane halves x 2 n-waves) through LDS
  float* ls = (float*)smem;                        // [128 rows][4][K]
  int* li = (int*)(smem + 128 * 4 * K * 4);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int row = wm * 64 + mi * 32 + l31;
    const int slot = wn * 2 + hi;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ls[(row * 4 + slot) * K + j] = best[mi].s[j];
      li[(row * 4 + slot) * K + j] = best[mi].i[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int row = threadIdx.x;
    TopK<K> t;
    t.init();
    for (int c = 0; c < 4 * K; ++c) t.push(ls[row * 4 * K + c], li[row * 4 * K + c]);
    const size_t o = ((size_t)chunk * nx_pad + m0 + row) * K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ps[o + j] = t.s[j];
      pi[o + j] = t.i[j];
    }
  }
}

// 256-row variant on the ping-pong tile engine (gemm_tile256.hpp).  The K slices of
// ALL y tiles of the chunk form one continuous DMA stream through the 4-slot LDS ring:
// the pipeline is filled once per workgroup, not once per y tile; at every tile
// boundary the accumulators are folded into the running top-k and cleared.
#ifdef SMI_XSIM_TRACE
// development aid (-DSMI_XSIM_TRACE, tools/xsim_trace.py): per workgroup and ping-pong group, the 100 MHz wall-clock ticks of the
// whole slice stream, of the folds inside it and the number of folds (lane 0 of waves 0 and 4 of the first 256 workgroups)
__device__ unsigned long long xs_trace_buf[256 * 2 * 4];
#endif

// The fold of a finished y tile (n0f = its first y row), as a macro: top-1 calls it through a lambda from two places (below),
// and a lambda around it costs top-4, which has no register to spare, 76 bytes of spills.
#define XS_FOLD(n0f)                                                                                                         \
  do {                                                                                                                       \
    float rowthr[8];                                                                                                         \
    _Pragma("unroll")                                                                                                        \
    for (int mi = 0; mi < 8; ++mi) {                                                                                         \
      float my = best[mi].s[KR - 1];                                                                                          \
      my = fmaxf(my, __shfl_xor(my, 16, 64));                                                                                \
      my = fmaxf(my, __shfl_xor(my, 32, 64));                                                                                \
      if (kg == 0) thr[(wr * 128 + mi * 16 + l15) * 4 + wc] = my;                                                            \
    }                                                                                                                        \
    _Pragma("unroll")                                                                                                        \
    for (int mi = 0; mi < 8; ++mi) {                                                                                         \
      const f32x4 t4 = *(const f32x4*)(thr + (wr * 128 + mi * 16 + l15) * 4);                                                \
      rowthr[mi] = fmaxf(fmaxf(t4[0], t4[1]), fmaxf(t4[2], t4[3]));                                                          \
    }                                                                                                                        \
    _Pragma("unroll")                                                                                                        \
    for (int mi = 0; mi < 8; ++mi) {                                                                                         \
      float vmax = -INFINITY;                                                                                                \
      _Pragma("unroll")                                                                                                      \
      for (int ni = 0; ni < 4; ++ni)                                                                                         \
        _Pragma("unroll")                                                                                                    \
        for (int r = 0; r < 4; ++r) vmax = fmaxf(vmax, acc.v[ni][mi][r]);                                                    \
      if (vmax >= rowthr[mi]) {                                                                                              \
        _Pragma("unroll")                                                                                                    \
        for (int ni = 0; ni < 4; ++ni)                                                                                       \
          _Pragma("unroll")                                                                                                  \
          for (int r = 0; r < 4; ++r) {                                                                                      \
            const int n = n0f + wc * 64 + ni * 16 + 4 * kg + r;                                                              \
            if (n < ny) best[mi].push(acc.v[ni][mi][r], n);                                                                  \
          }                                                                                                                  \
      }                                                                                                                      \
    }                                                                                                                        \
  } while (0)

// ---- running top-k lists in LDS (round 4) ----------------------------------------------------------------------------
// One list per X ROW of the tile, shared by the 16 lanes (4 lane groups x 4 column waves) that see scores of that row:
// K 64-bit keys, descending, key = order-preserving(score) << 32 | (0xffffffff - y index) -- the total order of `better`
// (score desc, index asc) as ONE unsigned compare.  A candidate is inserted with a cascade of LDS `max` atomics
// (slot j keeps the larger of {slot, carried key}, the smaller travels on): slot 0 sees every inserted key, so it ends
// as their maximum; slot 1 sees every key but that one; ... -- the final list is the top K whatever the interleaving
// of the 16 lanes, and identical from run to run.  What this buys over per-lane lists in registers (8 lists x K x 2
// words per lane: 16 VGPRs at top-1, 64 at top-4, where hipcc spilled and the deferred fold could not be used):
//  * no list registers at all: every K runs the schedule of top-1 (group 0 defers its fold into group 1's interval);
//  * the threshold of a row is the row's TRUE K-th best (one ds_read_b32 of the last key's score word), not the
//    maximum of four waves' private K-th bests: no shuffles, no publish step, fewer false passes;
//  * the walk ends with the lists final: no cross-lane / cross-wave merge.
// Insertions are rare after the first tiles (~K ln(n / K) per row over a chunk of n rows); they take the slow path.
constexpr unsigned long long XS_EMPTY = (0x007fffffull << 32) | 0x80000000ull;  // (-inf, index 0x7fffffff)
__device__ __forceinline__ uint32_t xs_ord(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float xs_unord(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
template <int K>
__device__ __forceinline__ void xs_insert(unsigned long long* list, float v, int n) {
  unsigned long long key = ((unsigned long long)xs_ord(v) << 32) | (uint32_t)(0xffffffffu - (uint32_t)n);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const unsigned long long old = __hip_atomic_fetch_max(list + j, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    key = old < key ? old : key;
    if (key == XS_EMPTY) break;
  }
}
// the 8 row thresholds of a lane (rows wr * 128 + mi * 16 + l15): score word of the last key of each list.  Inline asm, issue
// and wait in ONE statement: as C++ loads every one of them would get `s_waitcnt vmcnt(0)` from hipcc's LDS-DMA alias tracking
// (the operand slices of the next tile are in flight) -- gemm.hip, round-3 findings.  A stale value is a valid lower bound.
template <int K>
__device__ __forceinline__ void xs_thresholds(unsigned lds_addr, float (&t)[8]) {
  uint32_t o[8];
  asm volatile(
      "ds_read_b32 %0, %8 offset:%9\n\tds_read_b32 %1, %8 offset:%10\n\tds_read_b32 %2, %8 offset:%11\n\t"
      "ds_read_b32 %3, %8 offset:%12\n\tds_read_b32 %4, %8 offset:%13\n\tds_read_b32 %5, %8 offset:%14\n\t"
      "ds_read_b32 %6, %8 offset:%15\n\tds_read_b32 %7, %8 offset:%16\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(lds_addr), "n"(0 * 128 * K), "n"(1 * 128 * K), "n"(2 * 128 * K), "n"(3 * 128 * K), "n"(4 * 128 * K),
        "n"(5 * 128 * K), "n"(6 * 128 * K), "n"(7 * 128 * K)
      : "memory");
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) t[mi] = xs_unord(o[mi]);
}

#define XS_FOLD_LL(n0f)                                                                                                      \
  do {                                                                                                                       \
    float rowthr[8], vmx[8];                                                                                                 \
    xs_thresholds<K>(thr_addr, rowthr);                                                                                      \
    bool hit = false;                                                                                                        \
    _Pragma("unroll")                                                                                                        \
    for (int mi = 0; mi < 8; ++mi) {                                                                                         \
      float vmax = -INFINITY;                                                                                                \
      _Pragma("unroll")                                                                                                      \
      for (int ni = 0; ni < 4; ++ni)                                                                                         \
        _Pragma("unroll")                                                                                                    \
        for (int r = 0; r < 4; ++r) vmax = fmaxf(vmax, acc.v[ni][mi][r]);                                                    \
      vmx[mi] = vmax;                                                                                                        \
      hit |= vmax >= rowthr[mi];                                                                                             \
    }                                                                                                                        \
    if (__any(hit)) {                                                                                                        \
      _Pragma("unroll")                                                                                                      \
      for (int mi = 0; mi < 8; ++mi) {                                                                                       \
        if (vmx[mi] >= rowthr[mi]) {                                                                                         \
          unsigned long long* lst = lists + (wr * 128 + mi * 16 + l15) * K;                                                  \
          _Pragma("unroll")                                                                                                  \
          for (int ni = 0; ni < 4; ++ni)                                                                                     \
            _Pragma("unroll")                                                                                                \
            for (int r = 0; r < 4; ++r) {                                                                                    \
              const int n = n0f + wc * 64 + ni * 16 + 4 * kg + r;                                                            \
              const float v = acc.v[ni][mi][r];                                                                              \
              if (v >= rowthr[mi] && n < ny) xs_insert<K>(lst, v, n);                                                        \
            }                                                                                                                \
        }                                                                                                                    \
      }                                                                                                                      \
    }                                                                                                                        \
  } while (0)

// TM: Xn / Yn are TILE-MAJOR copies (common.hpp; packed into the workspace by xsim_run): a K slice of an operand is one
// contiguous 16 KiB block and the Y stream of a chunk one linear walk, instead of 256 pieces of 64 B a row apart -- the
// layout that bought the GEMMs +7 % end to end and +26 % on the bare operand stream.
// LL: the running top-k lists live in LDS (above), one per row; !LL: per-lane lists in registers (rounds 1-3, K <= 4)
template <int K, bool TM, bool LL>
__global__ __launch_bounds__(G2_THREADS) void xsim_tile256_kernel(const f16* __restrict__ Xn,
                                                                  const f16* __restrict__ Yn, int d,
                                                                  int ntx, int nty, int nchunks,
                                                                  int tiles_per_chunk, int ny,
                                                                  int nx_pad, float* __restrict__ ps,
                                                                  int* __restrict__ pi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile_m, chunk;
  g2_tile_coords(ntx, nchunks, tile_m, chunk);
  const int m0 = tile_m * G2_BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 16x16x32 MFMA blocks (gemm_tile256.hpp): lane -> row l15 of a 16-row block, 4 consecutive
  // columns at 4*kg of a 16-column block
  const int l15 = lane & 15, kg = lane >> 4, wr = wave >> 2, wc = wave & 3;

  const int t_begin = chunk * tiles_per_chunk;
  const int ntiles = min(nty, t_begin + tiles_per_chunk) - t_begin;
  const int nt = d / G2_BK;
  const int S = max(ntiles, 0) * nt;

  constexpr int KR = LL ? 1 : K;  // register lists exist only without LL
  TopK<KR> best[8];               // one running list per accumulator row block
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) best[mi].init();
  unsigned long long* lists = (unsigned long long*)(smem + G2_LDS_BYTES);  // LL: [256 rows][K] keys
  const unsigned thr_addr = (unsigned)(size_t)(smem + G2_LDS_BYTES) + ((wr * 128 + l15) * K + (K - 1)) * 8 + 4;
  if constexpr (LL) {
    if (tid < 256) {
#pragma unroll
      for (int j = 0; j < K; ++j) lists[tid * K + j] = XS_EMPTY;
    }
    __syncthreads();  // (before any LDS-DMA is in flight)
  }
  // Row thresholds shared by the 4 column waves (above the ring): thr[row][wc] = the K-th best score wave wc holds for
  // that row, a lower bound of the row's K-th best.  Published and read WITHOUT a barrier: the values only grow, so a
  // stale one is still a valid bound.  A lane's own K-th best is a much weaker test -- with it some lane of nearly
  // every 16-row block passes and the whole wave walks the insertion path every tile (top-4: 610 ms against 480 for
  // top-1 at 262 144 x 1 M, r02 experiment 24).
  float* thr = (float*)(smem + G2_LDS_BYTES);
  if constexpr (!LL) {
    if (kg == 0) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) thr[(wr * 128 + mi * 16 + l15) * 4 + wc] = -INFINITY;
    }
  }

  if (S > 0) {
    // DMA stream state: slice s = (tile s / nt, k = s % nt)
    const f16* xg[2];
    const f16* yg[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if constexpr (TM) {  // piece (wave * 2 + q) of a 16 KiB block: 16 rows x 64 B = 1 KiB, linear
        xg[q] = Xn + (size_t)tile_m * nt * TM_BLOCK + (wave * 2 + q) * 512 + lane * 8;
        yg[q] = Yn + (size_t)t_begin * nt * TM_BLOCK + (wave * 2 + q) * 512 + lane * 8;
      } else {
        const int row = (wave * 2 + q) * 16 + (lane >> 2);
        const int chunk16 = (lane & 3) ^ tm_swz(row);
        xg[q] = Xn + (size_t)(m0 + row) * d + chunk16 * 8;
        yg[q] = Yn + ((size_t)t_begin * G2_BN + row) * d + chunk16 * 8;
      }
    }
    int ik = 0, is = 0;
    auto issue = [&]() {
      char* slot = smem + (is & 3) * G2_SLOT_BYTES + wave * 2048;
      const size_t koff = TM ? (size_t)ik * TM_BLOCK : (size_t)ik * G2_BK;
      glds16(xg[0] + koff, slot);
      glds16(xg[1] + koff, slot + 1024);
      if constexpr (TM) {  // block (tile, k) of Y sits at (tile * nt + k) * TM_BLOCK: the chunk's slices are consecutive
        glds16(yg[0], slot + G2_BM * G2_BK * 2);
        glds16(yg[1], slot + G2_BM * G2_BK * 2 + 1024);
        yg[0] += TM_BLOCK;
        yg[1] += TM_BLOCK;
        ++is;
        if (++ik == nt) ik = 0;
      } else {
        glds16(yg[0] + koff, slot + G2_BM * G2_BK * 2);
        glds16(yg[1] + koff, slot + G2_BM * G2_BK * 2 + 1024);
        ++is;
        if (++ik == nt) {
          ik = 0;
          yg[0] += (size_t)G2_BN * d;
          yg[1] += (size_t)G2_BN * d;
        }
      }
    };
    const int t_sw = (kg ^ tm_swz(l15)) << 4;
    const int xoff = (wr * 128 + l15) * 64 + t_sw;
    const int woff = G2_BM * G2_BK * 2 + (wc * 64 + l15) * 64 + t_sw;

    GemmTile256Acc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc.v[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue();
    if (S > 1) issue();
    if (S > 2) issue();
    if (S > 2) SMI_WAIT_VMCNT(8);
    else if (S > 1) SMI_WAIT_VMCNT(4);
    else SMI_WAIT_VMCNT(0);
    SMI_BARRIER();
    if (wr == 1) SMI_BARRIER();

    int k = 0, n0 = t_begin * G2_BN;
#ifdef SMI_XSIM_TRACE
    unsigned long long tr_t0 = wall_clock64(), tr_fold = 0, tr_n = 0, tr_max = 0;
#endif
    // y tile finished: fold the 128x64 scores of this wave into the top-k (n0f = first y row of that tile).
    // (v_permlane16/32_swap joins instead of the two ds_bpermute shuffles measured SLOWER twice: r03 experiments 10 and 14)
    auto fold = [&](int n0f) {
#ifdef SMI_XSIM_TRACE
      const unsigned long long tr_f0 = wall_clock64();
#endif
      if constexpr (LL) XS_FOLD_LL(n0f);
      else XS_FOLD(n0f);
#ifdef SMI_XSIM_TRACE
      const unsigned long long dt = wall_clock64() - tr_f0;
      tr_fold += dt;
      tr_max = dt > tr_max ? dt : tr_max;
      ++tr_n;
#endif
    };
    // A fold makes its interval ~5x longer, and the OTHER group waits for it at the barrier.  Left where each group
    // finishes its tile (after its last MFMA segment) the two groups fold in DIFFERENT intervals, one after the other:
    // two long intervals per tile (the trace: 1.95 us per fold and group = 3.4 of a 29.6 us tile period at top-1).
    // Group 0 therefore DEFERS its fold by one interval -- it first reads the next tile's first slice, then folds in front
    // of that slice's MFMAs (which start from C = 0 and do not read the accumulators) -- so both groups fold in the same
    // interval and a tile has one long interval instead of two.
    // top-1 keeps the 12 fragments of the next tile's first slice in registers across the deferred fold; top-2 has no registers
    // for that (it spills) and reads them AFTER the fold, in the same interval; top-4 (64 list registers) spills either way and
    // ran 60 % slower deferred: it keeps the two-interval schedule.  Same box, 262 144 x 1 M: top-1 443.5 / 445.6 -> 434.9 /
    // 434.2 ms, top-2 493.2 / 492.6 -> 474.0 / 473.1 ms (r03 experiment 14).
    constexpr bool DEFER = LL || K <= 2;    // LL: no list registers, every K runs top-1's schedule
    constexpr bool LATE_READ = !LL && K == 2;
    bool pending = false;
    int n0_pending = 0;
    for (int s = 0; s < S; ++s) {
      const char* slot = smem + (s & 3) * G2_SLOT_BYTES;
      half8 fx[8], fw[4];
      auto read_frags = [&]() {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) fw[ni] = *(const half8*)(slot + woff + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) fx[mi] = *(const half8*)(slot + xoff + mi * 1024);
      };
      auto issue_and_wait = [&]() {
        if (s + 3 < S) {
          issue();
          SMI_WAIT_VMCNT(8);
        } else if (s + 2 < S) {
          SMI_WAIT_VMCNT(4);
        } else {
          SMI_WAIT_VMCNT(0);
        }
        SMI_LGKM0_BARRIER();
      };
      if (LATE_READ && pending) {  // group 0, first slice of the next tile: no fragment is live across the fold
        issue_and_wait();
        fold(n0_pending);
        pending = false;
        // The reads follow this interval's LDS-DMA issue: as C++ loads each would get `s_waitcnt vmcnt(0)` from hipcc's alias
        // tracking (gemm.hip) and wait for the slice that was just requested.  One asm statement issues and retires them.
        const unsigned wa = (unsigned)(size_t)(slot + woff), xa = (unsigned)(size_t)(slot + xoff);
        asm volatile(
            "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:1024\n\tds_read_b128 %2, %12 offset:2048\n\t"
            "ds_read_b128 %3, %12 offset:3072\n\tds_read_b128 %4, %13\n\tds_read_b128 %5, %13 offset:1024\n\t"
            "ds_read_b128 %6, %13 offset:2048\n\tds_read_b128 %7, %13 offset:3072\n\tds_read_b128 %8, %13 offset:4096\n\t"
            "ds_read_b128 %9, %13 offset:5120\n\tds_read_b128 %10, %13 offset:6144\n\tds_read_b128 %11, %13 offset:7168\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(fw[0]), "=&v"(fw[1]), "=&v"(fw[2]), "=&v"(fw[3]), "=&v"(fx[0]), "=&v"(fx[1]), "=&v"(fx[2]), "=&v"(fx[3]),
              "=&v"(fx[4]), "=&v"(fx[5]), "=&v"(fx[6]), "=&v"(fx[7])
            : "v"(wa), "v"(xa)
            : "memory");
      } else {
        read_frags();
        issue_and_wait();
        if (DEFER && pending) {  // top-1, group 0
          fold(n0_pending);
          pending = false;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if (k == 0) {
        // first slice of a y tile: C = 0 is an operand of the instruction, the 128 accumulator registers
        // are never cleared by hand
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int mi = 0; mi < 8; ++mi)
            acc.v[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ni], fx[mi], z, 0, 0, 0);
      } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int mi = 0; mi < 8; ++mi)
            acc.v[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ni], fx[mi], acc.v[ni][mi], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      SMI_BARRIER();
      if (++k == nt) {
        k = 0;
        if constexpr (DEFER) {
          if (wr == 0) {
            pending = true;
            n0_pending = n0;
          } else {
            fold(n0);
          }
        } else {
#ifdef SMI_XSIM_TRACE
          fold(n0);
#else
          XS_FOLD(n0);  // (!LL, K = 4: the macro, not the lambda -- 76 bytes of spills otherwise)
#endif
        }
        n0 += G2_BN;
      }
    }
    if (DEFER && pending) fold(n0_pending);
#ifdef SMI_XSIM_TRACE
    if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 256) {
      unsigned long long* o = xs_trace_buf + (blockIdx.x * 2 + wr) * 4;
      o[0] = wall_clock64() - tr_t0;
      o[1] = tr_fold;
      o[2] = tr_n;
      o[3] = tr_max;
    }
#endif
    if (wr == 0) SMI_BARRIER();
  }

  if constexpr (LL) {  // the lists are final: one thread per row writes its chunk-partial list
    __syncthreads();
    if (tid < 256) {
      const size_t o = ((size_t)chunk * nx_pad + m0 + tid) * K;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const unsigned long long key = lists[tid * K + j];
        ps[o + j] = xs_unord((uint32_t)(key >> 32));
        pi[o + j] = (int)(0xffffffffu - (uint32_t)key);
      }
    }
    return;
  }
  // a row's candidates sit in the 4 lane groups (kg) of 4 column waves: join the lane groups with
  // xor-shuffles (both partners end with the merged list), then the 4 waves through LDS
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
    for (int step = 16; step <= 32; step <<= 1) {
      float os[KR];
      int oi[KR];
#pragma unroll
      for (int j = 0; j < KR; ++j) {
        os[j] = __shfl_xor(best[mi].s[j], step, 64);
        oi[j] = __shfl_xor(best[mi].i[j], step, 64);
      }
#pragma unroll
      for (int j = 0; j < KR; ++j) best[mi].push(os[j], oi[j]);
    }
  }
  __syncthreads();
  float* ls = (float*)smem;  // [256 rows][4 waves][K]
  int* li = (int*)(smem + 256 * 4 * K * 4);
  if (kg == 0) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int row = wr * 128 + mi * 16 + l15;
#pragma unroll
      for (int j = 0; j < KR; ++j) {
        ls[(row * 4 + wc) * K + j] = best[mi].s[j];
        li[(row * 4 + wc) * K + j] = best[mi].i[j];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int row = threadIdx.x;
    TopK<K> t;
    t.init();
    for (int c = 0; c < 4 * K; ++c) t.push(ls[row * 4 * K + c], li[row * 4 * K + c]);
    const size_t o = ((size_t)chunk * nx_pad + m0 + row) * K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ps[o + j] = t.s[j];
      pi[o + j] = t.i[j];
    }
  }
}

// final merge over chunks; one thread per x row.  k_out <= K.
template <int K>
__global__ __launch_bounds__(256) void xsim_merge_kernel(const float* __restrict__ ps,
                                                         const int* __restrict__ pi, int nchunks,
                                                         int64_t nx, int nx_pad, int k_out,
                                                         int64_t y_off, int32_t* __restrict__ idx,
                                                         float* __restrict__ score) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= nx) return;
  TopK<K> t;
  t.init();
  for (int c = 0; c < nchunks; ++c) {
    const size_t o = ((size_t)c * nx_pad + row) * K;
#pragma unroll
    for (int j = 0; j < K; ++j) t.push(ps[o + j], pi[o + j]);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    if (j < k_out) {
      const bool valid = t.i[j] != 0x7fffffff;
      idx[row * k_out + j] = valid ? (int32_t)(t.i[j] + y_off) : -1;
      score[row * k_out + j] = t.s[j];
    }
  }
}

// ---------------------------------------------------------------- normalise
// dst[r,:] = f16(src[r,:] / max(||src[r,:]||, 1e-12)), rows >= n_valid zeroed.
template <typename T>
__global__ __launch_bounds__(256) void l2norm_kernel(const T* __restrict__ src, f16* __restrict__ dst,
                                                     int64_t rows, int64_t rows_pad, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows_pad) return;
  f16* o = dst + r * d;
  if (r >= rows) {
    for (int c = lane; c < d; c += 64) o[c] = (f16)0.f;
    return;
  }
  const T* s = src + r * d;
  // fast path (d = 1024 in SONAR): one pass, 16 B per lane per access, the row stays in registers
  constexpr int VEC = 16 / sizeof(T);      // elements per 16-B access
  constexpr int MAXV = 4;                  // up to d = 64 * VEC * MAXV
  if (d % (64 * VEC) == 0 && d <= 64 * VEC * MAXV && d % 512 == 0) {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int nv = d / (64 * VEC);
    vec_t v[MAXV];
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
      if (k < nv) {
        v[k] = *(const vec_t*)(s + (k * 64 + lane) * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) q += (float)v[k][e] * (float)v[k][e];
      }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    if constexpr (VEC == 8) {
#pragma unroll
      for (int k = 0; k < MAXV; ++k)
        if (k < nv) {
          half8 h;
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] = (f16)((float)v[k][e] * inv);
          *(half8*)(o + (k * 64 + lane) * 8) = h;
        }
    } else {  // fp32 source: two 16-B loads make one 16-B store -- pair the accesses of lanes' chunks
#pragma unroll
      for (int k = 0; k < MAXV; ++k)
        if (k < nv) {
          half4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (f16)((float)v[k][e] * inv);
          *(half4*)(o + (k * 64 + lane) * 4) = h;
        }
    }
    return;
  }
  float q = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float v = (float)s[c];
    q += v * v;
  }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(q)), 1e-12f);
  for (int c = lane; c < d; c += 64) o[c] = (f16)((float)s[c] * inv);
}

hipError_t launch_l2_normalize(const void* src, int src_is_f32, f16* dst, int64_t rows, int d,
                               hipStream_t stream) {
  const int64_t rows_pad = (rows + G2_BM - 1) / G2_BM * G2_BM;
  if (rows_pad == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((rows_pad + 3) / 4);
  if (src_is_f32)
    hipLaunchKernelGGL(l2norm_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)src,
                       dst, rows, rows_pad, d);
  else
    hipLaunchKernelGGL(l2norm_kernel<f16>, dim3(blocks), dim3(256), 0, stream, (const f16*)src, dst,
                       rows, rows_pad, d);
  return hipGetLastError();
}

// ------------------------------------------------- k-way merge of per-shard top-k lists
// part_score / part_idx: [parts][n][k], every list sorted (score desc, index asc) -- what smi_xsim_topk
// returns.  out: the k best of the union in the same total order.  One thread per row; used to fold the
// per-rank partial y-side neighbour lists of the sharded margin scoring.
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ ps, const int32_t* __restrict__ pi,
                                                         int parts, int64_t n, int k, float* __restrict__ os,
                                                         int32_t* __restrict__ oi) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  int head[64];
  for (int p = 0; p < parts; ++p) head[p] = 0;
  for (int j = 0; j < k; ++j) {
    float bv = -INFINITY;
    int bi = 0x7fffffff, bp = -1;
    for (int p = 0; p < parts; ++p) {
      if (head[p] >= k) continue;
      const size_t o = ((size_t)p * n + r) * k + head[p];
      const float v = ps[o];
      const int id = pi ? pi[o] : p * k + head[p];
      if (bp < 0 || v > bv || (v == bv && id < bi)) {
        bv = v;
        bi = id;
        bp = p;
      }
    }
    if (bp >= 0) ++head[bp];
    os[r * k + j] = bv;
    if (oi) oi[r * k + j] = bp >= 0 && pi ? bi : -1;
  }
}

hipError_t launch_topk_merge(const float* ps, const int32_t* pi, int parts, int64_t n, int k, float* os, int32_t* oi,
                             hipStream_t stream) {
  if (parts < 1 || parts > 64 || k < 1 || k > 8 || n <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ps, pi, parts, n, k,
                     os, oi);
  return hipGetLastError();
}

// ------------------------------------------------- margin re-scoring of the k-NN candidates
// LASER xsim (source/xsim.py: _score_margin / _score_knn): for x_i and its k nearest y candidates,
//   score(i, j) = margin(cos(x_i, y_j), (mean_kNN(x_i) + mean_kNN(y_j)) / 2),
// margin = ratio a / b, distance a - b, or (kind 2) the plain cosine a; the prediction is the candidate
// with the best score (first one on ties, as numpy's argmax).  mean_kNN(x_i) is the mean of x_i's own k
// forward scores, mean_kNN(y_j) the mean of y_j's k backward scores bwd[j][:].
// err_count (nullable) accumulates the rows whose prediction is not the aligned index i + x_off.
__global__ __launch_bounds__(256) void margin_select_kernel(const float* __restrict__ fs, const int32_t* __restrict__ fi,
                                                            int64_t nx, int k, const float* __restrict__ bwd,
                                                            int64_t ny, int kind, int64_t x_off,
                                                            int32_t* __restrict__ pred, float* __restrict__ pm,
                                                            int32_t* __restrict__ err_count) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int wrong = 0;
  if (r < nx) {
    float xm = 0.f;
    for (int j = 0; j < k; ++j) xm += fs[r * k + j];
    xm /= (float)k;
    float best = -INFINITY;
    int bj = 0;
    for (int j = 0; j < k; ++j) {
      const float a = fs[r * k + j];
      const int64_t y = fi[r * k + j];
      float m = a;
      if (kind != 2) {
        float ym = 0.f;
        if (y >= 0 && y < ny)
          for (int q = 0; q < k; ++q) ym += bwd[y * k + q];
        ym /= (float)k;
        const float b = 0.5f * (xm + ym);
        m = kind == 0 ? a / b : a - b;
      }
      if (j == 0 || m > best) {
        best = m;
        bj = j;
      }
    }
    const int p = fi[r * k + bj];
    pred[r] = p;
    if (pm) pm[r] = best;
    wrong = (int64_t)p != r + x_off;
  }
  if (err_count) {
    const unsigned long long bal = __ballot(wrong);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(err_count, (int)__popcll(bal));
  }
}

hipError_t launch_margin_select(const float* fs, const int32_t* fi, int64_t nx, int k, const float* bwd, int64_t ny,
                                int kind, int64_t x_off, int32_t* pred, float* pm, int32_t* err_count,
                                hipStream_t stream) {
  if (nx <= 0 || k < 1 || k > 8 || kind < 0 || kind > 2 || (kind != 2 && (!bwd || ny <= 0))) return hipErrorInvalidValue;
  hipLaunchKernelGGL(margin_select_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, stream, fs, fi, nx, k, bwd,
                     ny, kind, x_off, pred, pm, err_count);
  return hipGetLastError();
}

static int xsim_chunks(int64_t nty) { return (int)(nty < 8 ? nty : 8); }
static int round_k(int k) { return k <= 1 ? 1 : (k <= 2 ? 2 : (k <= 4 ? 4 : 8)); }

// SMI_XSIM_TM=0: mine on the row-major normalised matrices (rounds 1-2) instead of tile-major copies -- A/B switch
static bool xsim_tm() {
  return tune(TUNE_XSIM_TM, 1) != 0;
}
// SMI_XSIM_LL=0: per-lane top-k lists in registers (rounds 1-3; k = 8 then runs on the 128x128 engine) instead of the
// per-row lists in LDS -- A/B switch (the workspace formula does not depend on it)
static bool xsim_ll() {
  return tune(TUNE_XSIM_LL, 1) != 0;
}
// top-1 keeps its list in registers (16 VGPRs, no pressure there: 436.7 vs 439.6 ms with LDS lists at 262 144 x 1 M); k >= 2
// run on LDS lists: 475.1 -> 450.0 ms (k = 2), 545.7 -> 454.3 (k = 4), 657.9 -> 484.2 (k = 8, which the register version could
// only run on the 128x128 engine) -- same box, identical indices and scores (profiles/r04_experiments.txt, experiment 2)
static bool xsim_use_ll(int K) { return K >= 2 && xsim_ll(); }
static bool xsim_on_256(int K) { return K <= 4 || xsim_ll(); }
// per-chunk partial lists: [chunks][nx_pad][K] scores + indices
static size_t xsim_lists_bytes(int64_t nx_pad, int64_t ny_pad, int K) {
  return (size_t)xsim_chunks(ny_pad / GT_BN) * nx_pad * K * 8;
}
// ... followed (256x256 engine) by the tile-major copies of Xn and Yn
// The formula does NOT depend on the tuning switches (XSIM_LL decides whether k = 8 runs on the 256x256 engine and therefore
// whether the copies are written): it is the upper bound over their states, so a switch set by another host thread between a
// caller's workspace query and its smi_xsim_topk can never make xsim_run write past the size the API validated.
size_t xsim_workspace_bytes(int64_t nx_pad, int64_t ny_pad, int k, int d) {
  const int K = round_k(k);
  const size_t tm = (size_t)(nx_pad + ny_pad) * d * sizeof(f16);
  return xsim_lists_bytes(nx_pad, ny_pad, K) + tm;
}

template <int K>
static hipError_t xsim_run(const f16* Xn, int64_t nx, int64_t nx_pad, const f16* Yn, int64_t ny,
                           int64_t ny_pad, int d, int k, int64_t y_off, int32_t* idx, float* score,
                           void* ws, hipStream_t stream) {
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)xsim_tile_kernel<K>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done.set();
  }
  float* ps = (float*)ws;
  hipError_t e;
  int nchunks;
  if (xsim_on_256(K)) {
    // 256x256 tiles, continuous slice stream
    const bool ll = xsim_use_ll(K);
    constexpr int lds_ll = G2_LDS_BYTES + 256 * K * 8, lds_reg = G2_LDS_BYTES + 4096;
    static DeviceOnce attr256_done;
    if (!attr256_done.done()) {
      e = hipFuncSetAttribute((const void*)xsim_tile256_kernel<K, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_ll);
      if (e != hipSuccess) return e;
      e = hipFuncSetAttribute((const void*)xsim_tile256_kernel<K, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_ll);
      if (e != hipSuccess) return e;
      if constexpr (K <= 4) {
        e = hipFuncSetAttribute((const void*)xsim_tile256_kernel<K, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_reg);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)xsim_tile256_kernel<K, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_reg);
        if (e != hipSuccess) return e;
      }
      attr256_done.set();
    }
    const int ntx = (int)(nx_pad / G2_BM), nty = (int)(ny_pad / G2_BN);
    nchunks = xsim_chunks(nty);
    const int tpc = (nty + nchunks - 1) / nchunks;
    int* pi = (int*)((char*)ws + (size_t)nchunks * nx_pad * K * 4);
    const f16* xa = Xn;
    const f16* ya = Yn;
    const bool tm = xsim_tm();
    if (tm) {
      // tile-major copies of both operands behind the partial lists (one pass over each: ~1.5 ms of the 1.8 s at 1 M x 1 M)
      f16* xtm = (f16*)((char*)ws + xsim_lists_bytes(nx_pad, ny_pad, K));
      f16* ytm = xtm + (size_t)nx_pad * d;
      const f16* src[2] = {Xn, Yn};
      f16* dst[2] = {xtm, ytm};
      const int64_t rows[2] = {nx_pad, ny_pad};
      for (int o = 0; o < 2; ++o)
        for (int64_t r0 = 0; r0 < rows[o]; r0 += 65535LL * TM_ROWS) {  // grid.y limit of the pack kernel
          const int64_t nr = std::min<int64_t>(rows[o] - r0, 65535LL * TM_ROWS);
          e = launch_pack_tile_major(src[o] + r0 * d, dst[o] + r0 * d, (int)nr, d, 0, stream);
          if (e != hipSuccess) return e;
        }
      xa = xtm;
      ya = ytm;
    }
#define SMI_XSIM_LAUNCH(TMV, LLV, LDS)                                                                              \
  hipLaunchKernelGGL((xsim_tile256_kernel<K, TMV, LLV>), dim3(ntx * nchunks), dim3(G2_THREADS), LDS, stream, xa, ya, d, \
                     ntx, nty, nchunks, tpc, (int)ny, (int)nx_pad, ps, pi)
    if (ll) {
      if (tm) SMI_XSIM_LAUNCH(true, true, lds_ll);
      else SMI_XSIM_LAUNCH(false, true, lds_ll);
    } else {
      if constexpr (K <= 4) {
        if (tm) SMI_XSIM_LAUNCH(true, false, lds_reg);
        else SMI_XSIM_LAUNCH(false, false, lds_reg);
      }
    }
#undef SMI_XSIM_LAUNCH
  } else {
    const int ntx = (int)(nx_pad / GT_BM), nty = (int)(ny_pad / GT_BN);
    nchunks = xsim_chunks(nty);
    const int tpc = (nty + nchunks - 1) / nchunks;
    int* pi = (int*)((char*)ws + (size_t)nchunks * nx_pad * K * 4);
    hipLaunchKernelGGL(xsim_tile_kernel<K>, dim3(ntx * nchunks), dim3(GT_THREADS), GT_LDS_BYTES,
                       stream, Xn, Yn, d, ntx, nty, nchunks, tpc, (int)ny, (int)nx_pad, ps, pi);
  }
  int* pi = (int*)((char*)ws + (size_t)nchunks * nx_pad * K * 4);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(xsim_merge_kernel<K>, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, stream,
                     ps, pi, nchunks, nx, (int)nx_pad, k, y_off, idx, score);
  return hipGetLastError();
}

hipError_t launch_xsim_topk(const f16* Xn, int64_t nx, int64_t nx_pad, const f16* Yn, int64_t ny,
                            int64_t ny_pad, int d, int k, int64_t y_off, int32_t* idx, float* score,
                            void* ws, hipStream_t stream) {
  if (nx <= 0 || ny <= 0 || k < 1 || k > 8 || d % GT_BK || nx_pad % GT_BM || ny_pad % GT_BN ||
      ny_pad > 0x7fffff00LL || nx_pad > 0x7fffff00LL || nx_pad % G2_BM || ny_pad % G2_BN)
    return hipErrorInvalidValue;
  switch (round_k(k)) {
    case 1: return xsim_run<1>(Xn, nx, nx_pad, Yn, ny, ny_pad, d, k, y_off, idx, score, ws, stream);
    case 2: return xsim_run<2>(Xn, nx, nx_pad, Yn, ny, ny_pad, d, k, y_off, idx, score, ws, stream);
    case 4: return xsim_run<4>(Xn, nx, nx_pad, Yn, ny, ny_pad, d, k, y_off, idx, score, ws, stream);
    default: return xsim_run<8>(Xn, nx, nx_pad, Yn, ny, ny_pad, d, k, y_off, idx, score, ws, stream);
  }
}

}  // namespace smi

#ifdef SMI_XSIM_TRACE
extern "C" int smi_debug_xsim_trace(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(smi::xs_trace_buf), sizeof(smi::xs_trace_buf));
}
#endif
