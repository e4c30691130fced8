// Shared device/host helpers for the gfx950 (MI355X / CDNA4) SONAR hot path.
// Wavefront = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smi {

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef f16 half4 __attribute__((ext_vector_type(4)));
typedef f16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SMI_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define SMI_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// Direct global->LDS DMA of 16 B per lane. LDS destination = wave-uniform
// base + lane*16 (hardware rule), global source is per-lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(SMI_GLOBAL_PTR(gsrc), SMI_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// 4 B per lane (LDS destination = wave-uniform base + lane*4): per-tile constants (bias slices, row statistics) that travel with
// a tile's pipeline fill instead of through registers
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(SMI_GLOBAL_PTR(gsrc), SMI_LDS_PTR(lds_wave_base), 4, 0, 0);
}

// glds16 with the non-temporal cache policy (aux = 2, `nt`): for lines that exactly ONE workgroup reads, once.
__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(SMI_GLOBAL_PTR(gsrc), SMI_LDS_PTR(lds_wave_base), 16, 0, 2);
}

// Non-temporal (streaming) store for kernel OUTPUTS written as whole 128-B lines (16 B per lane, contiguous across the
// wave): the data is consumed by the next kernel, usually on other XCDs, so keeping it in this XCD's L2 only evicts the
// operand panels the workgroups are sharing.  Measured on the tile-major GEMM epilogues (r02 experiment 26): C2 step
// 109.6 -> 108.6 ms (the attention kernel reads the QKV output 9 % faster, the FFN pair ~1 %), decoder and speech
// -0.5..1 %.  NOT for partial-line stores: the 8-B pieces of the attention output went 4.4 -> 10.5 ms with it.
template <typename T>
__device__ __forceinline__ void store_nt(T* p, const T& v) {
  __builtin_nontemporal_store(v, p);
}

// MODE.FP16_OVFL (hardware register MODE, bit 23): with it set, a VALU fp16 result that overflows is clamped to +-65504
// instead of becoming +-inf (true infinities still pass).  The waves that store split-K PARTIAL sums in fp16 set it at entry:
// a partial above fp16's range must not turn a representable full sum into inf (tests/test_gpu_kernels.py).
__device__ __forceinline__ void fp16_saturate_on() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }

// Cross-lane reductions: the whole wave (wave_sum / wave_max, result in every lane), and for the single-query
// attention kernels (lane = 8 * pg + c) over the 8 lanes c of a position group and over the 8 position groups pg
// with the lane's c kept.  DPP row operations and the gfx950 lane-swap
// instructions run at VALU rate; __shfl_xor compiles to ds_bpermute_b32, an LDS-crossbar round trip per step
// (-DSMI_SHFL_REDUCE restores the shuffles for A/B builds).
#ifndef SMI_SHFL_REDUCE
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {              // all 64 lanes, result in every lane
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);                                           // 8 lanes joined
  v += dpp_f<0x140>(v);                                           // row_mirror: the other half of the 16-lane row
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum_over_8(float v) {            // lanes c = 0..7 of a group: xor 1, 2, then mirror
  v += dpp_f<0xB1>(v);                                            // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);                                            // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);                                           // row_half_mirror: lane i <-> 7 - i
  return v;
}
__device__ __forceinline__ float sum_over_groups_of_8(float v) {  // same c, all 8 position groups
  v += dpp_f<0x128>(v);                                           // row_ror:8 (lane i <- lane i + 8 within 16)
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);             // rows 16 apart joined on both sides
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float max_over_groups_of_8(float v) {
  v = fmaxf(v, dpp_f<0x128>(v));
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
#else
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sum_over_8(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}
__device__ __forceinline__ float sum_over_groups_of_8(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float max_over_groups_of_8(float v) {
  v = fmaxf(v, __shfl_xor(v, 8, 64));
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
#endif

// WEIGHT PREFETCH by surplus workgroups (round 4).  A small-batch forward is a chain of latency-bound launches that
// leaves HBM idle most of the time, and then streams a layer's 16.8 MB FFN matrices cold: 18.2 us for the FFN-inner
// projection at M = 256 against 14.0 us when the matrix was read just before (tools/probe_cold_weights.py).  A row
// kernel that occupies a quarter of the CUs therefore carries extra workgroups that simply READ the weights a later
// GEMM of the layer will ask for: the lines land in the Infinity Cache (memory side, shared by all XCDs).  Workgroup
// `wg` of `nwg` streams 4-KiB pieces wg, wg + nwg, ... with PF_DEPTH x 16-B loads in flight per thread (a 16.8 MB matrix
// over 192 workgroups is 22 loads per thread: ONE round trip; with 8 in flight it was three and the host kernel grew by
// 1.4 us); the xor of the data is stored under a condition that is practically never true so that the loads stay.
__device__ __forceinline__ void prefetch_range(const void* p, size_t bytes, int wg, int nwg, unsigned* sink) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* q = (const u32x4*)p;
  const size_t n16 = bytes >> 4;
  const size_t step = (size_t)nwg * blockDim.x;
  unsigned acc = 0;
  constexpr int PF_DEPTH = 24;
  for (size_t i = (size_t)wg * blockDim.x + threadIdx.x; i < n16; i += step * PF_DEPTH) {
    u32x4 v[PF_DEPTH];
#pragma unroll
    for (int j = 0; j < PF_DEPTH; ++j) {
      const size_t idx = min(i + j * step, n16 - 1);  // clamped, branch-free: every load is issued before the first use
      v[j] = q[idx];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PF_DEPTH; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
  }
  if (acc == 0x9e3779b9u && sink) *sink = acc;
}

// XCD-aware block id remap (bijective for any grid size): hardware deals
// block b to XCD b%8; give every XCD one contiguous range of logical ids so
// neighbouring tiles share that XCD's private L2.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// TILE-MAJOR operand layout of a K-major fp16 matrix A[R][K] (R % 256 == 0, K % 32 == 0): block
// (r/256, k/32) is 16 KiB contiguous and holds the LDS image the 256x256 tile engine wants (row
// rr = r%256 at rr*64 B, 16-B chunk c = (k%32)/8 at slot c ^ tm_swz(rr)).  A K slice of a tile is
// then ONE linear 16 KiB burst instead of 256 pieces of 64 B (gemm_tile256.hpp).
// Slot swizzle of a 64-B row: q = (rr>>2)&3 -> q ^ ((q&1)<<1), i.e. 0,3,2,1.  ds_read_b128 serves a wave
// in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) (MI355X_MICROARCH.md, LDS), and a
// 16x16x32 MFMA fragment read has lane l -> (row l&15, chunk l>>4): with this map the four 4-lane sets
// of a group that share a bank quarter (row mod 4) land on four different 16-B slots -> conflict free.
// (The plain q map of the first version was conflict free for the 32x32x16 fragment pattern only.)
__host__ __device__ __forceinline__ int tm_swz(int rr) {
  const int q = (rr >> 2) & 3;
  return q ^ ((q & 1) << 1);
}
constexpr int TM_ROWS = 256;
constexpr int TM_BLOCK = TM_ROWS * 32;  // elements per block
// element offset of A[r][k]
__host__ __device__ __forceinline__ size_t tm_offset(int r, int k, int K) {
  const int rr = r & 255, c = (k >> 3) & 3;
  return ((size_t)(r >> 8) * (K >> 5) + (k >> 5)) * TM_BLOCK + rr * 32 + ((c ^ tm_swz(rr)) << 3) + (k & 7);
}

}  // namespace smi
