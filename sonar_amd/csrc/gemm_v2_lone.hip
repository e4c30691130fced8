// The 160 x 256 LONE unit of the 4-wave engine: the two FFN projections of a decode step at M = 1280 rows (BASELINE configs[4]:
// 256 sentences x beam 5; reference call site sonar/inference_pipelines/text.py:305-346, layer wiring
// sonar/models/sonar_text/factory.py:261-274).
//
// 1280 rows are 5 row tiles of 256 -- 160 work units for either projection (FFN inner: 5 x 32 tiles; FFN out: 5 x 4 tiles x 8 K
// parts), 96 of 256 CUs idle for 1.27 ms of every 3.6 ms step -- or 8 row tiles of 160: 256 units, one per CU.  On the 8-wave
// engine the smaller tile bought nothing (rounds 1 and 3: a lone unit there is bound by its two-barrier interval structure).  On the
// 4-wave structure a unit's time follows what it moves and multiplies (tools/micro/engine_v2.hip `lone`, HBM-cold weights as in a
// decode step: 23.8 us per launch for a 256 x 256 unit, 21.5 with the MFMA count of a 160-row tile, 18.2 with its slice bytes too).
//
// One unit per workgroup, no persistence: 4 waves (2 x 2), wave tile 80(m) x 128(n), 160 accumulators in AGPRs, the inline-asm step
// structure of gemm_v2.hpp (gemm_v2_lone_asm.inc: 10 MFMAs per block).  A K = 32 slice in LDS is X rows [160][64 B] + W rows
// [256][64 B] = 26 KiB, so the ring has SIX slots (156 KiB) and the DMA runs FIVE slices ahead -- a lone unit streams its operands
// cold from HBM and lives on that lead.  The slot index is a run-time value (6 does not divide the 32 slices of a K = 1024 unit):
// LDS addresses are base + slot * 26 KiB, two VALU adds per step.
//   X source: the tile-major image has 256-row blocks; the unit's 160 rows are 10 pieces of 16 rows x 64 B = 1 KiB, piece p at
//   global row 160 t + 16 p.  Wave w copies pieces 2w and 2w+1 (32 rows: never across a block boundary, one 2 KiB run), HALF of piece
//   8 + (w >> 1) (8 rows, under an EXEC mask of 32 lanes) and W pieces 4w .. 4w+3: 7 DMA instructions per wave and slice, every wave
//   the same count (the counted waits are wave-uniform code).
// Outputs: EPI_RELU_F16 -> the tile-major fp16 hidden activation (FFN inner); EPI_BIAS_F16 -> row-major fp16 split-K slabs
// [kz][M][N], the bias in part 0, saturating fp16 (FFN out; consumed by sum_ln_kernel exactly as the 8-wave engine's slabs).
#include <algorithm>

#include "gemm_epi.hpp"
#include "gemm_v2.hpp"
#include "gemm_v2_lone_asm.inc"
#include "kernels.hpp"

namespace smi {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int VM_XB = 10240;            // X part of a slice
constexpr int VM_SLOT = VM_XB + 16384;  // 26 KiB
constexpr int VM_NSLOT = 6;
constexpr int VM_LEAD = VM_NSLOT - 1;
constexpr int VM_LDS_BYTES = VM_NSLOT * VM_SLOT;  // 156 KiB
constexpr int VM_DMA = 7;                         // LDS-DMA instructions per wave and slice

struct VmFrag {
  half8 x[2][5];
  half8 w[2][2];
};

#define SMI_VM_IO(J, F, XB)                                                                                              \
  [w0] "v"(F.w[(J) & 1][0]), [w1] "v"(F.w[(J) & 1][1]), [x0] "v"(F.x[XB][0]), [x1] "v"(F.x[XB][1]), [x2] "v"(F.x[XB][2]),  \
      [x3] "v"(F.x[XB][3]), [x4] "v"(F.x[XB][4])

// Source cursor of a wave: this lane's address of X pieces 2w / 2w+1 (gx), of its half of X piece 8 + (w >> 1) (gh) and of W pieces
// 4w .. 4w+3 (gw) in the slice that is issued next; all three advance by one 16 KiB tile-major block per slice.
struct VmStream {
  const char* gx;
  const char* gh;
  const char* gw;
  int inc;
};

template <bool ZERO, int XB>
__device__ __forceinline__ void vm_step(VmFrag& f, VmStream& st, unsigned xa_nxt, unsigned wa_cur, unsigned wa_nxt, unsigned m0x,
                                        unsigned m0h, unsigned m0w, unsigned long long hmask) {
#define SMI_VM_B0(NAME)                                                                                                    \
  asm volatile(NAME##_STR                                                                                                  \
               : [nw0] "=&v"(f.w[1][0]), [nw1] "=&v"(f.w[1][1]), [nx0] "=&v"(f.x[XB ^ 1][0]), [nx1] "=&v"(f.x[XB ^ 1][1])    \
               : SMI_VM_IO(0, f, XB), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gpx] "v"(st.gx), [m0x] "s"(m0x)                 \
               : "memory", V2M_BLOCK0_CLOB)
#define SMI_VM_B1(NAME)                                                                                                    \
  asm volatile(NAME##_STR                                                                                                  \
               : [nw0] "=&v"(f.w[0][0]), [nw1] "=&v"(f.w[0][1]), [nx0] "=&v"(f.x[XB ^ 1][2]), [nx1] "=&v"(f.x[XB ^ 1][3])    \
               : SMI_VM_IO(1, f, XB), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gph] "v"(st.gh), [gpw] "v"(st.gw), [m0h] "s"(m0h), \
                 [m0w] "s"(m0w), [hm] "s"(hmask)                                                                          \
               : "memory", V2M_BLOCK1_CLOB)
#define SMI_VM_B2(NAME)                                                                                                    \
  asm volatile(NAME##_STR                                                                                                  \
               : [nw0] "=&v"(f.w[1][0]), [nw1] "=&v"(f.w[1][1]), [nx0] "=&v"(f.x[XB ^ 1][4])                              \
               : SMI_VM_IO(2, f, XB), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gpw] "v"(st.gw), [m0w] "s"(m0w)                 \
               : "memory", V2M_BLOCK2_CLOB)
#define SMI_VM_B3(NAME)                                                                                                    \
  asm volatile(NAME##_STR                                                                                                  \
               : [nw0] "=&v"(f.w[0][0]), [nw1] "=&v"(f.w[0][1])                                                          \
               : SMI_VM_IO(3, f, XB), [wa] "v"(wa_nxt), [gpw] "v"(st.gw), [m0w] "s"(m0w)                                   \
               : "memory", V2M_BLOCK3_CLOB)
  if constexpr (ZERO) {
    SMI_VM_B0(V2M_BLOCK0Z);
    SMI_VM_B1(V2M_BLOCK1Z);
    SMI_VM_B2(V2M_BLOCK2Z);
    SMI_VM_B3(V2M_BLOCK3Z);
  } else {
    SMI_VM_B0(V2M_BLOCK0);
    SMI_VM_B1(V2M_BLOCK1);
    SMI_VM_B2(V2M_BLOCK2);
    SMI_VM_B3(V2M_BLOCK3);
  }
#undef SMI_VM_B0
#undef SMI_VM_B1
#undef SMI_VM_B2
#undef SMI_VM_B3
  st.gx += st.inc;
  st.gh += st.inc;
  st.gw += st.inc;
}

// the 7 DMA instructions of one slice (pipeline fill)
__device__ __forceinline__ void vm_fill(VmStream& st, unsigned m0x, unsigned m0h, unsigned m0w, unsigned long long hmask) {
  asm volatile(
      "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
      "s_mov_b32 m0, %4\n\ts_mov_b64 exec, %6\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b64 exec, -1\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
      "global_load_lds_dwordx4 %2, off offset:2048\n\tglobal_load_lds_dwordx4 %2, off offset:3072"
      :
      : "v"(st.gx), "v"(st.gh), "v"(st.gw), "s"(m0x), "s"(m0h), "s"(m0w), "s"(hmask)
      : "memory");
  st.gx += st.inc;
  st.gh += st.inc;
  st.gw += st.inc;
}

#define SMI_VM_RDOUT(J, NL, MI, V) \
  asm volatile(V2M_RD_##J##_##NL##_##MI##_STR : "=v"(V[0]), "=v"(V[1]), "=v"(V[2]), "=v"(V[3]))

// EPI_RELU_F16: out = tile-major fp16 [M][N].  EPI_BIAS_F16: out = row-major fp16 slabs, part kz at out + kz * part_stride bytes.
// Unit id -> (row tile tm of 160, column tile tn of 256, K part kz of nt slices each).
template <int EPI>
__global__ __launch_bounds__(V2_THREADS) void gemm_v2_lone_kernel(const f16* __restrict__ X, const f16* __restrict__ W,
                                                                  const float* __restrict__ bias, void* __restrict__ out_, int M,
                                                                  int N, int K, int ksplit, size_t part_stride) {
  static_assert(EPI == EPI_RELU_F16 || EPI == EPI_BIAS_F16, "tile-major relu hidden activation or fp16 split-K slabs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, kg = lane >> 4;
  if constexpr (EPI == EPI_BIAS_F16) fp16_saturate_on();  // fp16 partial sums saturate instead of overflowing to inf

  // XCD x (blocks b = x mod 8) owns the panels 4x .. 4x+3 of the (tn, kz) space and all 8 row tiles of each: a W panel part is
  // fetched from HBM once per XCD and hit by the other seven row tiles in that XCD's L2
  const int ntm = M / 160, ntn = N / 256;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = logical % ntm, panel = logical / ntm;
  const int tn = panel % ntn, kz = panel / ntn;
  const int kblocks = K / 32, nt = kblocks / ksplit, kb0 = kz * nt;

  const unsigned lds0 = (unsigned)(size_t)smem;
  const int t_sw = (kg ^ tm_swz(l15)) << 4;
  const unsigned xa0 = lds0 + (wr * 80 + l15) * 64 + t_sw;
  const unsigned wa0 = lds0 + VM_XB + (wc * 128 + l15) * 64 + t_sw;
  const unsigned long long hmask = (wave & 1) ? 0xffffffff00000000ull : 0x00000000ffffffffull;

  // global rows of this wave's X pieces; a piece never crosses a 256-row block of the tile-major image
  const int r0 = tm * 160;
  const int rx = r0 + 32 * wave, rh = r0 + 128 + 16 * (wave >> 1);
  VmStream st;
  st.gx = (const char*)X + ((size_t)(rx >> 8) * kblocks + kb0) * (TM_BLOCK * 2) + (rx & 255) * 64 + lane * 16;
  st.gh = (const char*)X + ((size_t)(rh >> 8) * kblocks + kb0) * (TM_BLOCK * 2) + (rh & 255) * 64 + lane * 16;
  st.gw = (const char*)W + ((size_t)tn * kblocks + kb0) * (TM_BLOCK * 2) + wave * 4096 + lane * 16;
  st.inc = TM_BLOCK * 2;
  auto m0x_of = [&](int slot) { return lds0 + slot * VM_SLOT + wave * 2048; };
  auto m0h_of = [&](int slot) { return lds0 + slot * VM_SLOT + (8 + (wave >> 1)) * 1024; };
  auto m0w_of = [&](int slot) { return lds0 + slot * VM_SLOT + VM_XB + wave * 4096; };

  // the lane's 32 bias values (columns wc*128 + ni*16 + 4*kg + r): plain loads, first used in the read-out
  // (issued unconditionally, from a valid address when there is no bias, and selected afterwards: a conditional load is a
  // branch + a wait per load in hipcc's code, two memory round trips in front of the pipeline fill of a ~20 us unit)
  f32x4 bv[8];
  const bool use_bias = bias != nullptr && kz == 0;
  const float* bsrc = use_bias ? bias + tn * 256 + wc * 128 + 4 * kg : (const float*)W + 4 * kg;
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) bv[ni] = *(const f32x4*)(bsrc + ni * 16);

  // ---- pipeline fill: slices 0 .. LEAD-1 (nt >= LEAD + 1) ----
#pragma unroll
  for (int s = 0; s < VM_LEAD; ++s) vm_fill(st, m0x_of(s), m0h_of(s), m0w_of(s), hmask);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"((VM_LEAD - 1) * VM_DMA) : "memory");
  VmFrag f;
  asm volatile(
      "ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:1024\n\tds_read_b128 %2, %7 offset:2048\n\t"
      "ds_read_b128 %3, %7 offset:3072\n\tds_read_b128 %4, %7 offset:4096\n\t"
      "ds_read_b128 %5, %8\n\tds_read_b128 %6, %8 offset:1024\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(f.x[0][0]), "=&v"(f.x[0][1]), "=&v"(f.x[0][2]), "=&v"(f.x[0][3]), "=&v"(f.x[0][4]), "=&v"(f.w[0][0]),
        "=&v"(f.w[0][1])
      : "v"(xa0), "v"(wa0)
      : "memory");

  // ---- K loop: step s consumes slice s (slot s % 6), reads slice s + 1, issues slice s + 5 into the slot of slice s - 1.
  // At the top of step s slices s+1 .. s+4 are in flight; the counted wait leaves the 3 younger ones.  Past the unit's last
  // slice the cursor stays on it (inc = 0): the re-fetch lands in a slot nobody reads again and keeps the count uniform.
  int cs = 0;  // slot of the current slice
  auto step = [&](auto zero_tag, auto xb_tag, int s) {
    constexpr bool ZERO = decltype(zero_tag)::value;
    constexpr int XB = decltype(xb_tag)::value;
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"((VM_LEAD - 2) * VM_DMA) : "memory");
    if (s + VM_LEAD == nt) {  // the cursor has run past the unit's last slice: stay on it
      st.gx -= st.inc;
      st.gh -= st.inc;
      st.gw -= st.inc;
      st.inc = 0;
    }
    const int ns = cs + 1 == VM_NSLOT ? 0 : cs + 1;
    const int ds = cs == 0 ? VM_NSLOT - 1 : cs - 1;
    vm_step<ZERO, XB>(f, st, xa0 + ns * VM_SLOT, wa0 + cs * VM_SLOT, wa0 + ns * VM_SLOT, m0x_of(ds), m0h_of(ds), m0w_of(ds), hmask);
    cs = ns;
  };
  step(std::true_type{}, std::integral_constant<int, 0>{}, 0);
  step(std::false_type{}, std::integral_constant<int, 1>{}, 1);
  for (int s = 2; s < nt; s += 2) {
    step(std::false_type{}, std::integral_constant<int, 0>{}, s);
    step(std::false_type{}, std::integral_constant<int, 1>{}, s + 1);
  }

  // ---- read-out ----
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");
  if (!use_bias) {
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) bv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int cidx = (kg & 1) * 2 + (kg >> 1);
  char* obase;
  size_t mi_stride;
  if constexpr (EPI == EPI_RELU_F16) {
    // tile-major [M][N]: row m = r0 + wr*80 + mi*16 + l15 sits in block (m >> 8, k-block) at (m & 255) * 64 B
    obase = (char*)out_ + ((size_t)tn * 8 + wc * 4) * (TM_BLOCK * 2) + (size_t)((cidx ^ tm_swz(l15)) << 4);
    mi_stride = 0;
  } else {
    obase = (char*)out_ + (size_t)kz * part_stride + ((size_t)(r0 + wr * 80 + l15) * N + tn * 256 + wc * 128 + cidx * 8) * 2;
    mi_stride = (size_t)16 * N * 2;
  }
  auto row_ptr = [&](int mi) -> char* {
    if constexpr (EPI == EPI_RELU_F16) {
      const int m = r0 + wr * 80 + mi * 16 + l15;
      return obase + (size_t)(m >> 8) * (N >> 5) * (TM_BLOCK * 2) + (m & 255) * 64;
    } else {
      return obase + mi * mi_stride;
    }
  };
  constexpr size_t JSTEP = EPI == EPI_RELU_F16 ? (size_t)TM_BLOCK * 2 : 64;  // next 32-column k-block
#define SMI_VM_CHUNK(J, MI)                                                                                          \
  {                                                                                                                  \
    f32x4 va, vb;                                                                                                    \
    SMI_VM_RDOUT(J, 0, MI, va);                                                                                      \
    SMI_VM_RDOUT(J, 1, MI, vb);                                                                                      \
    va = va + bv[2 * (J)];                                                                                           \
    vb = vb + bv[2 * (J) + 1];                                                                                       \
    const uint2 ha = __builtin_bit_cast(uint2, epi_act_pack<EPI>(va)), hb = __builtin_bit_cast(uint2, epi_act_pack<EPI>(vb)); \
    const auto s0 = __builtin_amdgcn_permlane16_swap(ha.x, hb.x, false, false);                                      \
    const auto s1 = __builtin_amdgcn_permlane16_swap(ha.y, hb.y, false, false);                                      \
    const u32x4 chunk = {s0[0], s1[0], s0[1], s1[1]};                                                                \
    u32x4* dst = (u32x4*)(rp[MI] + (J) * JSTEP);                                                                     \
    if constexpr (EPI == EPI_RELU_F16)                                                                               \
      store_nt(dst, chunk);                                                                                          \
    else                                                                                                             \
      *dst = chunk;                                                                                                  \
  }
  char* rp[5];
#pragma unroll
  for (int mi = 0; mi < 5; ++mi) rp[mi] = row_ptr(mi);
#define SMI_VM_KB(J) SMI_VM_CHUNK(J, 0) SMI_VM_CHUNK(J, 1) SMI_VM_CHUNK(J, 2) SMI_VM_CHUNK(J, 3) SMI_VM_CHUNK(J, 4)
  SMI_VM_KB(0) SMI_VM_KB(1) SMI_VM_KB(2) SMI_VM_KB(3)
#undef SMI_VM_KB
#undef SMI_VM_CHUNK
}

bool gemm_v2_lone_fits(int M, int N, int K, int ksplit) {
  if (tune(TUNE_DEC_M160, 1) == 0) return false;
  if (M % 1280 || N % 256 || K % 32 || ksplit < 1 || (K / 32) % ksplit) return false;
  const int nt = K / 32 / ksplit;
  if (nt < VM_LEAD + 1 || nt % 2) return false;
  const int64_t units = (int64_t)(M / 160) * (N / 256) * ksplit, units256 = (int64_t)(M / 256) * (N / 256) * ksplit;
  // lone units only: every unit on a CU of its own, and more CUs busy than with 256-row tiles
  return units <= num_cus() && units > units256;
}

template <int EPI>
static hipError_t launch_lone_unit(const f16* X, const f16* W, const float* bias, void* out, int M, int N, int K, int ksplit,
                                   size_t part_stride, hipStream_t stream) {
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_v2_lone_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       VM_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done.set();
  }
  const int grid = (M / 160) * (N / 256) * ksplit;
  hipLaunchKernelGGL((gemm_v2_lone_kernel<EPI>), dim3(grid), dim3(V2_THREADS), VM_LDS_BYTES, stream, X, W, bias, out, M, N, K,
                     ksplit, part_stride);
  return hipGetLastError();
}

// relu = 1: tile-major fp16 relu output [M][N] (ksplit must be 1); relu = 0: row-major fp16 slabs [ksplit][M][N]
hipError_t launch_gemm_v2_lone(int relu, const f16* X, const f16* W, const float* bias, void* out, int M, int N, int K, int ksplit,
                               hipStream_t stream) {
  if (relu) {
    if (ksplit != 1) return hipErrorInvalidValue;
    return launch_lone_unit<EPI_RELU_F16>(X, W, bias, out, M, N, K, 1, 0, stream);
  }
  return launch_lone_unit<EPI_BIAS_F16>(X, W, bias, out, M, N, K, ksplit, (size_t)M * N * 2, stream);
}

}  // namespace smi
