// LONE units of the 4-wave engine, 128 / 160 / 192 rows x 256 columns: the two FFN projections of a decode step at M = 1280 rows
// (BASELINE configs[4]: 256 sentences x beam 5; reference call site sonar/inference_pipelines/text.py:305-346, layer wiring
// sonar/models/sonar_text/factory.py:261-274) and of every other row count whose 256-row tiles would leave CUs idle (the small-batch
// text encoder: BASELINE configs[0], 1312 tokens = 1536 rows = 8 tiles of 192; text.py:173-183).
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
//   the same count (the counted waits are wave-uniform code).  The 128-row unit (MI 4) copies 2 X pieces per wave, the 192-row unit
//   (MI 6) three whole ones (its ring has 5 slots of 28 KiB).
// Outputs: EPI_RELU_F16 -> the tile-major fp16 hidden activation (FFN inner); EPI_BIAS_F16 -> row-major fp16 split-K slabs
// [kz][M][N], the bias in part 0, saturating fp16 (FFN out; consumed by sum_ln_kernel exactly as the 8-wave engine's slabs).
#include <algorithm>

#include "gemm_epi.hpp"
#include "gemm_v2.hpp"
#include "gemm_v2_lone_asm.inc"
#include "kernels.hpp"

namespace smi {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MI = 16-row blocks per wave: the unit has 32 MI rows
template <int MI>
struct VmShape {
  static constexpr int ROWS = 32 * MI;
  static constexpr int XB = 2048 * MI;                       // X part of a slice: 2 MI pieces of 1 KiB
  static constexpr int SLOT = XB + 16384;                    // 24 / 26 / 28 KiB
  static constexpr int NSLOT = (160 * 1024) / SLOT;          // 6 / 6 / 5
  static constexpr int LEAD = NSLOT - 1;                     // slices the DMA runs ahead
  static constexpr int LDS_BYTES = NSLOT * SLOT;             // 144 / 156 / 140 KiB
  static constexpr int DMA = MI == 4 ? 6 : 7;                // LDS-DMA instructions per wave and slice
};

template <int MI>
struct VmFrag {
  half8 x[2][MI];
  half8 w[2][2];
};

// Source cursor of a wave: this lane's address of its X pieces (ga, gb: whole pieces; gh: MI 5 its half of a piece, MI 6 a third
// whole piece, MI 4 unused) and of W pieces 4w .. 4w+3 (gw) in the slice that is issued next; all advance by one 16 KiB
// tile-major block per slice.  gb is stored pre-biased by -1 KiB (its DMA carries the instruction offset 1024, which shifts the
// LDS side too).
struct VmStream {
  const char* ga;
  const char* gb;
  const char* gh;
  const char* gw;
  int inc;
  __device__ __forceinline__ void advance() {
    ga += inc;
    gb += inc;
    gh += inc;
    gw += inc;
  }
};

struct VmDst {
  unsigned m0x, m0h, m0w;
};

#define SMI_VM_MF_IN4(J, F, XB) [w0] "v"(F.w[(J) & 1][0]), [w1] "v"(F.w[(J) & 1][1]), [x0] "v"(F.x[XB][0]), [x1] "v"(F.x[XB][1]), \
    [x2] "v"(F.x[XB][2]), [x3] "v"(F.x[XB][3])
#define SMI_VM_NW(J, F) [nw0] "=&v"(F.w[((J) + 1) & 1][0]), [nw1] "=&v"(F.w[((J) + 1) & 1][1])

template <int MI, bool ZERO, int XB>
__device__ __forceinline__ void vm_step(VmFrag<MI>& f, VmStream& st, unsigned xa_nxt, unsigned wa_cur, unsigned wa_nxt,
                                        const VmDst& d, unsigned long long hmask) {
  constexpr int XN = XB ^ 1;
  if constexpr (MI == 4) {
#define SMI_VM_STEP4(Z)                                                                                                       \
  asm volatile(V2M4_BLOCK0##Z##_STR : SMI_VM_NW(0, f), [nx0] "=&v"(f.x[XN][0]), [nx1] "=&v"(f.x[XN][1])                       \
               : SMI_VM_MF_IN4(0, f, XB), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gpx] "v"(st.ga), [gpx2] "v"(st.gb), [m0x] "s"(d.m0x) \
               : "memory", V2M4_BLOCK0_CLOB);                                                                                \
  asm volatile(V2M4_BLOCK1##Z##_STR : SMI_VM_NW(1, f), [nx0] "=&v"(f.x[XN][2]), [nx1] "=&v"(f.x[XN][3])                       \
               : SMI_VM_MF_IN4(1, f, XB), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gpw] "v"(st.gw), [m0w] "s"(d.m0w)               \
               : "memory", V2M4_BLOCK1_CLOB);                                                                                \
  asm volatile(V2M4_BLOCK2##Z##_STR : SMI_VM_NW(2, f)                                                                        \
               : SMI_VM_MF_IN4(2, f, XB), [wa] "v"(wa_cur), [gpw] "v"(st.gw), [m0w] "s"(d.m0w)                                 \
               : "memory", V2M4_BLOCK2_CLOB);                                                                                \
  asm volatile(V2M4_BLOCK3##Z##_STR : SMI_VM_NW(3, f)                                                                        \
               : SMI_VM_MF_IN4(3, f, XB), [wa] "v"(wa_nxt), [gpw] "v"(st.gw), [m0w] "s"(d.m0w)                                 \
               : "memory", V2M4_BLOCK3_CLOB)
    if constexpr (ZERO) {
      SMI_VM_STEP4(Z);
    } else {
      SMI_VM_STEP4();
    }
#undef SMI_VM_STEP4
  } else if constexpr (MI == 5) {
#define SMI_VM_IN5(J) SMI_VM_MF_IN4(J, f, XB), [x4] "v"(f.x[XB][4])
#define SMI_VM_STEP5(Z)                                                                                                       \
  asm volatile(V2M5_BLOCK0##Z##_STR : SMI_VM_NW(0, f), [nx0] "=&v"(f.x[XN][0]), [nx1] "=&v"(f.x[XN][1])                       \
               : SMI_VM_IN5(0), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gpx] "v"(st.ga), [gpx2] "v"(st.gb), [m0x] "s"(d.m0x)      \
               : "memory", V2M5_BLOCK0_CLOB);                                                                                \
  asm volatile(V2M5_BLOCK1##Z##_STR : SMI_VM_NW(1, f), [nx0] "=&v"(f.x[XN][2]), [nx1] "=&v"(f.x[XN][3])                       \
               : SMI_VM_IN5(1), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gph] "v"(st.gh), [gpw] "v"(st.gw), [m0h] "s"(d.m0h),      \
                 [m0w] "s"(d.m0w), [hm] "s"(hmask)                                                                           \
               : "memory", V2M5_BLOCK1_CLOB);                                                                                \
  asm volatile(V2M5_BLOCK2##Z##_STR : SMI_VM_NW(2, f), [nx0] "=&v"(f.x[XN][4])                                               \
               : SMI_VM_IN5(2), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gpw] "v"(st.gw), [m0w] "s"(d.m0w)                         \
               : "memory", V2M5_BLOCK2_CLOB);                                                                                \
  asm volatile(V2M5_BLOCK3##Z##_STR : SMI_VM_NW(3, f)                                                                        \
               : SMI_VM_IN5(3), [wa] "v"(wa_nxt), [gpw] "v"(st.gw), [m0w] "s"(d.m0w)                                           \
               : "memory", V2M5_BLOCK3_CLOB)
    if constexpr (ZERO) {
      SMI_VM_STEP5(Z);
    } else {
      SMI_VM_STEP5();
    }
#undef SMI_VM_STEP5
#undef SMI_VM_IN5
  } else {
#define SMI_VM_IN6(J) SMI_VM_MF_IN4(J, f, XB), [x4] "v"(f.x[XB][4]), [x5] "v"(f.x[XB][5])
#define SMI_VM_STEP6(Z)                                                                                                       \
  asm volatile(V2M6_BLOCK0##Z##_STR : SMI_VM_NW(0, f), [nx0] "=&v"(f.x[XN][0]), [nx1] "=&v"(f.x[XN][1])                       \
               : SMI_VM_IN6(0), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gpx] "v"(st.ga), [gpx2] "v"(st.gb), [m0x] "s"(d.m0x)      \
               : "memory", V2M6_BLOCK0_CLOB);                                                                                \
  asm volatile(V2M6_BLOCK1##Z##_STR : SMI_VM_NW(1, f), [nx0] "=&v"(f.x[XN][2]), [nx1] "=&v"(f.x[XN][3])                       \
               : SMI_VM_IN6(1), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gph] "v"(st.gh), [gpw] "v"(st.gw), [m0h] "s"(d.m0h),      \
                 [m0w] "s"(d.m0w)                                                                                            \
               : "memory", V2M6_BLOCK1_CLOB);                                                                                \
  asm volatile(V2M6_BLOCK2##Z##_STR : SMI_VM_NW(2, f), [nx0] "=&v"(f.x[XN][4]), [nx1] "=&v"(f.x[XN][5])                       \
               : SMI_VM_IN6(2), [xa] "v"(xa_nxt), [wa] "v"(wa_cur), [gpw] "v"(st.gw), [m0w] "s"(d.m0w)                         \
               : "memory", V2M6_BLOCK2_CLOB);                                                                                \
  asm volatile(V2M6_BLOCK3##Z##_STR : SMI_VM_NW(3, f)                                                                        \
               : SMI_VM_IN6(3), [wa] "v"(wa_nxt), [gpw] "v"(st.gw), [m0w] "s"(d.m0w)                                           \
               : "memory", V2M6_BLOCK3_CLOB)
    if constexpr (ZERO) {
      SMI_VM_STEP6(Z);
    } else {
      SMI_VM_STEP6();
    }
#undef SMI_VM_STEP6
#undef SMI_VM_IN6
  }
  st.advance();
}

// the DMA instructions of one slice (pipeline fill)
template <int MI>
__device__ __forceinline__ void vm_fill(VmStream& st, const VmDst& d, unsigned long long hmask) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024"
               :
               : "v"(st.ga), "v"(st.gb), "s"(d.m0x)
               : "memory");
  if constexpr (MI == 5)
    asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %2\n\tglobal_load_lds_dwordx4 %0, off\n\ts_mov_b64 exec, -1"
                 :
                 : "v"(st.gh), "s"(d.m0h), "s"(hmask)
                 : "memory");
  if constexpr (MI == 6)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(st.gh), "s"(d.m0h) : "memory");
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
               "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
               :
               : "v"(st.gw), "s"(d.m0w)
               : "memory");
  st.advance();
}

#define SMI_VM_RDOUT(MIV, J, NL, MI_, V) \
  asm volatile(V2M##MIV##_RD_##J##_##NL##_##MI_##_STR : "=v"(V[0]), "=v"(V[1]), "=v"(V[2]), "=v"(V[3]))

// SLAB = false: out = act(X W^T + bias) as tile-major fp16 [M][N] (EPI_RELU_F16 / EPI_BIAS_F16).  SLAB = true: out = row-major fp16
// split-K slabs, part kz at out + kz * part_stride bytes, the bias in part 0, saturating.
// Unit id -> (row tile tm of 32 MI rows, column tile tn of 256, K part kz of nt slices each).
template <int EPI, int MI, bool SLAB>
__global__ __launch_bounds__(V2_THREADS) void gemm_v2_lone_kernel(const f16* __restrict__ X, const f16* __restrict__ W,
                                                                  const float* __restrict__ bias, void* __restrict__ out_, int M,
                                                                  int N, int K, int ksplit, size_t part_stride) {
  static_assert(EPI == EPI_RELU_F16 || EPI == EPI_BIAS_F16, "bias / relu outputs");
  static_assert(!SLAB || EPI == EPI_BIAS_F16, "slabs carry partial sums: no activation");
  using S = VmShape<MI>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, kg = lane >> 4;
  if constexpr (SLAB) fp16_saturate_on();  // fp16 partial sums saturate instead of overflowing to inf

  // XCD x (blocks b = x mod 8) owns a contiguous range of the (tn, kz) panel space and all row tiles of each panel: a W panel
  // part is fetched from HBM once per XCD and hit by the other row tiles in that XCD's L2
  const int ntm = M / S::ROWS, ntn = N / 256;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = logical % ntm, panel = logical / ntm;
  const int tn = panel % ntn, kz = panel / ntn;
  const int kblocks = K / 32, nt = kblocks / ksplit, kb0 = kz * nt;

  const unsigned lds0 = (unsigned)(size_t)smem;
  const int t_sw = (kg ^ tm_swz(l15)) << 4;
  const unsigned xa0 = lds0 + (wr * 16 * MI + l15) * 64 + t_sw;
  const unsigned wa0 = lds0 + S::XB + (wc * 128 + l15) * 64 + t_sw;
  const unsigned long long hmask = (wave & 1) ? 0xffffffff00000000ull : 0x00000000ffffffffull;

  // X pieces of this wave (16 rows x 64 B = 1 KiB each; piece p of the unit = global row r0 + 16 p; a piece never crosses a
  // 256-row block of the tile-major image): MI 4: 2w, 2w+1; MI 5: 2w, 2w+1, half of 8 + (w >> 1); MI 6: 3w, 3w+1, 3w+2
  const int r0 = tm * S::ROWS;
  const int pa = MI == 6 ? 3 * wave : 2 * wave, pb = pa + 1, ph = MI == 6 ? 3 * wave + 2 : 8 + (wave >> 1);
  auto xpiece = [&](int p) {
    const int r = r0 + 16 * p;
    return (const char*)X + ((size_t)(r >> 8) * kblocks + kb0) * (TM_BLOCK * 2) + (r & 255) * 64 + lane * 16;
  };
  VmStream st;
  st.ga = xpiece(pa);
  st.gb = xpiece(pb) - 1024;
  st.gh = MI == 4 ? st.ga : xpiece(ph);
  st.gw = (const char*)W + ((size_t)tn * kblocks + kb0) * (TM_BLOCK * 2) + wave * 4096 + lane * 16;
  st.inc = TM_BLOCK * 2;
  auto dst_of = [&](int slot) {
    const unsigned b = lds0 + slot * S::SLOT;
    return VmDst{b + pa * 1024, b + ph * 1024, b + S::XB + wave * 4096};
  };

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
  for (int s = 0; s < S::LEAD; ++s) vm_fill<MI>(st, dst_of(s), hmask);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"((S::LEAD - 1) * S::DMA) : "memory");
  VmFrag<MI> f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.x[0][mi]) : "v"(xa0), "n"(mi * 1024) : "memory");
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(f.w[0][0]), "=&v"(f.w[0][1])
               : "v"(wa0)
               : "memory");
  // the X reads above were issued without a wait of their own; the lgkmcnt(0) of the W statement covers them, and these empty
  // statements keep every compiler use of their destinations behind it
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(f.x[0][mi]));
  __builtin_amdgcn_sched_barrier(0);

  // ---- K loop: step s consumes slice s (slot s % NSLOT), reads slice s + 1, issues slice s + LEAD into the slot of slice s - 1.
  // At the top of step s slices s+1 .. s+LEAD-1 are in flight; the counted wait leaves the LEAD - 2 younger ones.  Past the
  // unit's last slice the cursor stays on it (inc = 0): the re-fetch lands in a slot nobody reads again and keeps the count uniform.
  int cs = 0;  // slot of the current slice
  auto step = [&](auto zero_tag, auto xb_tag, int s) {
    constexpr bool ZERO = decltype(zero_tag)::value;
    constexpr int XB = decltype(xb_tag)::value;
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"((S::LEAD - 2) * S::DMA) : "memory");
    if (s + S::LEAD == nt) {  // the cursor has run past the unit's last slice: stay on it
      st.inc = -st.inc;
      st.advance();
      st.inc = 0;
    }
    const int ns = cs + 1 == S::NSLOT ? 0 : cs + 1;
    const int ds = cs == 0 ? S::NSLOT - 1 : cs - 1;
    vm_step<MI, ZERO, XB>(f, st, xa0 + ns * S::SLOT, wa0 + cs * S::SLOT, wa0 + ns * S::SLOT, dst_of(ds), hmask);
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
  char* rp[MI];  // this lane's chunk of k-block 0 in row (mi, l15)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = r0 + wr * 16 * MI + mi * 16 + l15;
    if constexpr (!SLAB)
      // tile-major [M][N]: row m sits in block (m >> 8, k-block) at (m & 255) * 64 B
      rp[mi] = (char*)out_ + ((size_t)(m >> 8) * (N >> 5) + (size_t)tn * 8 + wc * 4) * (TM_BLOCK * 2) + (m & 255) * 64 +
               ((cidx ^ tm_swz(l15)) << 4);
    else
      rp[mi] = (char*)out_ + (size_t)kz * part_stride + ((size_t)m * N + tn * 256 + wc * 128 + cidx * 8) * 2;
  }
  constexpr size_t JSTEP = !SLAB ? (size_t)TM_BLOCK * 2 : 64;  // next 32-column k-block
#define SMI_VM_CHUNK(MIV, J, MI_)                                                                                    \
  if constexpr ((MI_) < MI) {                                                                                        \
    f32x4 va, vb;                                                                                                    \
    SMI_VM_RDOUT(MIV, J, 0, MI_, va);                                                                                \
    SMI_VM_RDOUT(MIV, J, 1, MI_, vb);                                                                                \
    va = va + bv[2 * (J)];                                                                                           \
    vb = vb + bv[2 * (J) + 1];                                                                                       \
    const uint2 ha = __builtin_bit_cast(uint2, epi_act_pack<EPI>(va)), hb = __builtin_bit_cast(uint2, epi_act_pack<EPI>(vb)); \
    const auto s0 = __builtin_amdgcn_permlane16_swap(ha.x, hb.x, false, false);                                      \
    const auto s1 = __builtin_amdgcn_permlane16_swap(ha.y, hb.y, false, false);                                      \
    const u32x4 chunk = {s0[0], s1[0], s0[1], s1[1]};                                                                \
    u32x4* dst = (u32x4*)(rp[MI_] + (J) * JSTEP);                                                                    \
    if constexpr (!SLAB)                                                                                             \
      store_nt(dst, chunk);                                                                                          \
    else                                                                                                             \
      *dst = chunk;                                                                                                  \
  }
#define SMI_VM_KB4(J) SMI_VM_CHUNK(4, J, 0) SMI_VM_CHUNK(4, J, 1) SMI_VM_CHUNK(4, J, 2) SMI_VM_CHUNK(4, J, 3)
#define SMI_VM_KB5(J) SMI_VM_CHUNK(5, J, 0) SMI_VM_CHUNK(5, J, 1) SMI_VM_CHUNK(5, J, 2) SMI_VM_CHUNK(5, J, 3) SMI_VM_CHUNK(5, J, 4)
#define SMI_VM_KB6(J) \
  SMI_VM_CHUNK(6, J, 0) SMI_VM_CHUNK(6, J, 1) SMI_VM_CHUNK(6, J, 2) SMI_VM_CHUNK(6, J, 3) SMI_VM_CHUNK(6, J, 4) SMI_VM_CHUNK(6, J, 5)
  if constexpr (MI == 4) {
    SMI_VM_KB4(0) SMI_VM_KB4(1) SMI_VM_KB4(2) SMI_VM_KB4(3)
  } else if constexpr (MI == 5) {
    SMI_VM_KB5(0) SMI_VM_KB5(1) SMI_VM_KB5(2) SMI_VM_KB5(3)
  } else {
    SMI_VM_KB6(0) SMI_VM_KB6(1) SMI_VM_KB6(2) SMI_VM_KB6(3)
  }
#undef SMI_VM_KB4
#undef SMI_VM_KB5
#undef SMI_VM_KB6
#undef SMI_VM_CHUNK
}

// Rows per unit (128 / 160 / 192) for a launch, or 0: the tallest of the three heights that divides M, keeps every unit on a CU of
// its own and puts MORE units on the chip than 256-row tiles would (M a multiple of 256: the tile-major image has 256-row blocks).
static int lone_rows(int M, int N, int K, int ksplit) {
  const int mode = tune(TUNE_DEC_M160, 1);
  if (mode == 0) return 0;
  if (M % 256 || N % 256 || K % 32 || ksplit < 1 || (K / 32) % ksplit) return 0;
  const int nt = K / 32 / ksplit;
  // units with a K loop of >= 32 slices (K >= 1024 per unit): the attention-output projection's 16-slice units measured 18-19 %
  // SLOWER than the k-sliced 64x64 units they would replace (tools/probe_lone.py, profiles/r06r_probe_lone.log); DEC_M160=2 (tests)
  // takes every K loop the ring can run
  if (nt < (mode == 2 ? 8 : 32) || nt % 2) return 0;
  const int64_t units256 = (int64_t)(M / 256) * (N / 256) * ksplit;
  int best = 0;
  int64_t best_units = units256;
  for (int rows : {192, 160, 128}) {
    if (M % rows) continue;
    const int64_t units = (int64_t)(M / rows) * (N / 256) * ksplit;
    // at least half the chip: below that the k-sliced 64x64 units (gemm_lone16.hpp: M = 256, a batch of 5) are faster
    if (units <= num_cus() && units >= (mode == 2 ? 1 : num_cus() / 2) && units > best_units) {
      best = rows;
      best_units = units;
    }
  }
  return best;
}

bool gemm_v2_lone_fits(int M, int N, int K, int ksplit) { return lone_rows(M, N, K, ksplit) != 0; }

template <int EPI, int MI, bool SLAB>
static hipError_t launch_lone_unit(const f16* X, const f16* W, const float* bias, void* out, int M, int N, int K, int ksplit,
                                   size_t part_stride, hipStream_t stream) {
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_v2_lone_kernel<EPI, MI, SLAB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       VmShape<MI>::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done.set();
  }
  const int grid = (M / VmShape<MI>::ROWS) * (N / 256) * ksplit;
  hipLaunchKernelGGL((gemm_v2_lone_kernel<EPI, MI, SLAB>), dim3(grid), dim3(V2_THREADS), VmShape<MI>::LDS_BYTES, stream, X, W, bias, out,
                     M, N, K, ksplit, part_stride);
  return hipGetLastError();
}

// mode 0: row-major fp16 slabs [ksplit][M][N]; 1: tile-major fp16 relu output [M][N]; 2: tile-major fp16 bias output (ksplit 1)
hipError_t launch_gemm_v2_lone(int mode, const f16* X, const f16* W, const float* bias, void* out, int M, int N, int K, int ksplit,
                               hipStream_t stream) {
  const int rows = lone_rows(M, N, K, ksplit);
  if (!rows || (mode && ksplit != 1)) return hipErrorInvalidValue;
  const size_t ps = (size_t)M * N * 2;
#define SMI_VM_LAUNCH(MIV)                                                                                                \
  return mode == 1   ? launch_lone_unit<EPI_RELU_F16, MIV, false>(X, W, bias, out, M, N, K, 1, 0, stream)                 \
         : mode == 2 ? launch_lone_unit<EPI_BIAS_F16, MIV, false>(X, W, bias, out, M, N, K, 1, 0, stream)                 \
                     : launch_lone_unit<EPI_BIAS_F16, MIV, true>(X, W, bias, out, M, N, K, ksplit, ps, stream)
  if (rows == 128) SMI_VM_LAUNCH(4);
  if (rows == 160) SMI_VM_LAUNCH(5);
  SMI_VM_LAUNCH(6);
#undef SMI_VM_LAUNCH
}

}  // namespace smi
