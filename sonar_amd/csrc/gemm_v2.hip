// Tile-major fp16 GEMMs on the second-generation 256x256 engine (gemm_v2.hpp): C = act(rstd * (X . W^T) [- rstd * mean * c1] + c2),
// X [M, K] and W [N, K] tile-major fp16, C [M, N] tile-major fp16 (the X operand of the next GEMM).  These are the text
// encoder's fused QKV and FFN-inner projections (reference wiring: sonar/models/sonar_text/factory.py:130-153), the
// conformer's FFN-inner projections and every other <bias | relu | silu, tile-major in/out> launch with >= 24 K slices.
//
// Persistent: one workgroup of 4 waves per CU walks tiles in the raster of the 8-wave engine (gemm.hip).  A tile's life:
//   step 0          its per-tile constants (c2 / bias, c1, the rows' LayerNorm partial sums) are fetched by LDS-DMA into
//                   the wave's private LDS area; the accumulators start from the literal 0
//   last 3 steps    the operand stream moves on to the next tile (the ring never drains)
//   read-out        accumulators -> (LayerNorm fold) -> activation -> fp16 -> lane swaps -> 32 stores of 1 KiB per wave,
//                   each issued as soon as its chunk exists: the stores use the CU's memory path while the matrix pipe
//                   is idle anyway.  (Round 6, experiments 1-3, profiles/r06_experiments.txt: holding the finished tile
//                   in 128 VGPRs and issuing 2 stores per K step of the next tile does NOT hide them -- the K loop pulls
//                   32 KiB per 0.7 us through the CU's memory path, 89 % of the 52 GB/s per CU the L2s deliver, and every
//                   store byte under the loop slows the loop by its own transfer time: -2.6 us of stores, +2.6 us of loop.)
#include <algorithm>
#include <type_traits>

#include "gemm_epi.hpp"
#include "gemm_v2.hpp"
#include "kernels.hpp"

// -DV2_PROBE=<bits> (measurement builds, wrong results): 1 = no stores under the K loop, 2 = no read-out, 4 = plain (not
// non-temporal) stores
#ifndef V2_PROBE
#define V2_PROBE 0
#endif

namespace smi {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: out = act(acc + c2)  (c2 = the bias);  1: LayerNorm fold with the exact mean term;  2: fold with centred weights
template <int EPI, bool FOLD>
__global__ __launch_bounds__(V2_THREADS) void gemm_v2_kernel(const f16* __restrict__ X, const f16* __restrict__ W,
                                                             const float* __restrict__ c2, f16* __restrict__ out, int M, int N,
                                                             int K, int raster, GemmLnFold fold) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, kg = lane >> 4;
  const V2Ring rg = v2_make_ring(smem, wave, lane);
  const unsigned voff = wave * 4096 + lane * 16;

  // ---- tile walk: the rasters of gemm_tn256_kernel (gemm.hip) ----
  const int ntm = M / 256, ntn = N / 256, nt = K / 32, nout = ntm * ntn;
  const int nq = ntn / 4;
  const int nvirt = raster ? ((ntm + 63) / 64) * nq * 256 : nout;
  auto coords = [&](int t, int& tm_, int& tn_) -> bool {
    if (raster == 0) {
      constexpr int GM = 8;  // grouped 8(m) x ntn super-tiles in id order (gemm_tile256.hpp: g2_tile_coords_of)
      const int per_group = GM * ntn, group = t / per_group, first_m = group * GM;
      const int gsz = min(GM, ntm - first_m), in_group = t - group * per_group;
      tm_ = first_m + in_group % gsz;
      tn_ = in_group / gsz;
      return true;
    }
    const int q = t / 256, c = (t % 256) / 32, j = t % 32;
    tm_ = (c + 8 * (q / nq)) * 8 + j % 8;
    tn_ = ((q + (raster == 2 ? c : 0)) % nq) * 4 + j / 8;
    return tm_ < ntm;
  };
  int tile_m = 0, tile_n = 0;
  auto seek = [&](int t) {
    while (t < nvirt && !coords(t, tile_m, tile_n)) t += gridDim.x;
    return t;
  };
  int tile = seek(xcd_remap(blockIdx.x, gridDim.x));
  if (tile >= nvirt) return;

  const size_t panel = (size_t)nt * (TM_BLOCK * 2);  // bytes of one 256-row operand panel
  V2Stream st;
  st.xp = (const char*)X + (size_t)tile_m * panel + voff;
  st.wp = (const char*)W + (size_t)tile_n * panel + voff;
  st.inc = TM_BLOCK * 2;
  V2Frag f;
  v2_start(f, st, rg);

  // ---- per-wave constant area: c2[128] | c1[128] | partial sums p = 0..3: float2[128 rows] ----
  const unsigned cbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + V2_RING_BYTES + wave * V2_CONST_BYTES);
  auto fetch_consts = [&](int m0, int n0) {
    const float* c2p = c2 + n0 + wc * 128 + lane;
    if constexpr (FOLD) {
      const float* c1p = fold.c1 + n0 + wc * 128 + lane;
      // partial p of the wave's 128 rows: 1 KiB contiguous; the instruction offset applies to the LDS side too, so the
      // source pointers are pre-biased by -p KiB
      const char* rp[4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
        rp[p] = (const char*)(fold.part_in + (size_t)(p < fold.nparts ? p : 0) * M + m0 + wr * 128) + lane * 16 - p * 1024;
      asm volatile(
          "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
          "global_load_lds_dword %0, off\n\tglobal_load_lds_dword %0, off offset:256\n\t"
          "s_mov_b32 m0, %7\n\ts_nop 0\n\t"
          "global_load_lds_dword %1, off\n\tglobal_load_lds_dword %1, off offset:256\n\t"
          "s_mov_b32 m0, %8\n\ts_nop 0\n\t"
          "global_load_lds_dwordx4 %2, off\n\tglobal_load_lds_dwordx4 %3, off offset:1024\n\t"
          "global_load_lds_dwordx4 %4, off offset:2048\n\tglobal_load_lds_dwordx4 %5, off offset:3072"
          :
          : "v"(c2p), "v"(c1p), "v"(rp[0]), "v"(rp[1]), "v"(rp[2]), "v"(rp[3]), "s"(cbase), "s"(cbase + 512), "s"(cbase + 1024)
          : "memory");
    } else {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\tglobal_load_lds_dword %0, off offset:256"
                   :
                   : "v"(c2p), "s"(cbase)
                   : "memory");
    }
  };
  constexpr int NCONST = FOLD ? 8 : 2;  // vector-memory operations of fetch_consts per wave

  // store address of chunk (j, mi) = k-block j of the wave's 4, 16-row block mi: wave-uniform tile base + lane part +
  // j * 16 KiB + mi * 1 KiB
  const int cidx = (kg & 1) * 2 + (kg >> 1);
  const unsigned lane_part = (unsigned)(((wr * 128 + l15) * 32 + ((cidx ^ tm_swz(l15)) << 3)) * 2);
  // EPI_GLU_F16 (the conformer's pointwise_conv1 + GLU; out_h[m][g * 32 + c] = f16(a * sigmoid(b)), a / b = columns g * 64 + c /
  // g * 64 + 32 + c, gemm.hip): the output has N / 2 columns, a wave's 128 columns are two [32 values | 32 gates] groups = two
  // 32-column blocks of the tile-major output, 16 chunks per lane and tile
  constexpr bool GLU = EPI == EPI_GLU_F16;
  auto tile_out = [&](int tm_, int tn_) {
    if constexpr (GLU) return (char*)out + ((size_t)tm_ * (N >> 6) + (size_t)tn_ * 4 + wc * 2) * (TM_BLOCK * 2);
    return (char*)out + ((size_t)tm_ * (N >> 5) + (size_t)tn_ * 8 + wc * 4) * (TM_BLOCK * 2);
  };
  constexpr int NST = (V2_PROBE & 3) ? 0 : (GLU ? 16 : 32);  // stores per wave and tile

  bool more = true, first_tile = true;
  while (more) {
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    // ---- steps 0..3.  In the wave's in-order queue the previous tile's NST stores sit behind the DMA of slices 1 and 2
    // (issued by the previous tile's last two steps) and in front of slice 3 (issued by step 0): the counted waits of steps
    // 0 and 1 let them stay in flight, step 2 (slice 3) retires them -- 2 steps + the read-out after the first was issued.
    if (first_tile) {
      v2_step_top<8>();
      fetch_consts(m0, n0);
      v2_step_body<0, true>(f, st, rg);
      v2_step_top<8 + NCONST>();
    } else {
      v2_step_top<8 + NST>();
      fetch_consts(m0, n0);
      v2_step_body<0, true>(f, st, rg);
      v2_step_top<8 + NCONST + NST>();
    }
    first_tile = false;
    v2_step_body<1, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<2, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<3, false>(f, st, rg);
    for (int kb = 4; kb < nt - 4; kb += 4) {
      v2_step_top<8>();
      v2_step_body<0, false>(f, st, rg);
      v2_step_top<8>();
      v2_step_body<1, false>(f, st, rg);
      v2_step_top<8>();
      v2_step_body<2, false>(f, st, rg);
      v2_step_top<8>();
      v2_step_body<3, false>(f, st, rg);
    }
    // ---- the last four steps: step nt-4 issues this tile's last slice, the other three the next tile's slices 0..2 ----
    v2_step_top<8>();
    v2_step_body<0, false>(f, st, rg);
    const int tm_cur = tile_m, tn_cur = tile_n;
    tile = seek(tile + (int)gridDim.x);
    more = tile < nvirt;
    if (more) {
      unsigned vo = voff;  // opaque: keeps hipcc from holding X + voff / W + voff in four VGPRs across the loop (it spilled them)
      asm volatile("" : "+v"(vo));
      st.xp = (const char*)X + (size_t)tile_m * panel + vo;
      st.wp = (const char*)W + (size_t)tile_n * panel + vo;
    } else {
      st.xp -= st.inc;
      st.wp -= st.inc;
      st.inc = 0;
    }
    v2_step_top<8>();
    v2_step_body<1, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<2, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<3, false>(f, st, rg);

    // ---- read-out: accumulators -> 32 stores ----
    char* const obase = tile_out(tm_cur, tn_cur);
    // the lane's share of the tile constants (kept spread over the wave, fetched with ds_bpermute below): (sum, sum of
    // squares) partials of rows wr*128 + lane and + 64 + lane, c2 / c1 of columns wc*128 + lane and + 64 + lane.  They
    // landed long ago (the waits of steps 2.. retired them: in-order queue); asm loads, issue + wait in one statement.
    float c2r[2], c1r[2] = {0.f, 0.f};
    float row_rs[2] = {1.f, 1.f}, row_nm[2] = {0.f, 0.f};
    const unsigned ca = cbase + lane * 4, ra = cbase + 1024 + lane * 8;
    if constexpr (FOLD) {
      float2 pr[2][4];
      asm volatile(
          "ds_read_b64 %0, %12\n\tds_read_b64 %1, %12 offset:1024\n\tds_read_b64 %2, %12 offset:2048\n\t"
          "ds_read_b64 %3, %12 offset:3072\n\tds_read_b64 %4, %12 offset:512\n\tds_read_b64 %5, %12 offset:1536\n\t"
          "ds_read_b64 %6, %12 offset:2560\n\tds_read_b64 %7, %12 offset:3584\n\t"
          "ds_read_b32 %8, %13\n\tds_read_b32 %9, %13 offset:256\n\tds_read_b32 %10, %13 offset:512\n\t"
          "ds_read_b32 %11, %13 offset:768\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(pr[0][0]), "=&v"(pr[0][1]), "=&v"(pr[0][2]), "=&v"(pr[0][3]), "=&v"(pr[1][0]), "=&v"(pr[1][1]),
            "=&v"(pr[1][2]), "=&v"(pr[1][3]), "=&v"(c2r[0]), "=&v"(c2r[1]), "=&v"(c1r[0]), "=&v"(c1r[1])
          : "v"(ra), "v"(ca)
          : "memory");
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float2 z = {0.f, 0.f};
        // the summation order of the 8-wave engine: (p0 + p2) + (p1 + p3)
        const float2 p0 = pr[u][0], p1 = fold.nparts > 1 ? pr[u][1] : z, p2 = fold.nparts > 2 ? pr[u][2] : z,
                     p3 = fold.nparts > 3 ? pr[u][3] : z;
        const float sx = (p0.x + p2.x) + (p1.x + p3.x), sy = (p0.y + p2.y) + (p1.y + p3.y);
        const float mean = sx * fold.inv_k;
        const float var = fmaxf(sy * fold.inv_k - mean * mean, 0.f);
        row_rs[u] = __builtin_amdgcn_rsqf(var + fold.eps);
        row_nm[u] = -row_rs[u] * mean;
      }
    } else {
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(c2r[0]), "=&v"(c2r[1])
                   : "v"(ca)
                   : "memory");
    }
    asm volatile(V2_RDOUT_FIRST_STR);
    auto put = [&](int i, const u32x4& chunk) {  // chunk i = (j, mi)
      u32x4* dst = (u32x4*)(obase + (size_t)(i >> 3) * (TM_BLOCK * 2) + (i & 7) * 1024 + lane_part);
      if constexpr ((V2_PROBE & 1) != 0)
        asm volatile("" ::"v"(chunk), "v"(dst));
      else if constexpr ((V2_PROBE & 4) != 0)
        *dst = chunk;
      else
        store_nt(dst, chunk);
    };
    auto readout = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      float rsall[8], nmall[8];
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {  // row mi*16 + l15 of the wave's 128: lane (mi*16 + l15) & 63, register mi >> 2
        rsall[mi] = 1.f;
        nmall[mi] = 0.f;
        if constexpr (MODE != 0) {
          const int src = ((mi & 3) * 16 + l15) * 4;
          rsall[mi] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(row_rs[mi >> 2])));
          if constexpr (MODE == 1) nmall[mi] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(row_nm[mi >> 2])));
        }
      }
      if constexpr (GLU) {
        // the fold / bias affine of one accumulator block (the LayerNorm in front of this GEMM), no activation
        auto affine = [&](f32x4 v, const f32x4& c2q, const f32x4& c1q, float rs, float nm) {
          const f32x2 rs2 = {rs, rs}, nm2 = {nm, nm};
#pragma unroll
          for (int hp2 = 0; hp2 < 2; ++hp2) {
            const f32x2 c2p = {c2q[2 * hp2], c2q[2 * hp2 + 1]};
            f32x2 vp = {v[2 * hp2], v[2 * hp2 + 1]};
            if constexpr (MODE == 1) {
              const f32x2 c1p = {c1q[2 * hp2], c1q[2 * hp2 + 1]};
              vp = __builtin_elementwise_fma(rs2, vp, __builtin_elementwise_fma(nm2, c1p, c2p));
            } else if constexpr (MODE == 2) {
              vp = __builtin_elementwise_fma(rs2, vp, c2p);
            } else {
              vp = vp + c2p;
            }
            v[2 * hp2] = vp[0];
            v[2 * hp2 + 1] = vp[1];
          }
          return v;
        };
#define SMI_V2_GLU(G, JA, JG, MI)                                                                                       \
  {                                                                                                                    \
    f32x4 a0, a1, g0, g1;                                                                                              \
    SMI_V2_RDOUT_IDX(JA, 0, MI, a0);                                                                                   \
    SMI_V2_RDOUT_IDX(JA, 1, MI, a1);                                                                                   \
    SMI_V2_RDOUT_IDX(JG, 0, MI, g0);                                                                                   \
    SMI_V2_RDOUT_IDX(JG, 1, MI, g1);                                                                                   \
    a0 = affine(a0, c2q[0], c1q[0], rsall[MI], nmall[MI]);                                                             \
    a1 = affine(a1, c2q[1], c1q[1], rsall[MI], nmall[MI]);                                                             \
    g0 = affine(g0, c2q[2], c1q[2], rsall[MI], nmall[MI]);                                                             \
    g1 = affine(g1, c2q[3], c1q[3], rsall[MI], nmall[MI]);                                                             \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                    \
      a0[e] *= sigmoid_f(g0[e]);                                                                                       \
      a1[e] *= sigmoid_f(g1[e]);                                                                                       \
    }                                                                                                                  \
    const uint2 h0 = __builtin_bit_cast(uint2, epi_act_pack<EPI_BIAS_F16>(a0));                                        \
    const uint2 h1 = __builtin_bit_cast(uint2, epi_act_pack<EPI_BIAS_F16>(a1));                                        \
    const auto s0 = __builtin_amdgcn_permlane16_swap(h0.x, h1.x, false, false);                                        \
    const auto s1 = __builtin_amdgcn_permlane16_swap(h0.y, h1.y, false, false);                                        \
    put((G) * 8 + (MI), u32x4{s0[0], s1[0], s0[1], s1[1]});                                                            \
  }
#define SMI_V2_GLU_GROUP(G, JA, JG)                                                                                    \
  {                                                                                                                    \
    f32x4 c2q[4], c1q[4]; /* blocks ni = 4 G + q: values q = 0, 1, gates q = 2, 3 */                                   \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int r = 0; r < 4; ++r) {                      \
      const int col = (4 * (G) + q) * 16 + 4 * kg + r; /* of the wave's 128: lane col & 63, register col >> 6 */        \
      c2q[q][r] = __int_as_float(__builtin_amdgcn_ds_bpermute((col & 63) * 4, __float_as_int(c2r[(4 * (G) + q) >> 2]))); \
      c1q[q][r] = 0.f;                                                                                                 \
      if constexpr (MODE == 1)                                                                                         \
        c1q[q][r] = __int_as_float(__builtin_amdgcn_ds_bpermute((col & 63) * 4, __float_as_int(c1r[(4 * (G) + q) >> 2]))); \
    }                                                                                                                  \
    SMI_V2_GLU(G, JA, JG, 0) SMI_V2_GLU(G, JA, JG, 1) SMI_V2_GLU(G, JA, JG, 2) SMI_V2_GLU(G, JA, JG, 3)                \
    SMI_V2_GLU(G, JA, JG, 4) SMI_V2_GLU(G, JA, JG, 5) SMI_V2_GLU(G, JA, JG, 6) SMI_V2_GLU(G, JA, JG, 7)                \
  }
        SMI_V2_GLU_GROUP(0, 0, 1) SMI_V2_GLU_GROUP(1, 2, 3)
#undef SMI_V2_GLU_GROUP
#undef SMI_V2_GLU
        return;
      }
#define SMI_V2_PAIR(J, MI)                                                                                              \
  {                                                                                                                    \
    f32x4 va, vb;                                                                                                      \
    SMI_V2_RDOUT_IDX(J, 0, MI, va);                                                                                    \
    SMI_V2_RDOUT_IDX(J, 1, MI, vb);                                                                                    \
    put((J) * 8 + (MI), finish(va, vb, c2v, c1v, rsall[MI], nmall[MI]));                                               \
  }
      auto finish = [&](f32x4 va, f32x4 vb, const f32x4 (&c2v)[2], const f32x4 (&c1v)[2], float rs, float nm) {
        uint32_t h[2][2];
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
          f32x4 v = nl ? vb : va;
          const f32x2 rs2 = {rs, rs}, nm2 = {nm, nm};
#pragma unroll
          for (int hp2 = 0; hp2 < 2; ++hp2) {
            const f32x2 c2p = {c2v[nl][2 * hp2], c2v[nl][2 * hp2 + 1]};
            f32x2 vp = {v[2 * hp2], v[2 * hp2 + 1]};
            if constexpr (MODE == 1) {
              const f32x2 c1p = {c1v[nl][2 * hp2], c1v[nl][2 * hp2 + 1]};
              vp = __builtin_elementwise_fma(rs2, vp, __builtin_elementwise_fma(nm2, c1p, c2p));
            } else if constexpr (MODE == 2) {
              vp = __builtin_elementwise_fma(rs2, vp, c2p);
            } else {
              vp = vp + c2p;
            }
            v[2 * hp2] = vp[0];
            v[2 * hp2 + 1] = vp[1];
          }
          const uint2 hp = __builtin_bit_cast(uint2, epi_act_pack<EPI>(v));
          h[nl][0] = hp.x;
          h[nl][1] = hp.y;
        }
        // rows 16..31 / 48..63 of h[0] <-> rows 0..15 / 32..47 of h[1]: every lane then holds one whole 16-B chunk
        // (lane group kg owns chunk (kg&1)*2 + (kg>>1) of the 32-column k-block; gemm.hip, LAYOUT 2)
        const auto s0 = __builtin_amdgcn_permlane16_swap(h[0][0], h[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(h[0][1], h[1][1], false, false);
        return u32x4{s0[0], s1[0], s0[1], s1[1]};
      };
#define SMI_V2_KBLOCK(J)                                                                                               \
  {                                                                                                                    \
    f32x4 c2v[2], c1v[2];                                                                                              \
    _Pragma("unroll") for (int nl = 0; nl < 2; ++nl) _Pragma("unroll") for (int r = 0; r < 4; ++r) {                   \
      const int col = (2 * (J) + nl) * 16 + 4 * kg + r; /* of the wave's 128: lane col & 63, register col >> 6 */      \
      c2v[nl][r] = __int_as_float(__builtin_amdgcn_ds_bpermute((col & 63) * 4, __float_as_int(c2r[(2 * (J) + nl) >> 2]))); \
      c1v[nl][r] = 0.f;                                                                                                \
      if constexpr (MODE == 1)                                                                                         \
        c1v[nl][r] = __int_as_float(__builtin_amdgcn_ds_bpermute((col & 63) * 4, __float_as_int(c1r[(2 * (J) + nl) >> 2]))); \
    }                                                                                                                  \
    SMI_V2_PAIR(J, 0) SMI_V2_PAIR(J, 1) SMI_V2_PAIR(J, 2) SMI_V2_PAIR(J, 3) SMI_V2_PAIR(J, 4) SMI_V2_PAIR(J, 5)        \
    SMI_V2_PAIR(J, 6) SMI_V2_PAIR(J, 7)                                                                                \
  }
      SMI_V2_KBLOCK(0) SMI_V2_KBLOCK(1) SMI_V2_KBLOCK(2) SMI_V2_KBLOCK(3)
#undef SMI_V2_KBLOCK
#undef SMI_V2_PAIR
    };
    if constexpr ((V2_PROBE & 2) != 0) {
    } else if constexpr (FOLD) {
      if (fold.centered)
        readout(std::integral_constant<int, 2>{});
      else
        readout(std::integral_constant<int, 1>{});
    } else {
      readout(std::integral_constant<int, 0>{});
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The fp16 tile-major RESIDUAL STREAM on the same engine: x[m][n] = f16(float(x[m][n]) + RSTEP * (acc + bias[n])), read-modify-
// write in place (EPI_RESID_F16, RSTEP = 1: attention output / FFN output projections; EPI_RESID_HALF_F16, 0.5: the conformer's
// macaron FFNs), optionally leaving the LayerNorm-fold partial sums of the rows it has just written (EMIT; GemmLnFold
// producer: part_out[N/256][M] (sum, sum of squares) over the tile's 256 columns of the ROUNDED new values).
//   * the 32 old chunks of a lane (its piece of the old tile) are requested by asm loads the compiler does not see (it would
//     drain the LDS-DMA queue in front of the first use): k-blocks 0..2 in the tile's last three K steps, 8 loads per step,
//     k-block 3 at the start of the read-out, where its latency hides behind the arithmetic of k-blocks 0..2.  Vector-memory
//     operations retire in issue order, so two counted waits in the read-out cover them (the first one also retires slice 1
//     of the next tile, which step 0 needs anyway).  Nothing of the old tile's latency is exposed.
//   * a wave covers 128 of the tile's 256 columns: the two column waves of a row half leave their row sums in LDS during
//     the read-out and wave u adds and stores rows 64u .. 64u+63 behind the barrier of the next K step.
template <int EPI, bool EMIT>
__global__ __launch_bounds__(V2_THREADS) void gemm_v2_resid_kernel(const f16* __restrict__ X, const f16* __restrict__ W,
                                                                   const float* __restrict__ bias, f16* __restrict__ out, int M,
                                                                   int N, int K, int raster, float2* __restrict__ part_out) {
  static_assert(EPI == EPI_RESID_F16 || EPI == EPI_RESID_HALF_F16, "fp16 residual stream epilogues");
  constexpr float RSTEP = EPI == EPI_RESID_HALF_F16 ? 0.5f : 1.0f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, kg = lane >> 4;
  const V2Ring rg = v2_make_ring(smem, wave, lane);
  const unsigned voff = wave * 4096 + lane * 16;

  const int ntm = M / 256, ntn = N / 256, nt = K / 32, nout = ntm * ntn;
  const int nq = ntn / 4;
  const int nvirt = raster ? ((ntm + 63) / 64) * nq * 256 : nout;
  auto coords = [&](int t, int& tm_, int& tn_) -> bool {
    if (raster == 0) {
      constexpr int GM = 8;
      const int per_group = GM * ntn, group = t / per_group, first_m = group * GM;
      const int gsz = min(GM, ntm - first_m), in_group = t - group * per_group;
      tm_ = first_m + in_group % gsz;
      tn_ = in_group / gsz;
      return true;
    }
    const int q = t / 256, c = (t % 256) / 32, j = t % 32;
    tm_ = (c + 8 * (q / nq)) * 8 + j % 8;
    tn_ = ((q + (raster == 2 ? c : 0)) % nq) * 4 + j / 8;
    return tm_ < ntm;
  };
  int tile_m = 0, tile_n = 0;
  auto seek = [&](int t) {
    while (t < nvirt && !coords(t, tile_m, tile_n)) t += gridDim.x;
    return t;
  };
  int tile = seek(xcd_remap(blockIdx.x, gridDim.x));
  if (tile >= nvirt) return;

  const size_t panel = (size_t)nt * (TM_BLOCK * 2);
  V2Stream st;
  st.xp = (const char*)X + (size_t)tile_m * panel + voff;
  st.wp = (const char*)W + (size_t)tile_n * panel + voff;
  st.inc = TM_BLOCK * 2;
  V2Frag f;
  v2_start(f, st, rg);

  // per-wave LDS area above the ring: bias[128] at +0; this wave's row sums float2[128] at +6144
  const unsigned cbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + V2_RING_BYTES + wave * V2_CONST_BYTES);
  const unsigned lds_const0 = (unsigned)(size_t)smem + V2_RING_BYTES;
  const bool has_bias = bias != nullptr;
  const float* bias_src = has_bias ? bias : (const float*)W;  // no bias: the same two DMA instructions from a valid address
  auto fetch_consts = [&](int n0) {
    const float* bp = bias_src + (has_bias ? n0 + wc * 128 : 0) + lane;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\tglobal_load_lds_dword %0, off offset:256"
                 :
                 : "v"(bp), "s"(cbase)
                 : "memory");
  };
  constexpr int NCONST = 2;
  constexpr int NEMIT = EMIT ? 1 : 0;
  constexpr int NST = 32;

  const int cidx = (kg & 1) * 2 + (kg >> 1);
  const unsigned lane_part = (unsigned)(((wr * 128 + l15) * 32 + ((cidx ^ tm_swz(l15)) << 3)) * 2);
  auto tile_out = [&](int tm_, int tn_) {
    return (char*)out + ((size_t)tm_ * (N >> 5) + (size_t)tn_ * 8 + wc * 4) * (TM_BLOCK * 2);
  };
  // the row sums of the PREVIOUS tile: wave u adds the two column waves' halves of rows 64u .. 64u+63 and stores them
  int prev_m0 = 0, prev_tn = 0;
  auto emit_rowsums = [&]() {
    if constexpr (EMIT) {
      const int r = wave * 64 + lane;                      // tile row
      const unsigned src = lds_const0 + ((r >> 7) * 2) * V2_CONST_BYTES + 6144 + (r & 127) * 8;
      float2 a, b;
      asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(b)
                   : "v"(src), "n"(V2_CONST_BYTES)
                   : "memory");
      part_out[(size_t)prev_tn * M + prev_m0 + r] = float2{a.x + b.x, a.y + b.y};
    }
  };

  half8 oldv[4][8];  // [k-block j][16-row block mi]
  bool more = true, first_tile = true;
  while (more) {
    const int n0 = tile_n * 256;
    // ---- steps 0..3 (queue: ... DMA slice 1, DMA slice 2, [previous tile: NST stores] | step 0: consts, row sums, DMA slice 3)
    if (first_tile) {
      v2_step_top<8>();
      fetch_consts(n0);
      v2_step_body<0, true>(f, st, rg);
      v2_step_top<8 + NCONST>();
    } else {
      v2_step_top<8 + NST>();
      fetch_consts(n0);
      emit_rowsums();
      v2_step_body<0, true>(f, st, rg);
      v2_step_top<8 + NCONST + NEMIT + NST>();
    }
    first_tile = false;
    v2_step_body<1, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<2, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<3, false>(f, st, rg);
    for (int kb = 4; kb < nt - 4; kb += 4) {
      v2_step_top<8>();
      v2_step_body<0, false>(f, st, rg);
      v2_step_top<8>();
      v2_step_body<1, false>(f, st, rg);
      v2_step_top<8>();
      v2_step_body<2, false>(f, st, rg);
      v2_step_top<8>();
      v2_step_body<3, false>(f, st, rg);
    }
    // ---- the last four steps: the operand stream moves on; the old tile is requested, k-block j in step nt-3+j (8 loads
    // in front of the step's DMA: the next step's counted wait is 8 + 8)
    const int tm_cur = tile_m, tn_cur = tile_n;
    char* const obase = tile_out(tm_cur, tn_cur);
#define SMI_V2_OLD(J)                                                                                                   \
  {                                                                                                                     \
    const char* o0 = obase + (size_t)(J) * (TM_BLOCK * 2) + lane_part;                                                  \
    const char* o1 = o0 + 4096;                                                                                         \
    asm volatile(                                                                                                       \
        "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\t"                           \
        "global_load_dwordx4 %2, %8, off offset:2048\n\tglobal_load_dwordx4 %3, %8, off offset:3072\n\t"               \
        "global_load_dwordx4 %4, %9, off\n\tglobal_load_dwordx4 %5, %9, off offset:1024\n\t"                           \
        "global_load_dwordx4 %6, %9, off offset:2048\n\tglobal_load_dwordx4 %7, %9, off offset:3072"                    \
        : "=&v"(oldv[J][0]), "=&v"(oldv[J][1]), "=&v"(oldv[J][2]), "=&v"(oldv[J][3]), "=&v"(oldv[J][4]),                \
          "=&v"(oldv[J][5]), "=&v"(oldv[J][6]), "=&v"(oldv[J][7])                                                       \
        : "v"(o0), "v"(o1)                                                                                              \
        : "memory");                                                                                                    \
  }
    v2_step_top<8>();
    v2_step_body<0, false>(f, st, rg);
    tile = seek(tile + (int)gridDim.x);
    more = tile < nvirt;
    if (more) {
      unsigned vo = voff;  // opaque: keeps hipcc from holding X + voff / W + voff in four VGPRs across the loop (it spilled them)
      asm volatile("" : "+v"(vo));
      st.xp = (const char*)X + (size_t)tile_m * panel + vo;
      st.wp = (const char*)W + (size_t)tile_n * panel + vo;
    } else {
      st.xp -= st.inc;
      st.wp -= st.inc;
      st.inc = 0;
    }
    v2_step_top<8>();
    SMI_V2_OLD(0)
    v2_step_body<1, false>(f, st, rg);
    v2_step_top<16>();
    SMI_V2_OLD(1)
    v2_step_body<2, false>(f, st, rg);
    v2_step_top<16>();
    SMI_V2_OLD(2)
    v2_step_body<3, false>(f, st, rg);

    // ---- read-out.  k-block 3 of the old tile is requested now (its latency hides behind the read-out of k-blocks 0..2;
    // 96 instead of 128 registers of old data across the last K steps: with 128 hipcc spilled).  In the queue: ..., old 2,
    // DMA x 8 (step nt-1), old 3 x 8 -> the first counted wait (old 0..2 landed) leaves 16 in flight; the second, in front
    // of k-block 3, the 24 stores of k-blocks 0..2.  (The first wait statement names half of its destinations, the second
    // -- which nothing can be moved across -- the other half.)
    SMI_V2_OLD(3)
#undef SMI_V2_OLD
#define SMI_V2_OV(J) "+v"(oldv[J][0]), "+v"(oldv[J][1]), "+v"(oldv[J][2]), "+v"(oldv[J][3]), "+v"(oldv[J][4]), "+v"(oldv[J][5]), "+v"(oldv[J][6]), "+v"(oldv[J][7])
    asm volatile("s_waitcnt vmcnt(16)" : SMI_V2_OV(0), SMI_V2_OV(1)::"memory");
    asm volatile("" : SMI_V2_OV(2)::"memory");
    asm volatile(V2_RDOUT_FIRST_STR);
    float rs_sum[8], rs_sq[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) rs_sum[mi] = rs_sq[mi] = 0.f;
    const unsigned ba = cbase + cidx * 32;  // the lane's 8 bias values of k-block j: columns j*32 + cidx*8 .. +7 of the wave's 128
#define SMI_V2_RCHUNK(J, MI)                                                                                            \
  {                                                                                                                     \
    f32x4 va, vb;                                                                                                       \
    SMI_V2_RDOUT_IDX(J, 0, MI, va);                                                                                     \
    SMI_V2_RDOUT_IDX(J, 1, MI, vb);                                                                                     \
    float c8[8];                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                     \
      const auto sp = __builtin_amdgcn_permlane16_swap(__float_as_uint(va[i]), __float_as_uint(vb[i]), false, false);   \
      c8[i] = __uint_as_float(sp[0]);                                                                                   \
      c8[4 + i] = __uint_as_float(sp[1]);                                                                               \
    }                                                                                                                   \
    half8 o;                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                       \
        o[i] = (f16)__builtin_fmaf(c8[i] + (i < 4 ? b0[i] : b1[i - 4]), RSTEP, (float)oldv[J][MI][i]);                  \
    store_nt((half8*)(obase + (size_t)(J) * (TM_BLOCK * 2) + (MI) * 1024 + lane_part), o);                             \
    if constexpr (EMIT) {                                                                                               \
      const half2v ones = {(f16)1.f, (f16)1.f};                                                                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                   \
        const half2v t2 = {o[2 * i], o[2 * i + 1]};                                                                     \
        rs_sum[MI] = __builtin_amdgcn_fdot2(t2, ones, rs_sum[MI], false);                                               \
        rs_sq[MI] = __builtin_amdgcn_fdot2(t2, t2, rs_sq[MI], false);                                                   \
      }                                                                                                                 \
    }                                                                                                                   \
  }
#define SMI_V2_RKBLOCK(J)                                                                                               \
  {                                                                                                                     \
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};                                                         \
    if (has_bias)                                                                                                       \
      asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"           \
                   : "=&v"(b0), "=&v"(b1)                                                                               \
                   : "v"(ba), "n"((J) * 128), "n"((J) * 128 + 16)                                                       \
                   : "memory");                                                                                         \
    SMI_V2_RCHUNK(J, 0) SMI_V2_RCHUNK(J, 1) SMI_V2_RCHUNK(J, 2) SMI_V2_RCHUNK(J, 3) SMI_V2_RCHUNK(J, 4)                 \
    SMI_V2_RCHUNK(J, 5) SMI_V2_RCHUNK(J, 6) SMI_V2_RCHUNK(J, 7)                                                         \
  }
    SMI_V2_RKBLOCK(0) SMI_V2_RKBLOCK(1) SMI_V2_RKBLOCK(2)
    asm volatile("s_waitcnt vmcnt(24)" : SMI_V2_OV(3)::"memory");
#undef SMI_V2_OV
    SMI_V2_RKBLOCK(3)
#undef SMI_V2_RKBLOCK
#undef SMI_V2_RCHUNK
    if constexpr (EMIT) {
      // join the 4 lane groups of a row (lanes 16 apart); lane group 0 leaves the wave's 128 row sums in LDS
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        float v0 = rs_sum[mi], v1 = rs_sq[mi];
        auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v0), false, false);
        v0 = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
        auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v0), false, false);
        v0 = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
        s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v1), __float_as_uint(v1), false, false);
        v1 = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
        s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v1), __float_as_uint(v1), false, false);
        v1 = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
        if (kg == 0) {
          const float2 pv = {v0, v1};
          asm volatile("ds_write_b64 %0, %1" ::"v"(cbase + 6144 + (mi * 16 + l15) * 8), "v"(pv) : "memory");
        }
      }
      prev_m0 = tm_cur * 256;
      prev_tn = tn_cur;
    }
  }
  if constexpr (EMIT) {  // the last tile's row sums
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    emit_rowsums();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The decoder's tied logits projection on the same engine (beam search of an fp16 model: fp16 tile-major logits, no bias;
// reference: TiedProjection at sonar/models/sonar_text/factory.py:300-315, consumed by the beam search of
// sonar/inference_pipelines/text.py:305-346): out = f16(X W^T), and per (row, 256-column tile) the softmax statistics of the
// ROUNDED values -- tile_max = max_n v * scale, tile_sum = sum_n exp(v * scale - tile_max) over the columns n < valid_n -- so that
// the candidate selection never re-reads the 1 MB logits rows (GemmTileStats, kernels.hpp).  The read-out runs row block by row
// block (all 8 column blocks of a 16-row block: the lane's 32 values of a row), as the 8-wave engine's fused pass does; the two
// column waves of a row half leave (max, sum) of their 128 columns in LDS and wave u combines and stores rows 64u .. 64u+63
// behind the barrier of the next K step (two 4-B stores per lane).
__global__ __launch_bounds__(V2_THREADS) void gemm_v2_stats_kernel(const f16* __restrict__ X, const f16* __restrict__ W,
                                                                   f16* __restrict__ out, int M, int N, int K, GemmTileStats stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, kg = lane >> 4;
  const V2Ring rg = v2_make_ring(smem, wave, lane);
  const unsigned voff = wave * 4096 + lane * 16;

  // id-order raster (grouped 8(m) x ntn super-tiles): M = beam x batch rows are a handful of row tiles
  const int ntm = M / 256, ntn = N / 256, nt = K / 32, nout = ntm * ntn;
  int tile_m = 0, tile_n = 0;
  auto coords = [&](int t) {
    constexpr int GM = 8;
    const int per_group = GM * ntn, group = t / per_group, first_m = group * GM;
    const int gsz = min(GM, ntm - first_m), in_group = t - group * per_group;
    tile_m = first_m + in_group % gsz;
    tile_n = in_group / gsz;
  };
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  if (tile >= nout) return;
  coords(tile);

  const size_t panel = (size_t)nt * (TM_BLOCK * 2);
  V2Stream st;
  st.xp = (const char*)X + (size_t)tile_m * panel + voff;
  st.wp = (const char*)W + (size_t)tile_n * panel + voff;
  st.inc = TM_BLOCK * 2;
  V2Frag f;
  v2_start(f, st, rg);

  const unsigned cbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + V2_RING_BYTES + wave * V2_CONST_BYTES);
  const unsigned lds_const0 = (unsigned)(size_t)smem + V2_RING_BYTES;
  const float sc2 = stats.scale * 1.4426950408889634f;  // > 0 (the launcher checks): max and scaling commute
  constexpr int NST = 32, NEMIT = 2;
  const int cidx = (kg & 1) * 2 + (kg >> 1);
  const unsigned lane_part = (unsigned)(((wr * 128 + l15) * 32 + ((cidx ^ tm_swz(l15)) << 3)) * 2);

  // the previous tile's statistics: wave u combines the two column waves' halves of rows 64u .. 64u+63
  int prev_m0 = 0, prev_tn = 0;
  auto emit_stats = [&]() {
    const int r = wave * 64 + lane;
    const unsigned src = lds_const0 + ((r >> 7) * 2) * V2_CONST_BYTES + (r & 127) * 8;
    float2 a, b;
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(src), "n"(V2_CONST_BYTES)
                 : "memory");
    const float m = fmaxf(a.x, b.x);
    const float ms = m == -INFINITY ? 0.f : m;
    const float sum = a.y * __builtin_amdgcn_exp2f(a.x - ms) + b.y * __builtin_amdgcn_exp2f(b.x - ms);
    const size_t o = (size_t)prev_tn * M + prev_m0 + r;
    stats.tile_max[o] = m * 0.6931471805599453f;
    stats.tile_sum[o] = sum;
  };

  bool more = true, first_tile = true;
  while (more) {
    if (first_tile) {
      v2_step_top<8>();
      v2_step_body<0, true>(f, st, rg);
      v2_step_top<8>();
    } else {
      v2_step_top<8 + NST>();
      emit_stats();
      v2_step_body<0, true>(f, st, rg);
      v2_step_top<8 + NEMIT + NST>();
    }
    first_tile = false;
    v2_step_body<1, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<2, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<3, false>(f, st, rg);
    for (int kb = 4; kb < nt - 4; kb += 4) {
      v2_step_top<8>();
      v2_step_body<0, false>(f, st, rg);
      v2_step_top<8>();
      v2_step_body<1, false>(f, st, rg);
      v2_step_top<8>();
      v2_step_body<2, false>(f, st, rg);
      v2_step_top<8>();
      v2_step_body<3, false>(f, st, rg);
    }
    v2_step_top<8>();
    v2_step_body<0, false>(f, st, rg);
    const int tm_cur = tile_m, tn_cur = tile_n;
    tile += (int)gridDim.x;
    more = tile < nout;
    if (more) {
      coords(tile);
      unsigned vo = voff;
      asm volatile("" : "+v"(vo));
      st.xp = (const char*)X + (size_t)tile_m * panel + vo;
      st.wp = (const char*)W + (size_t)tile_n * panel + vo;
    } else {
      st.xp -= st.inc;
      st.wp -= st.inc;
      st.inc = 0;
    }
    v2_step_top<8>();
    v2_step_body<1, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<2, false>(f, st, rg);
    v2_step_top<8>();
    v2_step_body<3, false>(f, st, rg);

    // ---- read-out, one 16-row block at a time: 4 stores + the row's statistics ----
    char* const obase = (char*)out + ((size_t)tm_cur * (N >> 5) + (size_t)tn_cur * 8 + wc * 4) * (TM_BLOCK * 2) + lane_part;
    const int n0 = tn_cur * 256;
    asm volatile(V2_RDOUT_FIRST_STR);
#define SMI_V2_SROW(MI, FULL)                                                                                            \
  {                                                                                                                      \
    uint32_t h[8][2];                                                                                                    \
    SMI_V2_SBLK(0, 0, MI) SMI_V2_SBLK(0, 1, MI) SMI_V2_SBLK(1, 0, MI) SMI_V2_SBLK(1, 1, MI)                              \
    SMI_V2_SBLK(2, 0, MI) SMI_V2_SBLK(2, 1, MI) SMI_V2_SBLK(3, 0, MI) SMI_V2_SBLK(3, 1, MI)                              \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
      const auto s0 = __builtin_amdgcn_permlane16_swap(h[2 * j][0], h[2 * j + 1][0], false, false);                      \
      const auto s1 = __builtin_amdgcn_permlane16_swap(h[2 * j][1], h[2 * j + 1][1], false, false);                      \
      const u32x4 chunk = {s0[0], s1[0], s0[1], s1[1]};                                                                  \
      store_nt((u32x4*)(obase + (size_t)j * (TM_BLOCK * 2) + (MI) * 1024), chunk);                                       \
    }                                                                                                                    \
    f32x2 t[16]; /* the lane's 32 ROUNDED values of the row (FULL: raw; else scaled to the log2 domain and masked) */     \
    float mx = -INFINITY;                                                                                                \
    _Pragma("unroll") for (int ni = 0; ni < 8; ++ni) _Pragma("unroll") for (int q = 0; q < 2; ++q) {                     \
      const half2v hv = __builtin_bit_cast(half2v, h[ni][q]);                                                            \
      f32x2 v = {(float)hv[0], (float)hv[1]};                                                                            \
      if constexpr (!(FULL)) {                                                                                           \
        const int col = n0 + wc * 128 + ni * 16 + 4 * kg + 2 * q;                                                        \
        v[0] = col < stats.valid_n ? v[0] * sc2 : -INFINITY;                                                             \
        v[1] = col + 1 < stats.valid_n ? v[1] * sc2 : -INFINITY;                                                         \
      }                                                                                                                  \
      t[ni * 2 + q] = v;                                                                                                 \
      mx = fmaxf(mx, fmaxf(v[0], v[1]));                                                                                 \
    }                                                                                                                    \
    {                                                                                                                    \
      const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);           \
      mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));                                                          \
      const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);           \
      mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));                                                          \
    }                                                                                                                    \
    /* the row maximum in the log2 domain (scale > 0: max and scaling commute, bit for bit) */                           \
    const float mxs = (FULL) ? mx * sc2 : mx;                                                                            \
    const float nb = mxs == -INFINITY ? 0.f : -mxs;                                                                      \
    const f32x2 mul2 = {(FULL) ? sc2 : 1.f, (FULL) ? sc2 : 1.f}, nb2 = {nb, nb};                                         \
    f32x2 se2 = {0.f, 0.f};                                                                                              \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                                     \
      const f32x2 a = (FULL) ? __builtin_elementwise_fma(t[e], mul2, nb2) : t[e] + nb2;                                  \
      se2 += f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};                                          \
    }                                                                                                                    \
    float se = se2[0] + se2[1];                                                                                          \
    {                                                                                                                    \
      const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(se), __float_as_uint(se), false, false);           \
      se = __uint_as_float(a[0]) + __uint_as_float(a[1]);                                                                \
      const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(se), __float_as_uint(se), false, false);           \
      se = __uint_as_float(b[0]) + __uint_as_float(b[1]);                                                                \
    }                                                                                                                    \
    if (kg == 0) {                                                                                                       \
      const float2 pv = {mxs, se};                                                                                       \
      asm volatile("ds_write_b64 %0, %1" ::"v"(cbase + ((MI) * 16 + l15) * 8), "v"(pv) : "memory");                      \
    }                                                                                                                    \
  }
#define SMI_V2_SBLK(J, NL, MI)                                                          \
  {                                                                                     \
    f32x4 v;                                                                            \
    SMI_V2_RDOUT_IDX(J, NL, MI, v);                                                     \
    const uint2 hp = __builtin_bit_cast(uint2, epi_act_pack<EPI_BIAS_F16>(v));          \
    h[2 * (J) + (NL)][0] = hp.x;                                                        \
    h[2 * (J) + (NL)][1] = hp.y;                                                        \
  }
    if (n0 + 256 <= stats.valid_n) {  // every tile but the last column tile
      SMI_V2_SROW(0, true) SMI_V2_SROW(1, true) SMI_V2_SROW(2, true) SMI_V2_SROW(3, true)
      SMI_V2_SROW(4, true) SMI_V2_SROW(5, true) SMI_V2_SROW(6, true) SMI_V2_SROW(7, true)
    } else {
      SMI_V2_SROW(0, false) SMI_V2_SROW(1, false) SMI_V2_SROW(2, false) SMI_V2_SROW(3, false)
      SMI_V2_SROW(4, false) SMI_V2_SROW(5, false) SMI_V2_SROW(6, false) SMI_V2_SROW(7, false)
    }
#undef SMI_V2_SBLK
#undef SMI_V2_SROW
    prev_m0 = tm_cur * 256;
    prev_tn = tn_cur;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  emit_stats();
}

template <int EPI, bool FOLD>
static hipError_t launch_v2(const f16* X, const f16* W, const float* c2, f16* out, int M, int N, int K, hipStream_t stream,
                            const GemmLnFold* fold) {
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_v2_kernel<EPI, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       V2_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done.set();
  }
  const int ntm = M / 256, ntn = N / 256;
  const int grid = std::min(ntm * ntn, num_cus());
  const int want_raster = tune(TUNE_G2_RASTER, 2);
  const int raster = (want_raster && grid == 256 && ntn % 4 == 0 && ntn >= 16 && ((ntm + 7) / 8) % 8 == 0) ? want_raster : 0;
  hipLaunchKernelGGL((gemm_v2_kernel<EPI, FOLD>), dim3(grid), dim3(V2_THREADS), V2_LDS_BYTES, stream, X, W, c2, out, M, N, K,
                     raster, fold ? *fold : GemmLnFold{nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0});
  return hipGetLastError();
}

bool gemm_v2_stats_fits(int M, int N, int K, const GemmTileStats* stats) {
  if (tune(TUNE_G2V2, 1) != 1 || !stats || !stats->tile_max || !stats->tile_sum || !(stats->scale > 0.f)) return false;
  if (M % 256 || N % 256 || K % 128 || K / 32 < V2_MIN_SLICES) return false;
  return (int64_t)(M / 256) * (N / 256) >= tune(TUNE_G2V2_MIN, 128);
}

hipError_t launch_gemm_v2_stats(const f16* X, const f16* W, f16* out, int M, int N, int K, hipStream_t stream,
                                const GemmTileStats* stats, int grid_cap) {
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_v2_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done.set();
  }
  int grid = std::min((M / 256) * (N / 256), num_cus());
  if (grid_cap > 0) grid = std::min(grid, grid_cap);
  hipLaunchKernelGGL(gemm_v2_stats_kernel, dim3(grid), dim3(V2_THREADS), V2_LDS_BYTES, stream, X, W, out, M, N, K, *stats);
  return hipGetLastError();
}

bool gemm_v2_fits(int epi, int M, int N, int K, const float* bias, const GemmLnFold* fold) {
  if (tune(TUNE_G2V2, 1) == 0) return false;
  if (M % 256 || N % 256 || K % 128 || K / 32 < V2_MIN_SLICES) return false;
  // from half a chip of tiles up (the automatic 256x256 threshold): measured same box with the threshold at 128 instead of
  // 512, decoder C5 3.62 -> 3.58 ms per step, C1 2.66 -> 2.63 ms (profiles/r06_experiments.txt, experiment 6)
  if ((int64_t)(M / 256) * (N / 256) < tune(TUNE_G2V2_MIN, 128)) return false;
  if (epi == EPI_RESID_F16 || epi == EPI_RESID_HALF_F16)  // tile-major residual stream; fold: producer side only
    return !fold || !fold->part_in;
  if (epi != EPI_BIAS_F16 && epi != EPI_RELU_F16 && epi != EPI_SILU_F16 && epi != EPI_GLU_F16) return false;
  if (!bias) return false;
  if (fold && (!fold->part_in || !fold->c1 || fold->nparts < 1 || fold->nparts > 4)) return false;
  if (fold && (epi == EPI_SILU_F16 || epi == EPI_GLU_F16) && !fold->centered) return false;
  return true;
}

template <int EPI, bool EMIT>
static hipError_t launch_v2_resid(const f16* X, const f16* W, const float* bias, f16* out, int M, int N, int K,
                                  hipStream_t stream, float2* part_out) {
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_v2_resid_kernel<EPI, EMIT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done.set();
  }
  const int ntm = M / 256, ntn = N / 256;
  const int grid = std::min(ntm * ntn, num_cus());
  const int want_raster = tune(TUNE_G2_RASTER, 2);
  const int raster = (want_raster && grid == 256 && ntn % 4 == 0 && ntn >= 16 && ((ntm + 7) / 8) % 8 == 0) ? want_raster : 0;
  hipLaunchKernelGGL((gemm_v2_resid_kernel<EPI, EMIT>), dim3(grid), dim3(V2_THREADS), V2_LDS_BYTES, stream, X, W, bias, out, M,
                     N, K, raster, part_out);
  return hipGetLastError();
}

hipError_t launch_gemm_v2(int epi, const f16* X, const f16* W, const float* bias, f16* out, int M, int N, int K,
                          hipStream_t stream, const GemmLnFold* fold) {
  if (epi == EPI_RESID_F16 || epi == EPI_RESID_HALF_F16) {
    float2* po = fold ? fold->part_out : nullptr;
    if (epi == EPI_RESID_F16)
      return po ? launch_v2_resid<EPI_RESID_F16, true>(X, W, bias, out, M, N, K, stream, po)
                : launch_v2_resid<EPI_RESID_F16, false>(X, W, bias, out, M, N, K, stream, nullptr);
    return po ? launch_v2_resid<EPI_RESID_HALF_F16, true>(X, W, bias, out, M, N, K, stream, po)
              : launch_v2_resid<EPI_RESID_HALF_F16, false>(X, W, bias, out, M, N, K, stream, nullptr);
  }
#define SMI_V2_CASE(E)                                                                           \
  case E:                                                                                        \
    return fold ? launch_v2<E, true>(X, W, bias, out, M, N, K, stream, fold)                     \
                : launch_v2<E, false>(X, W, bias, out, M, N, K, stream, nullptr);
  switch (epi) {
    SMI_V2_CASE(EPI_BIAS_F16)
    SMI_V2_CASE(EPI_RELU_F16)
    SMI_V2_CASE(EPI_SILU_F16)
    SMI_V2_CASE(EPI_GLU_F16)
  }
#undef SMI_V2_CASE
  return hipErrorInvalidValue;
}

}  // namespace smi
