// Activations and fp16 packing shared by the GEMM epilogues (gemm.hip, gemm_v2.hip).
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace smi {

// Activations of the fp16 epilogues.  v_rcp_f32 (1 ulp) instead of an IEEE division: `/` expands to a
// ~10-instruction div_scale / fma / div_fixup sequence per element, 128 elements per lane and tile,
// for a result that is rounded to fp16 right after.
__device__ __forceinline__ float sigmoid_f(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
__device__ __forceinline__ float silu_f(float v) { return v * sigmoid_f(v); }
// tanh(v) = 1 - 2 / (exp(2v) + 1); exact limits at +-inf (exp -> inf gives 1, exp -> 0 gives -1)
__device__ __forceinline__ float tanh_f(float v) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * v) + 1.0f);
}

template <int EPI>
__device__ __forceinline__ f32x4 epi_act(f32x4 v) {
  if constexpr (EPI == EPI_RELU_F16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if constexpr (EPI == EPI_SILU_F16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
  } else if constexpr (EPI == EPI_TANH_F16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = tanh_f(v[e]);
  }
  return v;
}

// Activation + rounding of 4 accumulator values to fp16.  relu runs on the rounded halves as a packed
// signed 16-bit integer max with 0 (a negative fp16 is a negative int16): one v_pk_max_i16 per two
// values instead of the canonicalise + v_max_f32 pair per value that fmaxf compiles to (MFMA results
// are not known-canonical), and relu(round(x)) == round(relu(x)).
template <int EPI>
__device__ __forceinline__ half4 epi_act_pack(f32x4 v) {
  typedef short short2v __attribute__((ext_vector_type(2)));
  if constexpr (EPI != EPI_RELU_F16) v = epi_act<EPI>(v);
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  half2v lo = __builtin_convertvector(f32x2{v[0], v[1]}, half2v);  // v_cvt_pk_f16_f32
  half2v hi = __builtin_convertvector(f32x2{v[2], v[3]}, half2v);
  if constexpr (EPI == EPI_RELU_F16) {
    const short2v z = {0, 0};
    lo = __builtin_bit_cast(half2v, __builtin_elementwise_max(__builtin_bit_cast(short2v, lo), z));
    hi = __builtin_bit_cast(half2v, __builtin_elementwise_max(__builtin_bit_cast(short2v, hi), z));
  }
  return half4{lo[0], lo[1], hi[0], hi[1]};
}

}  // namespace smi
