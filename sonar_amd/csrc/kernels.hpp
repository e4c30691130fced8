// Internal launch API of the HIP kernels (host side).  Everything here is
// stream-ordered and never synchronises the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smi {

typedef _Float16 f16;

enum GemmEpilogue { EPI_BIAS_F16 = 0, EPI_RELU_F16 = 1, EPI_RESID_F32 = 2 };

// C = X[M,K] * W[N,K]^T (+bias, epilogue).  M%128==0, N%128==0, K%64==0.
hipError_t launch_gemm_tn(int epi, const f16* X, const f16* W, const float* bias, void* out, int M,
                          int N, int K, int ldo, hipStream_t stream);

// x[row(n,p), :] = E[ids[n*S+p], :] * scale + PE[p + pos_offset, :]   (packed rows)
hipError_t launch_embed_pack(const int64_t* ids, const int32_t* cu_seqlens, const f16* table,
                             const float* pos_table, float scale, int pos_offset, float* x, int N,
                             int S, int max_len, int d, int64_t vocab, hipStream_t stream);

// h[r,:] = f16(LN(x[r,:]) * w + b)
hipError_t launch_layernorm(const float* x, const float* w, const float* b, float eps, f16* h,
                            int rows, int d, hipStream_t stream);

// Final LN + masked pooling (mean) over each sentence's packed rows.
// out: [N, d] in fp16 or fp32; encoded (optional): [N, S, d] same dtype, pads zeroed.
hipError_t launch_ln_pool(const float* x, const float* w, const float* b, float eps,
                          const int32_t* cu_seqlens, void* out, int out_is_f32, void* encoded, int N,
                          int S, int d, int pooling, hipStream_t stream);

// Self-attention over packed rows. qkv: [T, 3*d] (q | k | v), ctx: [T, d].  head_dim 64.
hipError_t launch_attention(const f16* qkv, const int32_t* cu_seqlens, f16* ctx, int N, int max_len,
                            int d, int heads, hipStream_t stream);

// dst_f16[i] = f16(src_f32[i])
hipError_t launch_f32_to_f16(const float* src, f16* dst, size_t n, hipStream_t stream);
// dst_f32[i] = float(src_f16[i])
hipError_t launch_f16_to_f32(const f16* src, float* dst, size_t n, hipStream_t stream);

// xsim: row-normalise then mine top-k cosine neighbours of each X row among Y rows.
hipError_t launch_l2_normalize(const void* src, int src_is_f32, f16* dst, int64_t rows, int d,
                               hipStream_t stream);
size_t xsim_workspace_bytes(int64_t nx_pad, int64_t ny_pad, int k);
hipError_t launch_xsim_topk(const f16* Xn, int64_t nx, int64_t nx_pad, const f16* Yn, int64_t ny,
                            int64_t ny_pad, int d, int k, int64_t y_index_offset, int32_t* idx,
                            float* score, void* workspace, hipStream_t stream);

}  // namespace smi
