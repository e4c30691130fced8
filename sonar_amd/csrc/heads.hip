// Embedding heads: the BLASER quality-estimation MLP and the MuTox toxicity classifier that the
// reference runs on top of SONAR embeddings.
//   BLASER  sonar/models/blaser/model.py:82-125 (F.normalize, featurize_input, mlp) with
//           config.py:14-24 (hidden [3072, 1536], TANH, COMET = 6 blocks / QE = 4 blocks)
//   MuTox   sonar/models/mutox/factory.py:15-38 (Linear 1024-512, ReLU, 512-128, ReLU, 128-1),
//           model.py:18-24 (optional sigmoid)
// Layout: features and hidden activations are fp16 [rows padded to 128][dim]; the hidden layers run
// on the shared MFMA GEMM engines (gemm.hip) with the bias + activation epilogue, the final
// out_dim <= 8 layer is one wave per row in fp32.  Dropout is inference-inert.
#include <cmath>
#include <vector>

#include "api_common.hpp"
#include "common.hpp"

using namespace smi;
using namespace smi_host;

namespace smi {

// One wave per row.  BLOCKS feature blocks of d columns; lane l owns columns l*8 + 512*k.
template <typename T>
__global__ __launch_bounds__(256) void head_featurize_kernel(int form, const T* __restrict__ src,
                                                             const T* __restrict__ mt, const T* __restrict__ ref,
                                                             int rows, int rows_pad, int d, int norm,
                                                             f16* __restrict__ out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows_pad) return;
  const int blocks = form == 0 ? 1 : (form == 1 ? 4 : 6);
  f16* o = out + (size_t)r * blocks * d;
  if (r >= rows) {
    for (int c = lane; c < blocks * d; c += 64) o[c] = (f16)0.f;
    return;
  }
  const T* s = src + (size_t)r * d;
  const T* m = mt ? mt + (size_t)r * d : nullptr;
  const T* f = ref ? ref + (size_t)r * d : nullptr;
  float is = 1.f, im = 1.f, ir = 1.f;
  if (norm) {  // F.normalize: x / max(||x||_2, 1e-12)
    float qs = 0.f, qm = 0.f, qr = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float a = (float)s[c];
      qs += a * a;
      if (m) {
        const float b = (float)m[c];
        qm += b * b;
      }
      if (f) {
        const float e = (float)f[c];
        qr += e * e;
      }
    }
    is = 1.0f / fmaxf(sqrtf(wave_sum(qs)), 1e-12f);
    im = 1.0f / fmaxf(sqrtf(wave_sum(qm)), 1e-12f);
    ir = 1.0f / fmaxf(sqrtf(wave_sum(qr)), 1e-12f);
  }
  for (int c = lane; c < d; c += 64) {
    const float a = (float)s[c] * is;
    if (form == 0) {
      o[c] = (f16)a;
      continue;
    }
    const float b = (float)m[c] * im;
    if (form == 1) {  // QE: [src, mt, src*mt, |mt-src|]
      o[c] = (f16)a;
      o[d + c] = (f16)b;
      o[2 * d + c] = (f16)(a * b);
      o[3 * d + c] = (f16)fabsf(b - a);
    } else {  // COMET: [ref, mt, src*mt, ref*mt, |mt-src|, |mt-ref|]
      const float e = (float)f[c] * ir;
      o[c] = (f16)e;
      o[d + c] = (f16)b;
      o[2 * d + c] = (f16)(a * b);
      o[3 * d + c] = (f16)(e * b);
      o[4 * d + c] = (f16)fabsf(b - a);
      o[5 * d + c] = (f16)fabsf(b - e);
    }
  }
}

hipError_t launch_head_featurize(int form, const void* src, const void* mt, const void* ref, int in_is_f32,
                                 int rows, int d, int norm, f16* out, hipStream_t stream) {
  if (form < 0 || form > 2 || rows <= 0 || d <= 0 || !src || (form >= 1 && !mt) || (form == 2 && !ref))
    return hipErrorInvalidValue;
  const int rows_pad = (rows + 127) / 128 * 128;
  const dim3 grid((rows_pad + 3) / 4);
  if (form < 2) ref = nullptr;
  if (form < 1) mt = nullptr;
  if (in_is_f32)
    hipLaunchKernelGGL(head_featurize_kernel<float>, grid, dim3(256), 0, stream, form, (const float*)src,
                       (const float*)mt, (const float*)ref, rows, rows_pad, d, norm, out);
  else
    hipLaunchKernelGGL(head_featurize_kernel<f16>, grid, dim3(256), 0, stream, form, (const f16*)src, (const f16*)mt,
                       (const f16*)ref, rows, rows_pad, d, norm, out);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void head_output_kernel(const f16* __restrict__ h, int ldh,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          int rows, int K, int out_dim, int act,
                                                          float* __restrict__ out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const f16* x = h + (size_t)r * ldh;
  for (int o = 0; o < out_dim; ++o) {
    float acc = 0.f;
    for (int c = lane; c < K; c += 64) acc += (float)x[c] * w[(size_t)o * K + c];
    acc = wave_sum(acc) + b[o];
    if (act == 1) acc = tanhf(acc);
    if (act == 2) acc = 1.0f / (1.0f + expf(-acc));
    if (lane == 0) out[(size_t)r * out_dim + o] = acc;
  }
}

hipError_t launch_head_output(const f16* h, int ldh, const float* w, const float* b, int rows, int K, int out_dim,
                              int act, float* out, hipStream_t stream) {
  if (rows <= 0 || K <= 0 || out_dim < 1 || out_dim > 8 || act < 0 || act > 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(head_output_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, h, ldh, w, b, rows, K, out_dim,
                     act, out);
  return hipGetLastError();
}

}  // namespace smi

struct smi_mlp_head {
  smi_mlp_head_config cfg;
  std::vector<int> dims;          // input_dim, hidden..., out_dim
  std::vector<DevBuf> w, b;       // hidden layers: fp16 W [out,in] + fp32 b; last layer: fp32 W + fp32 b
  DevBuf act[2];                  // ping-pong hidden activations [rows_pad][max hidden]
  int64_t cap_rows = 0;
};

extern "C" {

int smi_mlp_head_create(const smi_mlp_head_config* cfg, const smi_mlp_head_layer* layers, smi_mlp_head** out) {
  if (!cfg || !layers || !out) return fail(SMI_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (cfg->n_layers < 1 || cfg->n_layers > 8 || cfg->input_dim <= 0)
    return fail(SMI_ERR_INVALID_ARG, "n_layers %d / input_dim %d", cfg->n_layers, cfg->input_dim);
  if (cfg->hidden_act < 0 || cfg->hidden_act > 1 || cfg->out_act < 0 || cfg->out_act > 2)
    return fail(SMI_ERR_INVALID_ARG, "bad activation code");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  smi_mlp_head* H = new smi_mlp_head();
  H->cfg = *cfg;
  H->dims.push_back(cfg->input_dim);
  for (int l = 0; l < cfg->n_layers; ++l) H->dims.push_back(layers[l].out_dim);
  int rc = SMI_OK;
  for (int l = 0; l < cfg->n_layers && rc == SMI_OK; ++l) {
    const int in = H->dims[l], od = H->dims[l + 1];
    const bool last = l == cfg->n_layers - 1;
    if (od <= 0 || (last && od > 8))
      rc = fail(SMI_ERR_UNSUPPORTED, "layer %d: out_dim %d (the output layer supports 1..8)", l, od);
    else if (!last && (od % 128 || in % 64))
      rc = fail(SMI_ERR_UNSUPPORTED, "hidden layer %d: %d -> %d needs in %% 64 == 0 and out %% 128 == 0", l, in, od);
    if (rc != SMI_OK) break;
    H->w.emplace_back();
    H->b.emplace_back();
    rc = upload(layers[l].w, (int64_t)od * in, !last, H->w.back(), "mlp weight");
    if (rc == SMI_OK) rc = upload(layers[l].b, od, false, H->b.back(), "mlp bias");
  }
  if (rc != SMI_OK) {
    delete H;
    return rc;
  }
  *out = H;
  return SMI_OK;
}

void smi_mlp_head_destroy(smi_mlp_head* h) {
  if (!h) return;
  (void)hipDeviceSynchronize();
  delete h;
}

int smi_head_featurize(int32_t form, const void* src, const void* mt, const void* ref, int32_t dtype, int32_t rows,
                       int32_t d, int32_t norm_emb, void* out_f16, void* stream) {
  if (!src || !out_f16 || rows <= 0 || d <= 0) return fail(SMI_ERR_INVALID_ARG, "bad argument");
  if (form < 0 || form > 2) return fail(SMI_ERR_INVALID_ARG, "form %d (0 identity, 1 QE, 2 COMET)", form);
  if (form >= 1 && !mt) return fail(SMI_ERR_INVALID_ARG, "mt embeddings are required");
  if (form == 2 && !ref)
    return fail(SMI_ERR_INVALID_ARG, "With the COMET input form of BLASER, a reference embedding must be provided.");
  if (dtype != SMI_F32 && dtype != SMI_F16) return fail(SMI_ERR_INVALID_ARG, "bad dtype");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(launch_head_featurize(form, src, mt, ref, dtype == SMI_F32, rows, d, norm_emb, (f16*)out_f16,
                                (hipStream_t)stream));
  return SMI_OK;
}

int smi_mlp_head_forward(smi_mlp_head* H, const void* x_f16, int32_t rows, int32_t out_act, float* out,
                         void* stream_v) {
  if (!H || !x_f16 || !out || rows <= 0) return fail(SMI_ERR_INVALID_ARG, "bad argument");
  if (out_act < -1 || out_act > 2) return fail(SMI_ERR_INVALID_ARG, "bad out_act %d", out_act);
  hipStream_t stream = (hipStream_t)stream_v;
  const int rows_pad = (rows + 127) / 128 * 128;
  const int nl = H->cfg.n_layers;
  int max_hidden = 0;
  for (int l = 1; l < nl; ++l) max_hidden = std::max(max_hidden, H->dims[l]);
  if (nl > 1 && rows_pad > H->cap_rows) {
    HIP_TRY(hipDeviceSynchronize());
    for (auto& a : H->act) HIP_TRY(a.alloc((size_t)rows_pad * max_hidden * 2));
    H->cap_rows = rows_pad;
  }
  const f16* cur = (const f16*)x_f16;
  for (int l = 0; l + 1 < nl; ++l) {
    f16* nxt = H->act[l & 1].as<f16>();
    const int epi = H->cfg.hidden_act == 0 ? EPI_RELU_F16 : EPI_TANH_F16;
    HIP_TRY(launch_gemm_tn(epi, cur, H->w[l].as<f16>(), H->b[l].as<float>(), nxt, rows_pad, H->dims[l + 1], H->dims[l],
                           H->dims[l + 1], stream));
    cur = nxt;
  }
  HIP_TRY(launch_head_output(cur, H->dims[nl - 1], H->w[nl - 1].as<float>(), H->b[nl - 1].as<float>(), rows,
                             H->dims[nl - 1], H->dims[nl], out_act < 0 ? H->cfg.out_act : out_act, out, stream));
  return SMI_OK;
}

}  // extern "C"
