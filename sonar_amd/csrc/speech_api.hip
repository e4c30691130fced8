// C-ABI implementation of the speech encoder (see include/sonar_mi355.h): weight packing
// (fused qkv, GLU-interleaved pointwise conv, folded BatchNorm, relative-position table) and the
// per-layer launch schedule of the conformer + attention pooler.
// Reference op order: sonar/models/sonar_speech/model.py:59-77; conformer block / frontend
// semantics per SURVEY a27-a29 (fairseq2 ~=0.4).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "api_common.hpp"
#include "common.hpp"

using namespace smi;
using namespace smi_host;

namespace smi {

// dst row g*64 + c      = src row g*32 + c        (value half of GLU channel g*32+c)
// dst row g*64 + 32 + c = src row d + g*32 + c    (gate half)
__global__ void glu_interleave_kernel(const f16* __restrict__ src, f16* __restrict__ dst, int d) {
  const int o = blockIdx.x;  // output channel
  const int g = o >> 5, c = o & 31;
  for (int k = threadIdx.x; k < d; k += blockDim.x) {
    dst[(size_t)(g * 64 + c) * d + k] = src[(size_t)o * d + k];
    dst[(size_t)(g * 64 + 32 + c) * d + k] = src[(size_t)(d + o) * d + k];
  }
}

}  // namespace smi

namespace {

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

int fetch_host(const smi_tensor& t, int64_t expect, std::vector<float>& out, const char* name) {
  if (!t.data) return fail(SMI_ERR_INVALID_ARG, "weight %s: null data", name);
  if (t.numel != expect)
    return fail(SMI_ERR_INVALID_ARG, "weight %s: numel %lld, expected %lld", name, (long long)t.numel, (long long)expect);
  const size_t es = t.dtype == SMI_F32 ? 4 : 2;
  std::vector<char> raw((size_t)expect * es);
  HIP_TRY(hipMemcpy(raw.data(), t.data, raw.size(), t.on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
  out.resize(expect);
  if (t.dtype == SMI_F32) {
    std::memcpy(out.data(), raw.data(), raw.size());
  } else {
    const _Float16* h = (const _Float16*)raw.data();
    for (int64_t i = 0; i < expect; ++i) out[i] = (float)h[i];
  }
  return SMI_OK;
}

int upload_host(const std::vector<float>& v, DevBuf& dst) {
  HIP_TRY(dst.alloc(v.size() * 4));
  HIP_TRY(hipMemcpy(dst.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return SMI_OK;
}

struct ConfLayer {
  DevBuf ffn1_ln_w, ffn1_ln_b, ffn1_w1, ffn1_b1, ffn1_w2, ffn1_b2;
  DevBuf attn_ln_w, attn_ln_b, w_qkv, b_qkv, w_o, b_o, w_r, u_bias, v_bias;
  DevBuf conv_ln_w, conv_ln_b, w_pw1, w_dw, bn_scale, bn_shift, w_pw2;
  DevBuf ffn2_ln_w, ffn2_ln_b, ffn2_w1, ffn2_b1, ffn2_w2, ffn2_b2;
  DevBuf ln_w, ln_b;
  // LayerNorm folded into the GEMM that consumes it (kernels.hpp: GemmLnFold; round 4): the weights pre-scaled by the
  // LayerNorm weight and row-centred, tile-major, and c1 (unused by the centred epilogue) / c2 = W . b + bias
  DevBuf wf_qkv, c1_qkv, c2_qkv, wf_pw1, c1_pw1, c2_pw1, wf_ffn2, c1_ffn2, c2_ffn2;
};
struct PoolLayer {
  DevBuf sv_w, sv_b, so_w, so_b, sln_w, sln_b;
  DevBuf cq_w, cq_b, ckv_w, ckv_b, co_w, co_b, cln_w, cln_b;
  DevBuf f1_w, f1_b, f2_w, f2_b, fln_w, fln_b;
};

// fbank constants shared by every handle / smi_fbank call on a device
struct FbankConsts {
  DevBuf window, mel_w, mel_range;
  bool ready = false;
};
// one set per device (a process may run engines on several GPUs); creation is serialised
std::mutex g_fbank_mu;
FbankConsts& fbank_consts() {
  static FbankConsts c[64];
  return c[DeviceOnce::dev()];
}

int ensure_fbank_consts() {
  std::lock_guard<std::mutex> lock(g_fbank_mu);
  FbankConsts& c = fbank_consts();
  if (c.ready) return SMI_OK;
  const int N = 400, NB = 80, NF = 256;
  std::vector<float> win(N), mw((size_t)NB * NF, 0.f);
  std::vector<int> range(NB * 2);
  for (int i = 0; i < N; ++i) win[i] = (float)std::pow(0.5 - 0.5 * std::cos(2.0 * M_PI * i / (N - 1)), 0.85);
  auto mel = [](double f) { return 1127.0 * std::log(1.0 + f / 700.0); };
  const double lo = mel(20.0), hi = mel(8000.0), delta = (hi - lo) / (NB + 1), bw = 16000.0 / 512.0;
  for (int b = 0; b < NB; ++b) {
    const double left = lo + b * delta, center = left + delta, right = center + delta;
    int k0 = NF, k1 = 0;
    for (int k = 0; k < NF; ++k) {
      const double m = mel(bw * k);
      if (m > left && m < right) {
        mw[(size_t)b * NF + k] = (float)(m <= center ? (m - left) / (center - left) : (right - m) / (right - center));
        k0 = std::min(k0, k);
        k1 = std::max(k1, k + 1);
      }
    }
    range[b * 2] = k0 < k1 ? k0 : 0;
    range[b * 2 + 1] = k0 < k1 ? k1 : 0;
  }
  HIP_TRY(c.window.alloc(win.size() * 4));
  HIP_TRY(c.mel_w.alloc(mw.size() * 4));
  HIP_TRY(c.mel_range.alloc(range.size() * 4));
  HIP_TRY(hipMemcpy(c.window.p, win.data(), win.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c.mel_w.p, mw.data(), mw.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c.mel_range.p, range.data(), range.size() * 4, hipMemcpyHostToDevice));
  c.ready = true;
  return SMI_OK;
}

}  // namespace

struct smi_speech_encoder {
  smi_speech_encoder_config cfg;
  int kpad = 0;  // stacked feature dim padded to a multiple of 64
  int ffn_tile_major = 0;  // macaron FFN operands (LN output, hidden, weights) in the tile-major layout
  int x16 = 0;             // SMI_ENC_FP16_RESIDUAL: the conformer's residual stream is fp16
  int mid_tm = 0;          // the attention context and the depthwise-conv output (X of the two N = K = d GEMMs) tile-major
  int x_tm = 0;            // the fp16 residual stream itself tile-major + 3 of a block's 5 LayerNorms folded into their GEMMs
  DevBuf pe_ln_w, pe_ln_b, proj_w, proj_b, ln_w, ln_b, pool_q0, pool_out_w, rel_table;
  std::vector<ConfLayer> layers;
  std::vector<PoolLayer> pooler;
  // workspace
  DevBuf cu, hf, x, h, big, qkv, ctx, glu, dw, rp, enc_h, lnpart;
  DevBuf xq, hq, pv, pq, pkv, pctx, pffn, pout;
};

namespace {

int pack_fused(const smi_tensor* ws, const smi_tensor* bs, int count, int64_t d_out, int64_t d_in, DevBuf& w, DevBuf& b,
               const char* name) {
  HIP_TRY(w.alloc((size_t)count * d_out * d_in * 2));
  HIP_TRY(b.alloc((size_t)count * d_out * 4));
  for (int i = 0; i < count; ++i) {
    DevBuf tw, tb;
    if (int rc = upload(ws[i], d_out * d_in, true, tw, name)) return rc;
    if (int rc = upload(bs[i], d_out, false, tb, name)) return rc;
    HIP_TRY(hipMemcpy((char*)w.p + (size_t)i * d_out * d_in * 2, tw.p, (size_t)d_out * d_in * 2, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy((char*)b.p + (size_t)i * d_out * 4, tb.p, (size_t)d_out * 4, hipMemcpyDeviceToDevice));
  }
  return SMI_OK;
}

}  // namespace

extern "C" {

int64_t smi_fbank_num_frames(int64_t nsamples) { return nsamples < 400 ? 0 : 1 + (nsamples - 400) / 160; }

int smi_fbank(const float* wave, int64_t nsamples, float waveform_scale, int32_t standardize, float* out, void* stream) {
  if (!wave || !out) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  if (nsamples < 400) return SMI_OK;
  if (int rc = ensure_fbank_consts()) return rc;
  FbankConsts& c = fbank_consts();
  HIP_TRY(launch_fbank(wave, nsamples, waveform_scale, standardize, c.window.as<float>(), c.mel_w.as<float>(),
                       c.mel_range.as<int>(), out, (hipStream_t)stream));
  return SMI_OK;
}

int smi_fbank_batch(const float* waves, const int64_t* offsets, int32_t n, float waveform_scale, int32_t standardize,
                    float* out, int64_t tpad, void* stream_v) {
  if (!waves || !offsets || !out) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (n <= 0 || tpad <= 0 || tpad > (1 << 24)) return fail(SMI_ERR_INVALID_ARG, "bad n / tpad");
  for (int i = 0; i < n; ++i) {
    if (offsets[i + 1] < offsets[i]) return fail(SMI_ERR_INVALID_ARG, "offsets must be non-decreasing");
    if (smi_fbank_num_frames(offsets[i + 1] - offsets[i]) > tpad)
      return fail(SMI_ERR_INVALID_ARG, "clip %d has more than tpad = %lld frames", i, (long long)tpad);
  }
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  if (int rc = ensure_fbank_consts()) return rc;
  FbankConsts& c = fbank_consts();
  hipStream_t stream = (hipStream_t)stream_v;
  int64_t* off_dev = nullptr;
  HIP_TRY(hipMallocAsync((void**)&off_dev, (size_t)(n + 1) * sizeof(int64_t), stream));
  HIP_TRY(hipMemcpyAsync(off_dev, offsets, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
  HIP_TRY(launch_fbank_batch(waves, off_dev, n, (int)tpad, waveform_scale, standardize, c.window.as<float>(),
                             c.mel_w.as<float>(), c.mel_range.as<int>(), out, stream));
  HIP_TRY(hipFreeAsync(off_dev, stream));
  return SMI_OK;
}

int smi_speech_encoder_create(const smi_speech_encoder_config* cfg, const smi_speech_encoder_weights* w,
                              smi_speech_encoder** out) {
  if (!cfg || !w || !out) return fail(SMI_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  const smi_speech_encoder_config& c = *cfg;
  if (c.model_dim <= 0 || c.model_dim != c.num_heads * 64 || c.model_dim != c.pooler_heads * 64)
    return fail(SMI_ERR_UNSUPPORTED, "model_dim %d must equal heads * 64", c.model_dim);
  if (c.model_dim % 256 || (c.model_dim / 256 > 4 && c.model_dim != 2048))
    return fail(SMI_ERR_UNSUPPORTED, "model_dim %d must be 256/512/768/1024/2048", c.model_dim);
  if (c.ffn_inner_dim % 128 || c.pooler_ffn_dim % 128) return fail(SMI_ERR_UNSUPPORTED, "ffn dims must be multiples of 128");
  if (c.conv_kernel != 31 && c.conv_kernel != 7) return fail(SMI_ERR_UNSUPPORTED, "conv_kernel %d not built (31, 7)", c.conv_kernel);
  if (c.num_mel_bins <= 0 || 2 * c.num_mel_bins > 192) return fail(SMI_ERR_UNSUPPORTED, "num_mel_bins %d", c.num_mel_bins);
  if (c.max_frames < 2 || c.num_layers < 1 || c.pooler_layers < 1 || c.bos_idx < 0 || c.bos_idx >= c.pooler_vocab)
    return fail(SMI_ERR_INVALID_ARG, "bad max_frames/num_layers/pooler_layers/bos_idx");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  if (!w->layers || !w->pooler) return fail(SMI_ERR_INVALID_ARG, "null layers");

  smi_speech_encoder* E = new smi_speech_encoder();
  E->cfg = c;
  const int64_t d = c.model_dim, f = c.ffn_inner_dim, fd = 2 * c.num_mel_bins, pf = c.pooler_ffn_dim;
  E->kpad = (int)round_up(fd, 64);
  int rc = SMI_OK;
  auto up = [&](const smi_tensor& t, int64_t numel, bool f16w, DevBuf& dst, const char* name) {
    if (rc == SMI_OK) rc = upload(t, numel, f16w, dst, name);
  };
  up(w->post_extract_layer_norm_w, fd, false, E->pe_ln_w, "post_extract_layer_norm.weight");
  up(w->post_extract_layer_norm_b, fd, false, E->pe_ln_b, "post_extract_layer_norm.bias");
  up(w->model_dim_proj_b, d, false, E->proj_b, "model_dim_proj.bias");
  up(w->layer_norm_w, d, false, E->ln_w, "layer_norm.weight");
  up(w->layer_norm_b, d, false, E->ln_b, "layer_norm.bias");
  up(w->pooler_projection_out_w, d * d, true, E->pool_out_w, "projection_out.weight");
  if (rc == SMI_OK) {  // model_dim_proj [d, fd] -> [d, kpad] zero-padded along K
    std::vector<float> pw;
    rc = fetch_host(w->model_dim_proj_w, d * fd, pw, "model_dim_proj.weight");
    if (rc == SMI_OK) {
      std::vector<_Float16> padded((size_t)d * E->kpad, (_Float16)0.f);
      for (int64_t r = 0; r < d; ++r)
        for (int64_t k = 0; k < fd; ++k) padded[(size_t)r * E->kpad + k] = (_Float16)pw[(size_t)r * fd + k];
      hipError_t he = E->proj_w.alloc(padded.size() * 2);
      if (he == hipSuccess) he = hipMemcpy(E->proj_w.p, padded.data(), padded.size() * 2, hipMemcpyHostToDevice);
      if (he != hipSuccess) rc = fail(SMI_ERR_HIP, "model_dim_proj upload: %s", hipGetErrorString(he));
    }
  }
  if (rc == SMI_OK) {  // pooler query: E_p[bos] * sqrt(d) + PE[0] (sin half 0, cos half 1)
    std::vector<float> emb;
    rc = fetch_host(w->pooler_embed, (int64_t)c.pooler_vocab * d, emb, "pooler embed");
    if (rc == SMI_OK) {
      std::vector<float> q0(d);
      const float scale = std::sqrt((float)d);
      for (int64_t k = 0; k < d; ++k) {
        const float e16 = (float)(_Float16)emb[(size_t)c.bos_idx * d + k];
        q0[k] = (float)(_Float16)(e16 * scale) + (k < d / 2 ? 0.f : 1.f);
      }
      rc = upload_host(q0, E->pool_q0);
    }
  }
  if (rc == SMI_OK) {  // relative-position table, ascending: row rho <-> rel = rho - (P - 1)
    const int64_t P = c.max_frames + 192;
    std::vector<_Float16> tab((size_t)(2 * P - 1) * d);
    for (int64_t rho = 0; rho < 2 * P - 1; ++rho) {
      const double rel = (double)(rho - (P - 1));
      for (int64_t i = 0; i < d / 2; ++i) {
        const double div = std::exp(-(double)(2 * i) * std::log(10000.0) / (double)d);
        tab[(size_t)rho * d + 2 * i] = (_Float16)std::sin(rel * div);
        tab[(size_t)rho * d + 2 * i + 1] = (_Float16)std::cos(rel * div);
      }
    }
    hipError_t he = E->rel_table.alloc(tab.size() * 2);
    if (he == hipSuccess) he = hipMemcpy(E->rel_table.p, tab.data(), tab.size() * 2, hipMemcpyHostToDevice);
    if (he != hipSuccess) rc = fail(he == hipErrorOutOfMemory ? SMI_ERR_OOM : SMI_ERR_HIP, "rel table: %s", hipGetErrorString(he));
  }
  E->layers.resize(c.num_layers);
  E->ffn_tile_major = f % 256 == 0;  // d % 256 == 0 is checked above
  E->x16 = (c.flags & SMI_ENC_FP16_RESIDUAL) != 0;
  // Round 4: the per-clip kernels (relative-position attention, depthwise conv) write their outputs tile-major -- any packed
  // row addresses its own 64-B slice of a block, no clip alignment is needed -- so the attention-output and pointwise_conv2
  // GEMMs (N = K = d, the least efficient shape of the block) read both operands as linear 16 KiB bursts.  SMI_SPEECH_MID_TM=0
  // restores the row-major operands (A/B, read at create).
  {
    E->mid_tm = E->ffn_tile_major && tune(TUNE_SPEECH_MID_TM, 1) != 0;
  }
  // Round 4: with every GEMM operand of a block tile-major, the fp16 residual stream can be tile-major too (the text
  // encoder's layout): the four residual GEMMs read-modify-write it straight from their accumulators (LAYOUT 3)
  // and leave per-row (sum, sum of squares), and the LayerNorms in front of the fused QKV, pointwise_conv1 and the second
  // FFN's inner projection are applied in those GEMMs' epilogues (centred weights) -- 72 row launches and their h round trips
  // per forward disappear.  The block-final LayerNorm rewrites the stream and stays a kernel (ln2, with the next block's
  // first LayerNorm).  SMI_SPEECH_X_TM=0 restores the row-major stream (A/B, read at create).
  {
    E->x_tm = E->mid_tm && E->x16 && d % 512 == 0 && tune(TUNE_SPEECH_X_TM, 1) != 0;
  }
  auto fold_prep = [&](const DevBuf& W, const DevBuf& g, const DevBuf& b, const float* bias, int64_t N, int64_t K,
                       DevBuf& Wf, DevBuf& c1, DevBuf& c2) -> int {
    HIP_TRY(Wf.alloc((size_t)N * K * 2));
    HIP_TRY(c1.alloc((size_t)N * 4));
    HIP_TRY(c2.alloc((size_t)N * 4));
    HIP_TRY(launch_ln_fold_prep(W.as<f16>(), g.as<float>(), b.as<float>(), bias, Wf.as<f16>(), c1.as<float>(),
                                c2.as<float>(), (int)N, (int)K, 1, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return to_tile_major(Wf, (int)N, (int)K);
  };
  for (int l = 0; l < c.num_layers && rc == SMI_OK; ++l) {
    const smi_conformer_layer& s = w->layers[l];
    ConfLayer& L = E->layers[l];
    up(s.ffn1_layer_norm_w, d, false, L.ffn1_ln_w, "ffn1_layer_norm.weight");
    up(s.ffn1_layer_norm_b, d, false, L.ffn1_ln_b, "ffn1_layer_norm.bias");
    up(s.ffn1_inner_w, f * d, true, L.ffn1_w1, "ffn1.inner_proj.weight");
    up(s.ffn1_inner_b, f, false, L.ffn1_b1, "ffn1.inner_proj.bias");
    up(s.ffn1_out_w, d * f, true, L.ffn1_w2, "ffn1.output_proj.weight");
    up(s.ffn1_out_b, d, false, L.ffn1_b2, "ffn1.output_proj.bias");
    up(s.self_attn_layer_norm_w, d, false, L.attn_ln_w, "self_attn_layer_norm.weight");
    up(s.self_attn_layer_norm_b, d, false, L.attn_ln_b, "self_attn_layer_norm.bias");
    up(s.out_w, d * d, true, L.w_o, "self_attn.output_proj.weight");
    up(s.out_b, d, false, L.b_o, "self_attn.output_proj.bias");
    up(s.r_proj_w, d * d, true, L.w_r, "self_attn.sdpa.r_proj.weight");
    up(s.u_bias, d, false, L.u_bias, "self_attn.sdpa.u_bias");
    up(s.v_bias, d, false, L.v_bias, "self_attn.sdpa.v_bias");
    up(s.conv_layer_norm_w, d, false, L.conv_ln_w, "conv_layer_norm.weight");
    up(s.conv_layer_norm_b, d, false, L.conv_ln_b, "conv_layer_norm.bias");
    up(s.depthwise_conv_w, d * c.conv_kernel, false, L.w_dw, "conv.depthwise_conv.weight");
    up(s.pointwise_conv2_w, d * d, true, L.w_pw2, "conv.pointwise_conv2.weight");
    up(s.ffn2_layer_norm_w, d, false, L.ffn2_ln_w, "ffn2_layer_norm.weight");
    up(s.ffn2_layer_norm_b, d, false, L.ffn2_ln_b, "ffn2_layer_norm.bias");
    up(s.ffn2_inner_w, f * d, true, L.ffn2_w1, "ffn2.inner_proj.weight");
    up(s.ffn2_inner_b, f, false, L.ffn2_b1, "ffn2.inner_proj.bias");
    up(s.ffn2_out_w, d * f, true, L.ffn2_w2, "ffn2.output_proj.weight");
    up(s.ffn2_out_b, d, false, L.ffn2_b2, "ffn2.output_proj.bias");
    up(s.layer_norm_w, d, false, L.ln_w, "layer_norm.weight");
    up(s.layer_norm_b, d, false, L.ln_b, "layer_norm.bias");
    if (rc == SMI_OK && E->mid_tm) {
      rc = to_tile_major(L.w_o, (int)d, (int)d);
      if (rc == SMI_OK) rc = to_tile_major(L.w_pw2, (int)d, (int)d);
    }
    if (rc == SMI_OK && E->x_tm)  // (row-major weights in, before they are packed)
      rc = fold_prep(L.ffn2_w1, L.ffn2_ln_w, L.ffn2_ln_b, L.ffn2_b1.as<float>(), f, d, L.wf_ffn2, L.c1_ffn2, L.c2_ffn2);
    if (rc == SMI_OK && E->ffn_tile_major) {
      rc = to_tile_major(L.ffn1_w1, (int)f, (int)d);
      if (rc == SMI_OK) rc = to_tile_major(L.ffn1_w2, (int)d, (int)f);
      if (rc == SMI_OK) rc = to_tile_major(L.ffn2_w1, (int)f, (int)d);
      if (rc == SMI_OK) rc = to_tile_major(L.ffn2_w2, (int)d, (int)f);
    }
    if (rc == SMI_OK) {
      const smi_tensor ws[3] = {s.q_w, s.k_w, s.v_w}, bs[3] = {s.q_b, s.k_b, s.v_b};
      rc = pack_fused(ws, bs, 3, d, d, L.w_qkv, L.b_qkv, "self_attn.qkv");
      if (rc == SMI_OK && E->x_tm)
        rc = fold_prep(L.w_qkv, L.attn_ln_w, L.attn_ln_b, L.b_qkv.as<float>(), 3 * d, d, L.wf_qkv, L.c1_qkv, L.c2_qkv);
      // the QKV and pointwise_conv1 GEMMs read a LayerNorm output: packed rows, no per-clip alignment in the
      // way, so their INPUTS are tile-major too (their outputs feed per-clip kernels and stay row-major)
      if (rc == SMI_OK && E->ffn_tile_major) rc = to_tile_major(L.w_qkv, (int)(3 * d), (int)d);
    }
    if (rc == SMI_OK) {  // pointwise_conv1 rows interleaved for the GLU epilogue
      DevBuf tmp;
      rc = upload(s.pointwise_conv1_w, 2 * d * d, true, tmp, "conv.pointwise_conv1.weight");
      if (rc == SMI_OK) {
        hipError_t he = L.w_pw1.alloc((size_t)2 * d * d * 2);
        if (he == hipSuccess) {
          hipLaunchKernelGGL(glu_interleave_kernel, dim3((unsigned)d), dim3(256), 0, nullptr, tmp.as<f16>(), L.w_pw1.as<f16>(), (int)d);
          he = hipDeviceSynchronize();
        }
        if (he != hipSuccess) rc = fail(SMI_ERR_HIP, "glu interleave: %s", hipGetErrorString(he));
        if (rc == SMI_OK && E->x_tm)  // the interleaved rows are scaled / centred row by row: the interleave commutes
          rc = fold_prep(L.w_pw1, L.conv_ln_w, L.conv_ln_b, nullptr, 2 * d, d, L.wf_pw1, L.c1_pw1, L.c2_pw1);
        if (rc == SMI_OK && E->ffn_tile_major) rc = to_tile_major(L.w_pw1, (int)(2 * d), (int)d);
      }
    }
    if (rc == SMI_OK) {  // BatchNorm (eval) folded to scale / shift
      std::vector<float> g, b, mu, var;
      rc = fetch_host(s.batch_norm_w, d, g, "batch_norm.weight");
      if (rc == SMI_OK) rc = fetch_host(s.batch_norm_b, d, b, "batch_norm.bias");
      if (rc == SMI_OK) rc = fetch_host(s.batch_norm_mean, d, mu, "batch_norm.running_mean");
      if (rc == SMI_OK) rc = fetch_host(s.batch_norm_var, d, var, "batch_norm.running_var");
      if (rc == SMI_OK) {
        std::vector<float> sc(d), sh(d);
        for (int64_t i = 0; i < d; ++i) {
          sc[i] = g[i] / std::sqrt(var[i] + c.bn_eps);
          sh[i] = b[i] - mu[i] * sc[i];
        }
        rc = upload_host(sc, L.bn_scale);
        if (rc == SMI_OK) rc = upload_host(sh, L.bn_shift);
      }
    }
  }
  E->pooler.resize(c.pooler_layers);
  for (int l = 0; l < c.pooler_layers && rc == SMI_OK; ++l) {
    const smi_pooler_layer& s = w->pooler[l];
    PoolLayer& L = E->pooler[l];
    up(s.self_v_w, d * d, true, L.sv_w, "pooler self_attn.v_proj.weight");
    up(s.self_v_b, d, false, L.sv_b, "pooler self_attn.v_proj.bias");
    up(s.self_out_w, d * d, true, L.so_w, "pooler self_attn.output_proj.weight");
    up(s.self_out_b, d, false, L.so_b, "pooler self_attn.output_proj.bias");
    up(s.self_attn_layer_norm_w, d, false, L.sln_w, "pooler self_attn_layer_norm.weight");
    up(s.self_attn_layer_norm_b, d, false, L.sln_b, "pooler self_attn_layer_norm.bias");
    up(s.cross_q_w, d * d, true, L.cq_w, "pooler encoder_decoder_attn.q_proj.weight");
    up(s.cross_q_b, d, false, L.cq_b, "pooler encoder_decoder_attn.q_proj.bias");
    up(s.cross_out_w, d * d, true, L.co_w, "pooler encoder_decoder_attn.output_proj.weight");
    up(s.cross_out_b, d, false, L.co_b, "pooler encoder_decoder_attn.output_proj.bias");
    up(s.cross_layer_norm_w, d, false, L.cln_w, "pooler encoder_decoder_attn_layer_norm.weight");
    up(s.cross_layer_norm_b, d, false, L.cln_b, "pooler encoder_decoder_attn_layer_norm.bias");
    up(s.ffn_inner_w, pf * d, true, L.f1_w, "pooler ffn.inner_proj.weight");
    up(s.ffn_inner_b, pf, false, L.f1_b, "pooler ffn.inner_proj.bias");
    up(s.ffn_out_w, d * pf, true, L.f2_w, "pooler ffn.output_proj.weight");
    up(s.ffn_out_b, d, false, L.f2_b, "pooler ffn.output_proj.bias");
    up(s.ffn_layer_norm_w, d, false, L.fln_w, "pooler ffn_layer_norm.weight");
    up(s.ffn_layer_norm_b, d, false, L.fln_b, "pooler ffn_layer_norm.bias");
    if (rc == SMI_OK) {
      const smi_tensor ws[2] = {s.cross_k_w, s.cross_v_w}, bs[2] = {s.cross_k_b, s.cross_v_b};
      rc = pack_fused(ws, bs, 2, d, d, L.ckv_w, L.ckv_b, "pooler cross kv");
    }
  }
  if (rc != SMI_OK) {
    delete E;
    return rc;
  }
  *out = E;
  return SMI_OK;
}

void smi_speech_encoder_destroy(smi_speech_encoder* enc) {
  if (!enc) return;
  (void)hipDeviceSynchronize();
  delete enc;
}

int smi_speech_encoder_forward(smi_speech_encoder* E, const float* fbank, const int32_t* fbank_lens, int32_t n,
                               int32_t t, void* out_emb, int32_t out_dtype, void* stream_v) {
  if (!E || !fbank || !out_emb) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (n <= 0 || t <= 0 || (t & 1)) return fail(SMI_ERR_INVALID_ARG, "need n > 0 and an even number of frames (t=%d)", t);
  if (out_dtype != SMI_F32 && out_dtype != SMI_F16) return fail(SMI_ERR_INVALID_ARG, "bad out_dtype");
  const smi_speech_encoder_config& c = E->cfg;
  hipStream_t stream = (hipStream_t)stream_v;
  const int d = c.model_dim, f = c.ffn_inner_dim, pf = c.pooler_ffn_dim;

  // stacked-frame offsets (frames // 2 per clip: the padding-mask length after stacking)
  std::vector<int32_t> cu(n + 1);
  int64_t total = 0;
  int tm = 0;
  cu[0] = 0;
  for (int i = 0; i < n; ++i) {
    const int len = fbank_lens ? fbank_lens[i] : t;
    if (len < 0 || len > t) return fail(SMI_ERR_INVALID_ARG, "fbank_lens[%d]=%d outside [0,%d]", i, len, t);
    total += len / 2;
    cu[i + 1] = (int32_t)total;
    tm = std::max(tm, len / 2);
  }
  if (tm > c.max_frames) return fail(SMI_ERR_INVALID_ARG, "%d stacked frames exceed max_frames %d", tm, c.max_frames);
  if (tm < 1) return fail(SMI_ERR_INVALID_ARG, "no clip has two or more frames");
  const int R = (int)round_up(total, 256), NP = (int)round_up(n, 256);
  const int rp_m = (int)round_up(2 * tm - 1, 128);

  HIP_TRY(E->cu.reserve((size_t)(n + 1) * 4));
  HIP_TRY(hipMemcpyAsync(E->cu.p, cu.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipStreamSynchronize(stream));  // cu lives on the host stack of this call
  const int32_t* dcu = E->cu.as<int32_t>();
  const size_t before = E->hf.bytes + E->x.bytes + E->ctx.bytes + E->glu.bytes + E->dw.bytes + E->big.bytes;
  HIP_TRY(E->hf.reserve((size_t)R * E->kpad * 2));
  HIP_TRY(E->x.reserve((size_t)R * d * 4));  // sized for fp32; an fp16 stream uses half of it
  HIP_TRY(E->h.reserve((size_t)R * d * 2));
  HIP_TRY(E->big.reserve((size_t)R * f * 2));
  HIP_TRY(E->qkv.reserve((size_t)R * 3 * d * 2));
  HIP_TRY(E->ctx.reserve((size_t)R * d * 2));
  HIP_TRY(E->glu.reserve((size_t)R * d * 2));
  HIP_TRY(E->dw.reserve((size_t)R * d * 2));
  HIP_TRY(E->rp.reserve((size_t)rp_m * d * 2));
  HIP_TRY(E->xq.reserve((size_t)NP * d * 4));
  HIP_TRY(E->hq.reserve((size_t)NP * d * 2));
  HIP_TRY(E->pv.reserve((size_t)NP * d * 2));
  HIP_TRY(E->pq.reserve((size_t)NP * d * 2));
  HIP_TRY(E->pkv.reserve((size_t)R * 2 * d * 2));
  HIP_TRY(E->pctx.reserve((size_t)NP * d * 2));
  HIP_TRY(E->pffn.reserve((size_t)NP * pf * 2));
  HIP_TRY(E->pout.reserve((size_t)NP * d * 4));
  if (E->hf.bytes + E->x.bytes + E->ctx.bytes + E->glu.bytes + E->dw.bytes + E->big.bytes != before) {
    // rows that only the GEMM tile padding touches must stay finite
    HIP_TRY(hipMemsetAsync(E->hf.p, 0, E->hf.bytes, stream));
    HIP_TRY(hipMemsetAsync(E->ctx.p, 0, E->ctx.bytes, stream));
    HIP_TRY(hipMemsetAsync(E->glu.p, 0, E->glu.bytes, stream));
    HIP_TRY(hipMemsetAsync(E->dw.p, 0, E->dw.bytes, stream));
    HIP_TRY(hipMemsetAsync(E->pctx.p, 0, E->pctx.bytes, stream));
    HIP_TRY(hipMemsetAsync(E->hq.p, 0, E->hq.bytes, stream));
    HIP_TRY(hipMemsetAsync(E->xq.p, 0, E->xq.bytes, stream));
  }
  void* x = E->x.p;
  const int x16 = E->x16;
  const int epi_res = x16 ? EPI_RESID_F16 : EPI_RESID_F32, epi_half = x16 ? EPI_RESID_HALF_F16 : EPI_RESID_HALF_F32;
  f16* h = E->h.as<f16>();
  f16* big = E->big.as<f16>();
  f16* qkv = E->qkv.as<f16>();
  f16* ctx = E->ctx.as<f16>();

  // frontend: stack 2 frames -> LN(160) -> Linear(160 -> d)
  HIP_TRY(launch_stack_ln(fbank, n, t, c.num_mel_bins, dcu, tm, E->pe_ln_w.as<float>(), E->pe_ln_b.as<float>(), c.ln_eps,
                          E->hf.as<f16>(), E->kpad, stream));
  const int xtm = E->x_tm;
  if (xtm) {  // row-major out of the projection (into h, free here), then one pass into the tile-major stream
    HIP_TRY(E->lnpart.reserve((size_t)(d / 256) * R * sizeof(float2)));
    HIP_TRY(launch_gemm_tn(EPI_BIAS_F16, E->hf.as<f16>(), E->proj_w.as<f16>(), E->proj_b.as<float>(), h, R, d, E->kpad, d, stream));
    HIP_TRY(launch_pack_tile_major(h, (f16*)x, R, d, 0, stream));
  } else {
    HIP_TRY(launch_gemm_tn(x16 ? EPI_BIAS_F16 : EPI_STORE_F32, E->hf.as<f16>(), E->proj_w.as<f16>(), E->proj_b.as<float>(), x, R, d,
                           E->kpad, d, stream));
  }
  // fold descriptors: the residual GEMMs leave lnpart[d/256][R] (sum, sum of squares) of the rows they have just written,
  // the next consuming GEMM reads them (one buffer: produced and consumed in stream order, once per pair)
  float2* lnpart = E->lnpart.as<float2>();
  const GemmLnFold produce{lnpart, nullptr, nullptr, 0, 0.f, 0.f, 0};
  auto consume = [&](const DevBuf& c1) { return GemmLnFold{nullptr, lnpart, c1.as<float>(), d / 256, 1.0f / d, c.ln_eps, 1}; };
  const int io_tm = GEMM_IN_TM | GEMM_OUT_TM;
  // relative positions rel in [-(tm-1), tm-1] (+ tile padding) from the ascending table
  const int64_t P = c.max_frames + 192;
  const f16* pe_slice = E->rel_table.as<f16>() + (size_t)((P - 1) - (tm - 1)) * d;

  // The two macaron FFNs (4 of the block's 8 GEMMs, 2/3 of its FLOPs) run on tile-major operands
  // (common.hpp): their LayerNorm input, the SiLU hidden activation and the weights.
  const int tmf = E->ffn_tile_major;
  const int ffn_in = tmf ? GEMM_IN_TM : 0, ffn_io = tmf ? GEMM_IN_TM | GEMM_OUT_TM : 0;
  const int mid_in = E->mid_tm ? GEMM_IN_TM : 0;
  HIP_TRY(launch_layernorm(x, E->layers[0].ffn1_ln_w.as<float>(), E->layers[0].ffn1_ln_b.as<float>(), c.ln_eps, h, R, d, stream,
                           tmf, x16, xtm));
  // q | k | v tile-major between the fused QKV GEMM and the attention: with the LayerNorm fold on the tile-major stream only (the
  // GEMM then runs on the 4-wave engine, which has no row-major epilogue) and when the attention kernel that reads it is on
  const int qkv_tm = xtm && d % 256 == 0 && relpos_attention_reads_tile_major();
  int glu_tm = 0;  // set per layer where the fold-consumer GLU GEMM is launched
  for (int l = 0; l < c.num_layers; ++l) {
    ConfLayer& L = E->layers[l];
    // x += 0.5 * FFN1(LN(x))
    HIP_TRY(launch_gemm_tn(EPI_SILU_F16 | ffn_io, h, L.ffn1_w1.as<f16>(), L.ffn1_b1.as<float>(), big, R, f, d, f, stream));
    if (xtm) {
      HIP_TRY(launch_gemm_tn(EPI_RESID_HALF_F16 | io_tm, big, L.ffn1_w2.as<f16>(), L.ffn1_b2.as<float>(), x, R, d, f, d, stream,
                             nullptr, &produce));
      // x += RelPosMHA(LN(x)): the LayerNorm rides in the fused QKV GEMM, which multiplies the stream itself
      const GemmLnFold cq = consume(L.c1_qkv);
      HIP_TRY(launch_gemm_tn(EPI_BIAS_F16 | GEMM_IN_TM | (qkv_tm ? GEMM_OUT_TM : 0), (const f16*)x, L.wf_qkv.as<f16>(),
                             L.c2_qkv.as<float>(), qkv, R, 3 * d, d, 3 * d, stream, nullptr, &cq));
    } else {
      HIP_TRY(launch_gemm_tn(epi_half | ffn_in, big, L.ffn1_w2.as<f16>(), L.ffn1_b2.as<float>(), x, R, d, f, d,
                             stream));
      // x += RelPosMHA(LN(x))
      HIP_TRY(launch_layernorm(x, L.attn_ln_w.as<float>(), L.attn_ln_b.as<float>(), c.ln_eps, h, R, d, stream, tmf, x16));
      HIP_TRY(launch_gemm_tn(EPI_BIAS_F16 | ffn_in, h, L.w_qkv.as<f16>(), L.b_qkv.as<float>(), qkv, R, 3 * d, d, 3 * d, stream));
    }
    HIP_TRY(launch_gemm_tn(EPI_BIAS_F16, pe_slice, L.w_r.as<f16>(), nullptr, E->rp.p, rp_m, d, d, d, stream));
    HIP_TRY(launch_relpos_attention(qkv, dcu, E->rp.as<f16>(), tm - 1, rp_m, L.u_bias.as<float>(), L.v_bias.as<float>(), ctx,
                                    n, tm, d, c.num_heads, stream, E->mid_tm, qkv_tm));
    // x += Conv(LN(x)): pointwise(d->2d)+GLU, depthwise+BN+SiLU, pointwise(d->d)
    if (xtm) {
      HIP_TRY(launch_gemm_tn(EPI_RESID_F16 | io_tm, ctx, L.w_o.as<f16>(), L.b_o.as<float>(), x, R, d, d, d, stream, nullptr,
                             &produce));
      const GemmLnFold cg = consume(L.c1_pw1);
      // tile-major GLU output where the 4-wave engine takes the launch (its GLU read-out; SPEECH_GLU_TM=0: row-major, 8-wave engine)
      glu_tm = tune(TUNE_SPEECH_GLU_TM, 1) != 0 && gemm_v2_fits(EPI_GLU_F16, R, 2 * d, d, L.c2_pw1.as<float>(), &cg);
      HIP_TRY(launch_gemm_tn(EPI_GLU_F16 | (2 << 8) | GEMM_IN_TM | (glu_tm ? GEMM_OUT_TM : 0), (const f16*)x, L.wf_pw1.as<f16>(),
                             L.c2_pw1.as<float>(), E->glu.p, R, 2 * d, d, d, stream, nullptr, &cg));
    } else {
      HIP_TRY(launch_gemm_tn(epi_res | mid_in, ctx, L.w_o.as<f16>(), L.b_o.as<float>(), x, R, d, d, d, stream));
      HIP_TRY(launch_layernorm(x, L.conv_ln_w.as<float>(), L.conv_ln_b.as<float>(), c.ln_eps, h, R, d, stream, tmf, x16));
      HIP_TRY(launch_gemm_tn(EPI_GLU_F16 | (2 << 8) | ffn_in, h, L.w_pw1.as<f16>(), nullptr, E->glu.p, R, 2 * d, d, d, stream));
    }
    HIP_TRY(launch_dwconv_bn_silu(E->glu.as<f16>(), dcu, L.w_dw.as<float>(), L.bn_scale.as<float>(), L.bn_shift.as<float>(),
                                  E->dw.as<f16>(), n, tm, d, c.conv_kernel, stream, E->mid_tm, glu_tm));
    // x += 0.5 * FFN2(LN(x))
    if (xtm) {
      HIP_TRY(launch_gemm_tn(EPI_RESID_F16 | io_tm, E->dw.as<f16>(), L.w_pw2.as<f16>(), nullptr, x, R, d, d, d, stream, nullptr,
                             &produce));
      const GemmLnFold cf = consume(L.c1_ffn2);
      HIP_TRY(launch_gemm_tn(EPI_SILU_F16 | io_tm, (const f16*)x, L.wf_ffn2.as<f16>(), L.c2_ffn2.as<float>(), big, R, f, d, f,
                             stream, nullptr, &cf));
      HIP_TRY(launch_gemm_tn(EPI_RESID_HALF_F16 | io_tm, big, L.ffn2_w2.as<f16>(), L.ffn2_b2.as<float>(), x, R, d, f, d, stream));
    } else {
      HIP_TRY(launch_gemm_tn(epi_res | mid_in, E->dw.as<f16>(), L.w_pw2.as<f16>(), nullptr, x, R, d, d, d, stream));
      HIP_TRY(launch_layernorm(x, L.ffn2_ln_w.as<float>(), L.ffn2_ln_b.as<float>(), c.ln_eps, h, R, d, stream, tmf, x16));
      HIP_TRY(launch_gemm_tn(EPI_SILU_F16 | ffn_io, h, L.ffn2_w1.as<f16>(), L.ffn2_b1.as<float>(), big, R, f, d, f, stream));
      HIP_TRY(launch_gemm_tn(epi_half | ffn_in, big, L.ffn2_w2.as<f16>(), L.ffn2_b2.as<float>(), x, R, d, f, d,
                             stream));
    }
    // x = LN_block(x); h = next block's ffn1 LN, or the model-level LayerNorm after the last block
    const bool last = l + 1 == c.num_layers;
    const float* w2 = last ? E->ln_w.as<float>() : E->layers[l + 1].ffn1_ln_w.as<float>();
    const float* b2 = last ? E->ln_b.as<float>() : E->layers[l + 1].ffn1_ln_b.as<float>();
    // (the last block's h is the encoder output the pooler reads row-major)
    HIP_TRY(launch_ln2(x, L.ln_w.as<float>(), L.ln_b.as<float>(), w2, b2, c.ln_eps, h, R, d, stream, last ? 0 : tmf, x16, xtm));
  }
  // ---- attention pooler: h now holds the encoder output (fp16) ----
  float* xq = E->xq.as<float>();
  f16* hq = E->hq.as<f16>();
  HIP_TRY(launch_broadcast_row(E->pool_q0.as<float>(), xq, n, d, stream));
  HIP_TRY(launch_f32_to_f16(xq, hq, (size_t)n * d, stream));
  for (int l = 0; l < c.pooler_layers; ++l) {
    PoolLayer& L = E->pooler[l];
    // self-attention over a single token == output_proj(v_proj(x))
    HIP_TRY(launch_gemm_tn(EPI_BIAS_F16, hq, L.sv_w.as<f16>(), L.sv_b.as<float>(), E->pv.p, NP, d, d, d, stream));
    HIP_TRY(launch_gemm_tn(EPI_RESID_F32, E->pv.as<f16>(), L.so_w.as<f16>(), L.so_b.as<float>(), xq, NP, d, d, d, stream));
    HIP_TRY(launch_ln2(xq, L.sln_w.as<float>(), L.sln_b.as<float>(), nullptr, nullptr, c.ln_eps, hq, n, d, stream));
    // cross-attention over the clip's frames
    HIP_TRY(launch_gemm_tn(EPI_BIAS_F16, hq, L.cq_w.as<f16>(), L.cq_b.as<float>(), E->pq.p, NP, d, d, d, stream));
    HIP_TRY(launch_gemm_tn(EPI_BIAS_F16, h, L.ckv_w.as<f16>(), L.ckv_b.as<float>(), E->pkv.p, R, 2 * d, d, 2 * d, stream));
    HIP_TRY(launch_pool_attention(E->pq.as<f16>(), E->pkv.as<f16>(), dcu, E->pctx.as<f16>(), n, d, c.pooler_heads, stream));
    HIP_TRY(launch_gemm_tn(EPI_RESID_F32, E->pctx.as<f16>(), L.co_w.as<f16>(), L.co_b.as<float>(), xq, NP, d, d, d, stream));
    HIP_TRY(launch_ln2(xq, L.cln_w.as<float>(), L.cln_b.as<float>(), nullptr, nullptr, c.ln_eps, hq, n, d, stream));
    // FFN (ReLU)
    HIP_TRY(launch_gemm_tn(EPI_RELU_F16, hq, L.f1_w.as<f16>(), L.f1_b.as<float>(), E->pffn.p, NP, pf, d, pf, stream));
    HIP_TRY(launch_gemm_tn(EPI_RESID_F32, E->pffn.as<f16>(), L.f2_w.as<f16>(), L.f2_b.as<float>(), xq, NP, d, pf, d, stream));
    HIP_TRY(launch_ln2(xq, L.fln_w.as<float>(), L.fln_b.as<float>(), nullptr, nullptr, c.ln_eps, hq, n, d, stream));
  }
  HIP_TRY(launch_gemm_tn(EPI_STORE_F32, hq, E->pool_out_w.as<f16>(), nullptr, E->pout.p, NP, d, d, d, stream));
  if (out_dtype == SMI_F32)
    HIP_TRY(hipMemcpyAsync(out_emb, E->pout.p, (size_t)n * d * 4, hipMemcpyDeviceToDevice, stream));
  else
    HIP_TRY(launch_f32_to_f16(E->pout.as<float>(), (f16*)out_emb, (size_t)n * d, stream));
  return SMI_OK;
}

}  // extern "C"
