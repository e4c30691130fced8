// Text encoder on the generic-dimension kernels (flex.hip): every SonarTextEncoderConfig the reference's factory
// accepts that the MFMA engines do not tile for -- model_dim / head counts other than heads x 64, `normalize_before`,
// `layernorm_embedding`, learned or absent positions, and the attention pooler with embedding_dim != model_dim
// (sonar/models/sonar_text/factory.py:72-226, sonar/nn/encoder_pooler.py:49-95).  fp32 activations and weights, the
// padded [n, s] layout of the reference (no packing), one launch per reference module.
#include <vector>

#include "api_common.hpp"

using namespace smi;
using namespace smi_host;

namespace smi_host {

struct FlexLayer {
  DevBuf ln1_w, ln1_b, w_qkv, b_qkv, w_o, b_o, ln2_w, ln2_b, w_1, b_1, w_2, b_2;
};
struct FlexPoolerLayer {
  DevBuf ln1_w, ln1_b, sv_w, sv_b, so_w, so_b;
  DevBuf ln2_w, ln2_b, cq_w, cq_b, ck_w, ck_b, cv_w, cv_b, co_w, co_b;
  DevBuf ln3_w, ln3_b, f1_w, f1_b, f2_w, f2_b;
};

struct FlexEncoder {
  smi_text_encoder_config cfg;
  int edim = 0;  // sentence-vector width
  DevBuf embed, pos, lnf_w, lnf_b, lne_w, lne_b, lnemb_w, lnemb_b;
  std::vector<FlexLayer> layers;
  DevBuf pq, lnp_w, lnp_b, proj_w, proj_b;
  std::vector<FlexPoolerLayer> pooler;
  // workspace (grow-only)
  DevBuf x, h, qkv, ctx, ffn, lens, xq, hq, t1, t2, kx, vx, out32;
  int64_t weight_bytes = 0;
  int64_t bytes() const {
    return weight_bytes + (int64_t)(x.bytes + h.bytes + qkv.bytes + ctx.bytes + ffn.bytes + kx.bytes + vx.bytes);
  }
};

bool flex_encoder_wanted(const smi_text_encoder_config& c) {
  const bool fast_shape = c.model_dim > 0 && c.num_heads > 0 && c.model_dim == c.num_heads * 64 && c.model_dim % 256 == 0 &&
                          (c.model_dim / 256 <= 4 || c.model_dim == 2048) && c.ffn_inner_dim > 0 && c.ffn_inner_dim % 128 == 0;
  return !fast_shape || c.pooling == SMI_POOL_ATTENTION || (c.flags & (SMI_ENC_NORMALIZE_BEFORE | SMI_ENC_LAYERNORM_EMBEDDING | SMI_ENC_NO_POSITIONS)) ||
         (c.embedding_dim && c.embedding_dim != c.model_dim);
}

int flex_encoder_create(const smi_text_encoder_config* cfg, const smi_text_encoder_weights* w, FlexEncoder** out) {
  const smi_text_encoder_config& c = *cfg;
  if (c.model_dim <= 0 || c.num_heads <= 0 || c.model_dim % c.num_heads)
    return fail(SMI_ERR_INVALID_ARG, "model_dim %d must be a positive multiple of num_heads %d", c.model_dim, c.num_heads);
  if (c.model_dim / c.num_heads > 256) return fail(SMI_ERR_UNSUPPORTED, "head_dim %d > 256 is not covered", c.model_dim / c.num_heads);
  if (c.ffn_inner_dim <= 0 || c.num_layers < 0 || c.vocab_size <= 0 || c.max_seq_len <= 0 || c.pos_offset < 0)
    return fail(SMI_ERR_INVALID_ARG, "bad ffn_inner_dim/num_layers/vocab_size/max_seq_len/pos_offset");
  if (c.pooling < SMI_POOL_MEAN || c.pooling > SMI_POOL_ATTENTION) return fail(SMI_ERR_INVALID_ARG, "bad pooling %d", c.pooling);
  const int64_t d = c.model_dim, f = c.ffn_inner_dim;
  const int64_t E = c.embedding_dim > 0 ? c.embedding_dim : d;
  const bool attn = c.pooling == SMI_POOL_ATTENTION;
  if (!attn && E != d) return fail(SMI_ERR_INVALID_ARG, "embedding_dim %d != model_dim %d needs attention pooling", (int)E, (int)d);
  if (attn) {
    if (c.pooler_heads <= 0 || E % c.pooler_heads || E / c.pooler_heads > 256 || c.pooler_ffn_dim <= 0 || c.pooler_layers < 0)
      return fail(SMI_ERR_INVALID_ARG, "bad attention pooler shape (layers %d, heads %d, ffn %d, embedding_dim %d)",
                  c.pooler_layers, c.pooler_heads, c.pooler_ffn_dim, (int)E);
    if (c.pooler_layers > 0 && !w->pooler) return fail(SMI_ERR_INVALID_ARG, "null pooler layers");
  }
  if (c.num_layers > 0 && !w->layers) return fail(SMI_ERR_INVALID_ARG, "null layers");
  FlexEncoder* e = new FlexEncoder();
  e->cfg = c;
  e->edim = (int)E;
  int rc = SMI_OK;
  auto up = [&](const smi_tensor& t, int64_t numel, DevBuf& dst, const char* name) {
    if (rc == SMI_OK) rc = upload(t, numel, false, dst, name);
    if (rc == SMI_OK) e->weight_bytes += (int64_t)dst.bytes;
  };
  up(w->embed, c.vocab_size * d, e->embed, "embed");
  if (!(c.flags & SMI_ENC_NO_POSITIONS)) up(w->pos_table, (int64_t)(c.max_seq_len + c.pos_offset) * d, e->pos, "pos_table");
  up(w->final_layer_norm_w, d, e->lnf_w, "layer_norm.weight");
  up(w->final_layer_norm_b, d, e->lnf_b, "layer_norm.bias");
  if (c.flags & SMI_ENC_NORMALIZE_BEFORE) {
    up(w->encoder_layer_norm_w, d, e->lne_w, "encoder.layer_norm.weight");
    up(w->encoder_layer_norm_b, d, e->lne_b, "encoder.layer_norm.bias");
  }
  if (c.flags & SMI_ENC_LAYERNORM_EMBEDDING) {
    up(w->embed_layer_norm_w, d, e->lnemb_w, "encoder_frontend.layer_norm.weight");
    up(w->embed_layer_norm_b, d, e->lnemb_b, "encoder_frontend.layer_norm.bias");
  }
  e->layers.resize(c.num_layers);
  for (int l = 0; l < c.num_layers && rc == SMI_OK; ++l) {
    const smi_text_encoder_layer& s = w->layers[l];
    FlexLayer& L = e->layers[l];
    up(s.self_attn_layer_norm_w, d, L.ln1_w, "self_attn_layer_norm.weight");
    up(s.self_attn_layer_norm_b, d, L.ln1_b, "self_attn_layer_norm.bias");
    up(s.ffn_layer_norm_w, d, L.ln2_w, "ffn_layer_norm.weight");
    up(s.ffn_layer_norm_b, d, L.ln2_b, "ffn_layer_norm.bias");
    up(s.out_w, d * d, L.w_o, "output_proj.weight");
    up(s.out_b, d, L.b_o, "output_proj.bias");
    up(s.ffn_inner_w, f * d, L.w_1, "ffn.inner_proj.weight");
    up(s.ffn_inner_b, f, L.b_1, "ffn.inner_proj.bias");
    up(s.ffn_out_w, d * f, L.w_2, "ffn.output_proj.weight");
    up(s.ffn_out_b, d, L.b_2, "ffn.output_proj.bias");
    DevBuf tq, tk, tv, bq, bk, bv;
    up(s.q_w, d * d, tq, "q_proj.weight");
    up(s.k_w, d * d, tk, "k_proj.weight");
    up(s.v_w, d * d, tv, "v_proj.weight");
    up(s.q_b, d, bq, "q_proj.bias");
    up(s.k_b, d, bk, "k_proj.bias");
    up(s.v_b, d, bv, "v_proj.bias");
    if (rc == SMI_OK) {  // fused [q; k; v] projection
      const size_t wb = (size_t)d * d * 4, bb = (size_t)d * 4;
      hipError_t he = L.w_qkv.alloc(3 * wb);
      if (he == hipSuccess) he = L.b_qkv.alloc(3 * bb);
      if (he == hipSuccess) he = hipMemcpy(L.w_qkv.p, tq.p, wb, hipMemcpyDeviceToDevice);
      if (he == hipSuccess) he = hipMemcpy((char*)L.w_qkv.p + wb, tk.p, wb, hipMemcpyDeviceToDevice);
      if (he == hipSuccess) he = hipMemcpy((char*)L.w_qkv.p + 2 * wb, tv.p, wb, hipMemcpyDeviceToDevice);
      if (he == hipSuccess) he = hipMemcpy(L.b_qkv.p, bq.p, bb, hipMemcpyDeviceToDevice);
      if (he == hipSuccess) he = hipMemcpy((char*)L.b_qkv.p + bb, bk.p, bb, hipMemcpyDeviceToDevice);
      if (he == hipSuccess) he = hipMemcpy((char*)L.b_qkv.p + 2 * bb, bv.p, bb, hipMemcpyDeviceToDevice);
      if (he != hipSuccess) rc = fail(he == hipErrorOutOfMemory ? SMI_ERR_OOM : SMI_ERR_HIP, "packing qkv: %s", hipGetErrorString(he));
    }
  }
  if (attn) {
    const int64_t pf = c.pooler_ffn_dim;
    up(w->pooler_query, E, e->pq, "pooler query");
    up(w->pooler_proj_w, E * E, e->proj_w, "pooler.projection_out.weight");
    up(w->pooler_proj_b, E, e->proj_b, "pooler.projection_out.bias");
    if (c.flags & SMI_ENC_NORMALIZE_BEFORE) {
      up(w->pooler_layer_norm_w, E, e->lnp_w, "pooler.decoder.layer_norm.weight");
      up(w->pooler_layer_norm_b, E, e->lnp_b, "pooler.decoder.layer_norm.bias");
    }
    e->pooler.resize(c.pooler_layers);
    for (int l = 0; l < c.pooler_layers && rc == SMI_OK; ++l) {
      const smi_text_pooler_layer& s = w->pooler[l];
      FlexPoolerLayer& L = e->pooler[l];
      up(s.self_attn_layer_norm_w, E, L.ln1_w, "pooler self_attn_layer_norm.weight");
      up(s.self_attn_layer_norm_b, E, L.ln1_b, "pooler self_attn_layer_norm.bias");
      up(s.self_v_w, E * E, L.sv_w, "pooler self_attn.v_proj.weight");
      up(s.self_v_b, E, L.sv_b, "pooler self_attn.v_proj.bias");
      up(s.self_out_w, E * E, L.so_w, "pooler self_attn.output_proj.weight");
      up(s.self_out_b, E, L.so_b, "pooler self_attn.output_proj.bias");
      up(s.cross_layer_norm_w, E, L.ln2_w, "pooler encoder_decoder_attn_layer_norm.weight");
      up(s.cross_layer_norm_b, E, L.ln2_b, "pooler encoder_decoder_attn_layer_norm.bias");
      up(s.cross_q_w, E * E, L.cq_w, "pooler encoder_decoder_attn.q_proj.weight");
      up(s.cross_q_b, E, L.cq_b, "pooler encoder_decoder_attn.q_proj.bias");
      up(s.cross_k_w, E * d, L.ck_w, "pooler encoder_decoder_attn.k_proj.weight");
      up(s.cross_k_b, E, L.ck_b, "pooler encoder_decoder_attn.k_proj.bias");
      up(s.cross_v_w, E * d, L.cv_w, "pooler encoder_decoder_attn.v_proj.weight");
      up(s.cross_v_b, E, L.cv_b, "pooler encoder_decoder_attn.v_proj.bias");
      up(s.cross_out_w, E * E, L.co_w, "pooler encoder_decoder_attn.output_proj.weight");
      up(s.cross_out_b, E, L.co_b, "pooler encoder_decoder_attn.output_proj.bias");
      up(s.ffn_layer_norm_w, E, L.ln3_w, "pooler ffn_layer_norm.weight");
      up(s.ffn_layer_norm_b, E, L.ln3_b, "pooler ffn_layer_norm.bias");
      up(s.ffn_inner_w, pf * E, L.f1_w, "pooler ffn.inner_proj.weight");
      up(s.ffn_inner_b, pf, L.f1_b, "pooler ffn.inner_proj.bias");
      up(s.ffn_out_w, E * pf, L.f2_w, "pooler ffn.output_proj.weight");
      up(s.ffn_out_b, E, L.f2_b, "pooler ffn.output_proj.bias");
    }
  }
  if (rc != SMI_OK) {
    delete e;
    return rc;
  }
  *out = e;
  return SMI_OK;
}

void flex_encoder_destroy(FlexEncoder* e) { delete e; }
int64_t flex_encoder_bytes(const FlexEncoder* e) { return e ? e->bytes() : 0; }
int flex_encoder_embedding_dim(const FlexEncoder* e) { return e->edim; }

int flex_encoder_forward(FlexEncoder* e, const int64_t* ids, const int32_t* seq_lens, int n, int s, void* out_emb,
                         void* out_encoded, int out_dtype, int32_t* bad_ids_dev, hipStream_t stream) {
  const smi_text_encoder_config& c = e->cfg;
  const int d = c.model_dim, f = c.ffn_inner_dim, E = e->edim, heads = c.num_heads, hd = d / heads;
  const int rows = n * s;
  const bool pre = (c.flags & SMI_ENC_NORMALIZE_BEFORE) != 0;
  HIP_TRY(e->x.reserve((size_t)rows * d * 4));
  HIP_TRY(e->h.reserve((size_t)rows * d * 4));
  HIP_TRY(e->qkv.reserve((size_t)rows * 3 * d * 4));
  HIP_TRY(e->ctx.reserve((size_t)rows * d * 4));
  HIP_TRY(e->ffn.reserve((size_t)rows * f * 4));
  HIP_TRY(e->lens.reserve((size_t)n * 4));
  HIP_TRY(e->out32.reserve((size_t)n * E * 4));
  {
    std::vector<int32_t> hl(n);
    for (int i = 0; i < n; ++i) {
      hl[i] = seq_lens ? seq_lens[i] : s;
      if (hl[i] < 0 || hl[i] > s) return fail(SMI_ERR_INVALID_ARG, "seq_lens[%d]=%d outside [0,%d]", i, hl[i], s);
    }
    HIP_TRY(hipMemcpyAsync(e->lens.p, hl.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));  // `hl` is pageable host memory that dies with this scope
  }
  float* x = e->x.as<float>();
  float* h = e->h.as<float>();
  float* qkv = e->qkv.as<float>();
  float* ctx = e->ctx.as<float>();
  float* ffn = e->ffn.as<float>();
  const int32_t* lens = e->lens.as<int32_t>();
  HIP_TRY(launch_flex_embed(ids, nullptr, e->embed.as<float>(), e->pos.p ? e->pos.as<float>() : nullptr, c.embed_scale, x,
                            rows, d, s, c.pos_offset, -1, c.vocab_size, bad_ids_dev, stream));
  if (c.flags & SMI_ENC_LAYERNORM_EMBEDDING)
    HIP_TRY(launch_flex_layernorm(x, e->lnemb_w.as<float>(), e->lnemb_b.as<float>(), c.ln_eps, x, rows, d, stream));
  for (auto& L : e->layers) {  // StandardTransformerEncoderLayer, norm_order PRE (factory.py:122-128)
    HIP_TRY(launch_flex_layernorm(x, L.ln1_w.as<float>(), L.ln1_b.as<float>(), c.ln_eps, h, rows, d, stream));
    HIP_TRY(launch_flex_linear(h, d, L.w_qkv.as<float>(), L.b_qkv.as<float>(), qkv, 3 * d, rows, 3 * d, d, 0, nullptr, 0, stream));
    HIP_TRY(launch_flex_attention(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, ctx, d, n, s, s, lens, heads, hd, 0, stream));
    HIP_TRY(launch_flex_linear(ctx, d, L.w_o.as<float>(), L.b_o.as<float>(), x, d, rows, d, d, 0, x, d, stream));
    HIP_TRY(launch_flex_layernorm(x, L.ln2_w.as<float>(), L.ln2_b.as<float>(), c.ln_eps, h, rows, d, stream));
    HIP_TRY(launch_flex_linear(h, d, L.w_1.as<float>(), L.b_1.as<float>(), ffn, f, rows, f, d, 1, nullptr, 0, stream));
    HIP_TRY(launch_flex_linear(ffn, f, L.w_2.as<float>(), L.b_2.as<float>(), x, d, rows, d, f, 0, x, d, stream));
  }
  if (pre) HIP_TRY(launch_flex_layernorm(x, e->lne_w.as<float>(), e->lne_b.as<float>(), c.ln_eps, x, rows, d, stream));
  HIP_TRY(launch_flex_layernorm(x, e->lnf_w.as<float>(), e->lnf_b.as<float>(), c.ln_eps, x, rows, d, stream));
  if (out_encoded) HIP_TRY(launch_flex_store_encoded(x, lens, n, s, d, out_encoded, out_dtype == SMI_F16, stream));

  float* out32 = e->out32.as<float>();
  if (c.pooling != SMI_POOL_ATTENTION) {
    HIP_TRY(launch_flex_pool(x, lens, c.pooling, out32, n, s, d, stream));
  } else {
    // AttentionEncoderOutputPooler (encoder_pooler.py:49-95): ONE query token per sentence through decoder layers that
    // cross-attend to the encoder output, then projection_out
    const int ph = c.pooler_heads, phd = E / ph, pf = c.pooler_ffn_dim;
    HIP_TRY(e->xq.reserve((size_t)n * E * 4));
    HIP_TRY(e->hq.reserve((size_t)n * E * 4));
    HIP_TRY(e->t1.reserve((size_t)n * std::max(E, pf) * 4));
    HIP_TRY(e->t2.reserve((size_t)n * E * 4));
    HIP_TRY(e->kx.reserve((size_t)rows * E * 4));
    HIP_TRY(e->vx.reserve((size_t)rows * E * 4));
    float* xq = e->xq.as<float>();
    float* hq = e->hq.as<float>();
    float* t1 = e->t1.as<float>();
    float* t2 = e->t2.as<float>();
    float* kx = e->kx.as<float>();
    float* vx = e->vx.as<float>();
    HIP_TRY(hipMemsetAsync(xq, 0, (size_t)n * E * 4, stream));
    HIP_TRY(launch_flex_add_rows(xq, e->pq.as<float>(), n, E, n, stream));  // every sentence starts from the same query
    auto ln = [&](const DevBuf& w_, const DevBuf& b_, const float* src, float* dst) {
      return launch_flex_layernorm(src, w_.as<float>(), b_.as<float>(), c.ln_eps, dst, n, E, stream);
    };
    for (auto& L : e->pooler) {
      // self-attention over one token: W_o (W_v u + b_v) + b_o
      const float* u = xq;
      if (pre) {
        HIP_TRY(ln(L.ln1_w, L.ln1_b, xq, hq));
        u = hq;
      }
      HIP_TRY(launch_flex_linear(u, E, L.sv_w.as<float>(), L.sv_b.as<float>(), t1, E, n, E, E, 0, nullptr, 0, stream));
      HIP_TRY(launch_flex_linear(t1, E, L.so_w.as<float>(), L.so_b.as<float>(), xq, E, n, E, E, 0, xq, E, stream));
      if (!pre) HIP_TRY(ln(L.ln1_w, L.ln1_b, xq, xq));
      // cross-attention to the encoder output (keys = the sentence's valid positions)
      u = xq;
      if (pre) {
        HIP_TRY(ln(L.ln2_w, L.ln2_b, xq, hq));
        u = hq;
      }
      HIP_TRY(launch_flex_linear(u, E, L.cq_w.as<float>(), L.cq_b.as<float>(), t1, E, n, E, E, 0, nullptr, 0, stream));
      HIP_TRY(launch_flex_linear(x, d, L.ck_w.as<float>(), L.ck_b.as<float>(), kx, E, rows, E, d, 0, nullptr, 0, stream));
      HIP_TRY(launch_flex_linear(x, d, L.cv_w.as<float>(), L.cv_b.as<float>(), vx, E, rows, E, d, 0, nullptr, 0, stream));
      HIP_TRY(launch_flex_attention(t1, E, kx, vx, E, t2, E, n, 1, s, lens, ph, phd, 0, stream));
      HIP_TRY(launch_flex_linear(t2, E, L.co_w.as<float>(), L.co_b.as<float>(), xq, E, n, E, E, 0, xq, E, stream));
      if (!pre) HIP_TRY(ln(L.ln2_w, L.ln2_b, xq, xq));
      // feed-forward
      u = xq;
      if (pre) {
        HIP_TRY(ln(L.ln3_w, L.ln3_b, xq, hq));
        u = hq;
      }
      HIP_TRY(launch_flex_linear(u, E, L.f1_w.as<float>(), L.f1_b.as<float>(), t1, pf, n, pf, E, 1, nullptr, 0, stream));
      HIP_TRY(launch_flex_linear(t1, pf, L.f2_w.as<float>(), L.f2_b.as<float>(), xq, E, n, E, pf, 0, xq, E, stream));
      if (!pre) HIP_TRY(ln(L.ln3_w, L.ln3_b, xq, xq));
    }
    if (pre) HIP_TRY(ln(e->lnp_w, e->lnp_b, xq, xq));
    HIP_TRY(launch_flex_linear(xq, E, e->proj_w.as<float>(), e->proj_b.as<float>(), out32, E, n, E, E, 0, nullptr, 0, stream));
  }
  if (out_dtype == SMI_F32)
    HIP_TRY(hipMemcpyAsync(out_emb, out32, (size_t)n * E * 4, hipMemcpyDeviceToDevice, stream));
  else
    HIP_TRY(launch_f32_to_f16(out32, (f16*)out_emb, (size_t)n * E, stream));
  return SMI_OK;
}

}  // namespace smi_host
