// C-ABI implementation (see include/sonar_mi355.h).  Host-side runtime of the
// engine: weight packing into the fp16/fp32 HBM layout the kernels want,
// stream-ordered workspace, and the per-layer launch schedule of the text
// encoder forward pass (reference op order: sonar/models/sonar_text/model.py:130-143
// + factory.py:102-153, pre-LN layers, model-level final LayerNorm, pooling).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "api_common.hpp"

using namespace smi;

namespace smi {
std::atomic<int64_t> g_tune[TUNE_COUNT];  // tuning.hpp: zero-initialised = every switch unset
}

namespace smi_host {

std::string& last_error() {
  thread_local std::string err;
  return err;
}

int upload(const smi_tensor& t, int64_t expect_numel, bool want_f16, DevBuf& dst, const char* name,
           int64_t pad_numel) {
  if (!t.data) return fail(SMI_ERR_INVALID_ARG, "weight %s: null data", name);
  if (t.numel != expect_numel)
    return fail(SMI_ERR_INVALID_ARG, "weight %s: numel %lld, expected %lld", name,
                (long long)t.numel, (long long)expect_numel);
  if (t.dtype != SMI_F32 && t.dtype != SMI_F16)
    return fail(SMI_ERR_INVALID_ARG, "weight %s: bad dtype %d", name, t.dtype);
  const size_t n = (size_t)t.numel;
  const size_t total = std::max<size_t>(n, (size_t)pad_numel);
  const size_t src_es = t.dtype == SMI_F32 ? 4 : 2;
  const size_t dst_es = want_f16 ? 2 : 4;
  HIP_TRY(dst.alloc(total * dst_es));
  if (total > n) HIP_TRY(hipMemset((char*)dst.p + n * dst_es, 0, (total - n) * dst_es));
  const bool same = (t.dtype == SMI_F16) == want_f16;
  const hipMemcpyKind kind = t.on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (same) {
    HIP_TRY(hipMemcpy(dst.p, t.data, n * src_es, kind));
    return SMI_OK;
  }
  // convert through a bounded staging buffer (the embedding table is ~1 GB fp32)
  const size_t chunk = (size_t)32 << 20;  // elements
  DevBuf stage;
  if (!t.on_device) HIP_TRY(stage.alloc(std::min(n, chunk) * src_es));
  for (size_t off = 0; off < n; off += chunk) {
    const size_t m = std::min(chunk, n - off);
    const void* src = (const char*)t.data + off * src_es;
    if (!t.on_device) {
      HIP_TRY(hipMemcpy(stage.p, src, m * src_es, hipMemcpyHostToDevice));
      src = stage.p;
    }
    if (want_f16)
      HIP_TRY(launch_f32_to_f16((const float*)src, dst.as<f16>() + off, m, nullptr));
    else
      HIP_TRY(launch_f16_to_f32((const f16*)src, dst.as<float>() + off, m, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
  }
  return SMI_OK;
}

}  // namespace smi_host

using namespace smi_host;

namespace {

struct Layer {
  DevBuf ln1_w, ln1_b, w_qkv, b_qkv, w_o, b_o, ln2_w, ln2_b, w_1, b_1, w_2, b_2;
  // LayerNorm fold (kernels.hpp: GemmLnFold): the projections that follow a LayerNorm with the LayerNorm weight
  // multiplied into their columns, and the two per-column constants of the fold
  DevBuf w_qkv_f, c1_qkv, c2_qkv, w_1_f, c1_1, c2_1;
};

constexpr int kCuRing = 8;

}  // namespace

struct smi_text_encoder {
  smi_text_encoder_config cfg;
  smi_host::FlexEncoder* flex = nullptr;  // set: the model runs on the generic-dimension kernels (flex_encoder.hip)
  DevBuf embed, pos, lnf_w, lnf_b;
  std::vector<Layer> layers;
  // workspace (capacity in packed+padded token rows)
  int64_t cap_rows = 0;
  DevBuf x, h, qkv, ctx, ffn;
  DevBuf parts;  // split-K slabs of the attention-output / FFN-output projections (small batches only; fp32, or fp16 on an fp16 stream)
  DevBuf rowpart;  // LayerNorm fold: float2 [d / 256][rows] partial (sum, sum of squares) of the residual rows
  bool lnfold = false;  // the folded weights exist (tile-major fp16 configuration, d = 1024)
  bool lnfold_centered = false;  // ... and their rows are centred (no "- mean * c1" term in the epilogue)
  int num_cus = 256;
  // cu_seqlens staging ring: pinned host + device copies
  int32_t* h_cu[kCuRing] = {};
  DevBuf d_cu[kCuRing];
  hipEvent_t cu_ev[kCuRing] = {};
  int64_t cu_cap = 0;
  int cu_next = 0;
  int64_t weight_bytes = 0;
  // out-of-vocabulary token ids: the embedding kernel raises this flag (host-mapped, device-visible);
  // it is reported -- and cleared -- by smi_text_encoder_status() only (sticky until then); the Python model object
  // calls that after every forward() unless its caller defers the check to the end of a queue of batches
  int32_t* bad_ids = nullptr;      // pinned host word
  int32_t* bad_ids_dev = nullptr;  // its device address
  // every GEMM operand (weights, h, ctx, ffn) in the tile-major layout of common.hpp
  bool tile_major = false;
  bool x16 = false;  // SMI_ENC_FP16_RESIDUAL: the residual stream x is fp16
  // optional per-launch event timing
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  struct ProfRec { int slot; hipEvent_t a, b; };
  std::vector<ProfRec> prof;
  size_t ev_used = 0;

  ~smi_text_encoder() {
    if (flex) smi_host::flex_encoder_destroy(flex);
    for (hipEvent_t ev : ev_pool) (void)hipEventDestroy(ev);
    if (bad_ids) (void)hipHostFree(bad_ids);
    for (int i = 0; i < kCuRing; ++i) {
      if (h_cu[i]) (void)hipHostFree(h_cu[i]);
      if (cu_ev[i]) (void)hipEventDestroy(cu_ev[i]);
    }
  }
};

namespace {

int ensure_cu(smi_text_encoder* e, int64_t n) {
  if (n + 1 <= e->cu_cap) return SMI_OK;
  HIP_TRY(hipDeviceSynchronize());
  const int64_t cap = std::max<int64_t>(n + 1, 4096);
  for (int i = 0; i < kCuRing; ++i) {
    if (e->h_cu[i]) (void)hipHostFree(e->h_cu[i]);
    e->h_cu[i] = nullptr;
    HIP_TRY(hipHostMalloc((void**)&e->h_cu[i], cap * sizeof(int32_t), hipHostMallocDefault));
    HIP_TRY(e->d_cu[i].alloc(cap * sizeof(int32_t)));
    if (!e->cu_ev[i]) HIP_TRY(hipEventCreateWithFlags(&e->cu_ev[i], hipEventDisableTiming));
  }
  e->cu_cap = cap;
  return SMI_OK;
}

int ensure_workspace(smi_text_encoder* e, int64_t rows) {
  if (rows <= e->cap_rows) return SMI_OK;
  HIP_TRY(hipDeviceSynchronize());
  const int64_t d = e->cfg.model_dim, f = e->cfg.ffn_inner_dim;
  e->cap_rows = 0;
  HIP_TRY(e->x.alloc((size_t)rows * d * (e->x16 ? 2 : 4)));
  HIP_TRY(e->h.alloc((size_t)rows * d * 2));
  HIP_TRY(e->qkv.alloc((size_t)rows * 3 * d * 2));
  HIP_TRY(e->ctx.alloc((size_t)rows * d * 2));
  HIP_TRY(e->ffn.alloc((size_t)rows * f * 2));
  // rows that no kernel writes (tile padding) must hold finite values
  HIP_TRY(hipMemset(e->x.p, 0, e->x.bytes));
  HIP_TRY(hipMemset(e->h.p, 0, e->h.bytes));
  HIP_TRY(hipMemset(e->qkv.p, 0, e->qkv.bytes));
  HIP_TRY(hipMemset(e->ctx.p, 0, e->ctx.bytes));
  HIP_TRY(hipMemset(e->ffn.p, 0, e->ffn.bytes));
  e->cap_rows = rows;
  return SMI_OK;
}

hipEvent_t prof_event(smi_text_encoder* e) {
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    e->ev_pool.push_back(ev);
  }
  return e->ev_pool[e->ev_used++];
}

// Brackets one launch with events when profiling is on.
struct ProfScope {
  smi_text_encoder* e;
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  int slot;
  ProfScope(smi_text_encoder* e_, int slot_, hipStream_t s_) : e(e_), s(s_), slot(slot_) {
    if (e->profiling && (a = prof_event(e)) && (b = prof_event(e))) (void)hipEventRecord(a, s);
  }
  ~ProfScope() {
    if (a && b) {
      (void)hipEventRecord(b, s);
      e->prof.push_back({slot, a, b});
    }
  }
};

int check_cfg(const smi_text_encoder_config& c) {
  if (c.model_dim <= 0 || c.num_heads <= 0 || c.model_dim != c.num_heads * 64)
    return fail(SMI_ERR_UNSUPPORTED, "model_dim %d must equal num_heads %d * 64", c.model_dim,
                c.num_heads);
  if (c.model_dim % 256 || (c.model_dim / 256 > 4 && c.model_dim != 2048))
    return fail(SMI_ERR_UNSUPPORTED, "model_dim %d must be 256/512/768/1024/2048", c.model_dim);
  if (c.ffn_inner_dim <= 0 || c.ffn_inner_dim % 128)
    return fail(SMI_ERR_UNSUPPORTED, "ffn_inner_dim %d must be a multiple of 128", c.ffn_inner_dim);
  if (c.num_layers < 0 || c.vocab_size <= 0 || c.max_seq_len <= 0 || c.pos_offset < 0)
    return fail(SMI_ERR_INVALID_ARG, "bad num_layers/vocab_size/max_seq_len/pos_offset");
  if (c.pooling < SMI_POOL_MEAN || c.pooling > SMI_POOL_LAST)
    return fail(SMI_ERR_INVALID_ARG, "bad pooling %d", c.pooling);
  return SMI_OK;
}

}  // namespace

extern "C" {

const char* smi_version(void) { return "sonar_mi355 0.1.0 (gfx950)"; }

// ---- tuning registry (tuning.hpp): the only way a switch reaches the library ----
int smi_tuning_set(const char* name, int32_t value) {
  const int i = tune_index(name);
  if (i < 0) return fail(SMI_ERR_INVALID_ARG, "unknown tuning switch '%s'", name ? name : "(null)");
  g_tune[i].store((1ll << 40) | (int64_t)(uint32_t)value, std::memory_order_relaxed);
  return SMI_OK;
}
int smi_tuning_unset(const char* name) {
  const int i = tune_index(name);
  if (i < 0) return fail(SMI_ERR_INVALID_ARG, "unknown tuning switch '%s'", name ? name : "(null)");
  g_tune[i].store(0, std::memory_order_relaxed);
  return SMI_OK;
}
int smi_tuning_get(const char* name, int32_t* value, int32_t* is_set) {
  const int i = tune_index(name);
  if (i < 0 || !value || !is_set) return fail(SMI_ERR_INVALID_ARG, "unknown tuning switch '%s'", name ? name : "(null)");
  const int64_t v = g_tune[i].load(std::memory_order_relaxed);
  *is_set = v != 0;
  *value = v ? (int32_t)(uint32_t)(v & 0xffffffffll) : 0;
  return SMI_OK;
}
const char* smi_tuning_name(int32_t index) { return tune_name(index); }
int smi_abi_version(void) { return SMI_ABI_VERSION; }
const char* smi_last_error(void) { return smi_host::last_error().c_str(); }

int smi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int smi_init(int device_id) {
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(hipSetDevice(device_id));
  return SMI_OK;
}

int smi_text_encoder_create(const smi_text_encoder_config* cfg, const smi_text_encoder_weights* w,
                            int64_t max_tokens_hint, smi_text_encoder** out) {
  if (!cfg || !w || !out) return fail(SMI_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (flex_encoder_wanted(*cfg)) {  // a shape / option outside the MFMA engines' tiling: generic-dimension kernels
    if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
    FlexEncoder* fe = nullptr;
    if (int rc = flex_encoder_create(cfg, w, &fe)) return rc;
    smi_text_encoder* e = new smi_text_encoder();
    e->cfg = *cfg;
    e->flex = fe;
    *out = e;
    return SMI_OK;
  }
  if (int rc = check_cfg(*cfg)) return rc;
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  if (cfg->num_layers > 0 && !w->layers) return fail(SMI_ERR_INVALID_ARG, "null layers");

  smi_text_encoder* e = new smi_text_encoder();
  e->cfg = *cfg;
  const int64_t d = cfg->model_dim, f = cfg->ffn_inner_dim;
  int rc = SMI_OK;
  auto up = [&](const smi_tensor& t, int64_t numel, bool f16, DevBuf& dst, const char* name) {
    if (rc == SMI_OK) rc = upload(t, numel, f16, dst, name);
  };
  up(w->embed, cfg->vocab_size * d, true, e->embed, "embed");
  up(w->pos_table, (int64_t)(cfg->max_seq_len + cfg->pos_offset) * d, false, e->pos, "pos_table");
  up(w->final_layer_norm_w, d, false, e->lnf_w, "layer_norm.weight");
  up(w->final_layer_norm_b, d, false, e->lnf_b, "layer_norm.bias");
  e->layers.resize(cfg->num_layers);
  e->tile_major = f % 256 == 0;  // d % 256 == 0 always (check_cfg)
  e->x16 = (cfg->flags & SMI_ENC_FP16_RESIDUAL) != 0;
  for (int l = 0; l < cfg->num_layers && rc == SMI_OK; ++l) {
    const smi_text_encoder_layer& s = w->layers[l];
    Layer& L = e->layers[l];
    up(s.self_attn_layer_norm_w, d, false, L.ln1_w, "self_attn_layer_norm.weight");
    up(s.self_attn_layer_norm_b, d, false, L.ln1_b, "self_attn_layer_norm.bias");
    up(s.ffn_layer_norm_w, d, false, L.ln2_w, "ffn_layer_norm.weight");
    up(s.ffn_layer_norm_b, d, false, L.ln2_b, "ffn_layer_norm.bias");
    up(s.out_w, d * d, true, L.w_o, "output_proj.weight");
    up(s.out_b, d, false, L.b_o, "output_proj.bias");
    up(s.ffn_inner_w, f * d, true, L.w_1, "ffn.inner_proj.weight");
    up(s.ffn_inner_b, f, false, L.b_1, "ffn.inner_proj.bias");
    up(s.ffn_out_w, d * f, true, L.w_2, "ffn.output_proj.weight");
    up(s.ffn_out_b, d, false, L.b_2, "ffn.output_proj.bias");
    // fused [q; k; v] projection: one [3d, d] weight, one [3d] bias
    if (rc == SMI_OK) {
      DevBuf tq, tk, tv, bq, bk, bv;
      up(s.q_w, d * d, true, tq, "q_proj.weight");
      up(s.k_w, d * d, true, tk, "k_proj.weight");
      up(s.v_w, d * d, true, tv, "v_proj.weight");
      up(s.q_b, d, false, bq, "q_proj.bias");
      up(s.k_b, d, false, bk, "k_proj.bias");
      up(s.v_b, d, false, bv, "v_proj.bias");
      if (rc == SMI_OK) {
        hipError_t he = L.w_qkv.alloc((size_t)3 * d * d * 2);
        if (he == hipSuccess) he = L.b_qkv.alloc((size_t)3 * d * 4);
        const size_t wb = (size_t)d * d * 2, bb = (size_t)d * 4;
        if (he == hipSuccess) he = hipMemcpy(L.w_qkv.p, tq.p, wb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy((char*)L.w_qkv.p + wb, tk.p, wb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy((char*)L.w_qkv.p + 2 * wb, tv.p, wb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy(L.b_qkv.p, bq.p, bb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy((char*)L.b_qkv.p + bb, bk.p, bb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy((char*)L.b_qkv.p + 2 * bb, bv.p, bb, hipMemcpyDeviceToDevice);
        if (he != hipSuccess)
          rc = fail(he == hipErrorOutOfMemory ? SMI_ERR_OOM : SMI_ERR_HIP, "packing qkv: %s",
                    hipGetErrorString(he));
      }
    }
    // LayerNorm fold: W (.) g, c1, c2 for the two projections that read a LayerNorm output (the configuration that uses
    // them: fp16 tile-major residual stream, d = 1024; SMI_ENC_LNFOLD=0 keeps the LayerNorm launches for A/B runs)
    // SMI_ENC_LNFOLD: 0 = LayerNorm launches, 1 = fold with the exact "- mean * c1" epilogue term, 2 (default) = fold with
    // row-centred weights (kernels.hpp: GemmLnFold.centered)
    const int lnfold_env = tune(TUNE_ENC_LNFOLD, 2);
    e->lnfold = lnfold_env > 0 && e->tile_major && e->x16 && d == 1024;
    e->lnfold_centered = lnfold_env == 2;
    if (rc == SMI_OK && e->lnfold) {
      hipError_t he = L.w_qkv_f.alloc((size_t)3 * d * d * 2);
      if (he == hipSuccess) he = L.c1_qkv.alloc((size_t)3 * d * 4);
      if (he == hipSuccess) he = L.c2_qkv.alloc((size_t)3 * d * 4);
      if (he == hipSuccess) he = L.w_1_f.alloc((size_t)f * d * 2);
      if (he == hipSuccess) he = L.c1_1.alloc((size_t)f * 4);
      if (he == hipSuccess) he = L.c2_1.alloc((size_t)f * 4);
      if (he == hipSuccess)
        he = launch_ln_fold_prep(L.w_qkv.as<f16>(), L.ln1_w.as<float>(), L.ln1_b.as<float>(), L.b_qkv.as<float>(),
                                 L.w_qkv_f.as<f16>(), L.c1_qkv.as<float>(), L.c2_qkv.as<float>(), 3 * (int)d, (int)d,
                                 e->lnfold_centered, nullptr);
      if (he == hipSuccess)
        he = launch_ln_fold_prep(L.w_1.as<f16>(), L.ln2_w.as<float>(), L.ln2_b.as<float>(), L.b_1.as<float>(),
                                 L.w_1_f.as<f16>(), L.c1_1.as<float>(), L.c2_1.as<float>(), (int)f, (int)d,
                                 e->lnfold_centered, nullptr);
      if (he == hipSuccess) he = hipStreamSynchronize(nullptr);
      if (he != hipSuccess)
        rc = fail(he == hipErrorOutOfMemory ? SMI_ERR_OOM : SMI_ERR_HIP, "LayerNorm fold: %s", hipGetErrorString(he));
      if (rc == SMI_OK) rc = to_tile_major(L.w_qkv_f, 3 * (int)d, (int)d);
      if (rc == SMI_OK) rc = to_tile_major(L.w_1_f, (int)f, (int)d);
    }
    if (rc == SMI_OK && e->tile_major) {
      rc = to_tile_major(L.w_qkv, 3 * (int)d, (int)d);
      if (rc == SMI_OK) rc = to_tile_major(L.w_o, (int)d, (int)d);
      if (rc == SMI_OK) rc = to_tile_major(L.w_1, (int)f, (int)d);
      if (rc == SMI_OK) rc = to_tile_major(L.w_2, (int)d, (int)f);
    }
  }
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      e->num_cus = cus;
  }
  if (rc == SMI_OK && max_tokens_hint > 0) rc = ensure_workspace(e, (max_tokens_hint + 255) / 256 * 256);
  if (rc != SMI_OK) {
    delete e;
    return rc;
  }
  e->weight_bytes = (int64_t)(e->embed.bytes + e->pos.bytes + 2 * e->lnf_w.bytes);
  for (auto& L : e->layers)
    e->weight_bytes += (int64_t)(L.w_qkv.bytes + L.b_qkv.bytes + L.w_o.bytes + L.b_o.bytes +
                                 L.w_1.bytes + L.b_1.bytes + L.w_2.bytes + L.b_2.bytes +
                                 4 * L.ln1_w.bytes + L.w_qkv_f.bytes + L.w_1_f.bytes + 2 * L.c1_qkv.bytes + 2 * L.c1_1.bytes);
  *out = e;
  return SMI_OK;
}

void smi_text_encoder_destroy(smi_text_encoder* enc) {
  if (!enc) return;
  (void)hipDeviceSynchronize();
  delete enc;
}

int64_t smi_text_encoder_device_bytes(const smi_text_encoder* e) {
  if (!e) return 0;
  if (e->flex) return flex_encoder_bytes(e->flex);
  return e->weight_bytes + (int64_t)(e->x.bytes + e->h.bytes + e->qkv.bytes + e->ctx.bytes + e->ffn.bytes + e->parts.bytes);
}

int smi_text_encoder_forward(smi_text_encoder* e, const int64_t* ids, const int32_t* seq_lens,
                             int32_t n, int32_t s, void* out_emb, void* out_encoded,
                             int32_t out_dtype, void* stream_v) {
  if (!e || !ids || !out_emb) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (n <= 0 || s <= 0) return fail(SMI_ERR_INVALID_ARG, "empty batch (n=%d, s=%d)", n, s);
  if (out_dtype != SMI_F32 && out_dtype != SMI_F16)
    return fail(SMI_ERR_INVALID_ARG, "bad out_dtype %d", out_dtype);
  const smi_text_encoder_config& c = e->cfg;
  if (s > c.max_seq_len)
    return fail(SMI_ERR_INVALID_ARG, "seq_len %d exceeds max_seq_len %d of the encoder", s,
                c.max_seq_len);
  hipStream_t stream = (hipStream_t)stream_v;
  if (!e->bad_ids) {
    HIP_TRY(hipHostMalloc((void**)&e->bad_ids, sizeof(int32_t), hipHostMallocMapped));
    *e->bad_ids = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&e->bad_ids_dev, e->bad_ids, 0));
  }
  // (the out-of-vocabulary flag is reported -- and cleared -- only by smi_text_encoder_status(), after a stream
  //  synchronisation: testing it here, unsynchronised, refused a later VALID batch at a timing-dependent point)
  if (e->flex) return flex_encoder_forward(e->flex, ids, seq_lens, n, s, out_emb, out_encoded, out_dtype, e->bad_ids_dev, stream);

  if (int rc = ensure_cu(e, n)) return rc;
  const int slot = e->cu_next;
  e->cu_next = (slot + 1) % kCuRing;
  HIP_TRY(hipEventSynchronize(e->cu_ev[slot]));  // previous use of this slot has been copied
  int32_t* cu = e->h_cu[slot];
  int64_t total = 0;
  int max_len = 0;
  cu[0] = 0;
  for (int i = 0; i < n; ++i) {
    const int len = seq_lens ? seq_lens[i] : s;
    if (len < 0 || len > s) return fail(SMI_ERR_INVALID_ARG, "seq_lens[%d]=%d outside [0,%d]", i, len, s);
    total += len;
    if (total > 0x7fffff00LL) return fail(SMI_ERR_UNSUPPORTED, "too many tokens in one batch");
    cu[i + 1] = (int32_t)total;
    max_len = std::max(max_len, len);
  }
  const int32_t* d_cu = e->d_cu[slot].as<int32_t>();
  HIP_TRY(hipMemcpyAsync((void*)d_cu, cu, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipEventRecord(e->cu_ev[slot], stream));

  const int d = c.model_dim, f = c.ffn_inner_dim;
  if (total == 0) {  // nothing but empty sentences: pooled vectors are zero
    HIP_TRY(hipMemsetAsync(out_emb, 0, (size_t)n * d * (out_dtype == SMI_F32 ? 4 : 2), stream));
    if (out_encoded)
      HIP_TRY(hipMemsetAsync(out_encoded, 0, (size_t)n * s * d * (out_dtype == SMI_F32 ? 4 : 2), stream));
    return SMI_OK;
  }
  const int64_t rows = (total + 255) / 256 * 256;  // 256-row GEMM tiles
  if (int rc = ensure_workspace(e, rows)) return rc;
  const int M = (int)rows;

  void* x = e->x.p;
  const int x16 = e->x16;
  const size_t xes = x16 ? 2 : 4;
  const int epi_resid = x16 ? EPI_RESID_F16 : EPI_RESID_F32;
  f16* h = e->h.as<f16>();
  f16* qkv = e->qkv.as<f16>();
  f16* ctx = e->ctx.as<f16>();
  f16* ffn = e->ffn.as<f16>();

  // The fp16 residual stream itself is tile-major when everything that touches it has that path (d = 1024, no
  // encoded_seqs output): the residual epilogues of the attention-output and FFN-output GEMMs then read-modify-write
  // it straight from the accumulators instead of staging the tile through LDS in 8 barrier-separated passes.
  const bool x_tm_enabled = tune(TUNE_ENC_X_TM, 1) != 0;  // ENC_X_TM=0: row-major residual stream (A/B measurements)
  // Small batches: the FFN output projection (K = ffn_inner_dim) has (M/256)*(d/256) output tiles -- a handful of
  // CUs would each walk 256 K slices (C1, 1312 tokens: 130 us per layer, 3/4 of the forward).  It is then run as a
  // split-K GEMM into fp32 slabs (ks * tiles units, one round on the chip) that a small kernel folds into x.
  int ffn2_ks = 1;
  {
    const int64_t tiles = (rows / 256) * (d / 256);
    while (ffn2_ks < 8 && tiles * ffn2_ks * 2 <= e->num_cus && f % (ffn2_ks * 2 * 512) == 0) ffn2_ks *= 2;
    if (d % 256 || ffn2_ks < 2) ffn2_ks = 1;
  }
  // In that mode the whole layer takes the decoder's shape (6 launches instead of 8; SMI_ENC_SB=0 keeps the round-2
  // schedule for A/B runs): the attention output projection is split-K too (every 128x128 unit on a CU of its own, the
  // lone-tile ring engine), and the slabs of both projections are folded into the residual stream by the fused
  // sum + LayerNorm kernel that produces the next GEMM's input -- no separate fold, no separate LayerNorm.
  const bool sb_env = tune(TUNE_ENC_SB, 1) != 0;
  const bool sb = sb_env && ffn2_ks > 1 && c.num_layers > 0;
  const int out_ks = sb ? gemm_splitk_parts((int)rows, d, d, 8) : 1;
  const int max_ks = std::max(ffn2_ks, out_ks);
  if (ffn2_ks > 1 && e->parts.bytes < (size_t)max_ks * rows * d * 4) {
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(e->parts.alloc((size_t)max_ks * rows * d * 4));
  }
  const int x_tm = x_tm_enabled && e->tile_major && x16 && d == 1024 && !out_encoded && ffn2_ks == 1;
  if (rows > total) {
    if (x_tm)  // the rows of the last 256-row panel are interleaved: clear the whole panel, the embedding refills it
      HIP_TRY(hipMemsetAsync((char*)x + (size_t)(rows - 256) * d * xes, 0, (size_t)256 * d * xes, stream));
    else
      HIP_TRY(hipMemsetAsync((char*)x + (size_t)total * d * xes, 0, (size_t)(rows - total) * d * xes, stream));
  }
  { ProfScope ps(e, SMI_PROF_EMBED, stream);
  HIP_TRY(launch_embed_pack(ids, d_cu, e->embed.as<f16>(), e->pos.as<float>(), c.embed_scale,
                            c.pos_offset, x, n, s, max_len, d, c.vocab_size, stream, x16, e->bad_ids_dev, x_tm)); }
  // x (residual stream, fp32 or fp16 with SMI_ENC_FP16_RESIDUAL) stays row-major; h, qkv, ctx, ffn and the weights are tile-major
  const int tm = e->tile_major;
  const int in_tm = tm ? GEMM_IN_TM : 0, io_tm = tm ? GEMM_IN_TM | GEMM_OUT_TM : 0;
  const int x_out_tm = x_tm ? GEMM_OUT_TM : 0;
  // LayerNorm fold (kernels.hpp: GemmLnFold): with the tile-major fp16 stream the two LayerNorms of a layer are not
  // launched; the residual GEMMs leave the row sums of the stream, the QKV / FFN-inner GEMMs multiply the stream itself
  // by the pre-scaled weights and apply mean / rstd in their epilogues.
  const bool lnfold = x_tm && e->lnfold;
  const int nparts = d / 256;
  GemmLnFold fold_prod{nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0}, fold_cons = fold_prod;
  if (lnfold) {
    if (e->rowpart.bytes < (size_t)nparts * M * 8) {
      HIP_TRY(hipStreamSynchronize(stream));
      HIP_TRY(e->rowpart.alloc((size_t)nparts * M * 8));
    }
    fold_prod.part_out = e->rowpart.as<float2>();
    fold_cons.part_in = e->rowpart.as<float2>();
    fold_cons.nparts = nparts;
    fold_cons.inv_k = 1.0f / d;
    fold_cons.eps = c.ln_eps;
    fold_cons.centered = e->lnfold_centered;
    ProfScope ps(e, SMI_PROF_LAYERNORM, stream);  // the first LayerNorm's statistics have no producing GEMM
    HIP_TRY(launch_row_stats_tm((const f16*)x, e->rowpart.as<float2>(), M, d, nparts, stream));
  }
  const bool pf_on = true;  // weight prefetch by the row kernels' surplus workgroups (SMI_PREFETCH=0 disables it: A/B runs)
  for (int l = 0; l < c.num_layers && sb; ++l) {  // small batches: the decoder-shaped layer (see above)
    Layer& L = e->layers[l];
    void* parts = e->parts.p;
    const size_t ps = (size_t)M * d;
    // fp16 slabs for an fp16 residual stream (the fp16 model: the reference rounds every sublayer output to fp16; here each
    // split-K partial is rounded once, the sum is formed in fp32 and meets the stream in one rounding as before):
    // half the slab traffic between a projection and the kernel that folds it.  SMI_ENC_SLAB_F16=0: fp32 slabs (A/B runs)
    const bool slab_env = tune(TUNE_ENC_SLAB_F16, 1) != 0;
    const int sf16 = slab_env && x16;
    { ProfScope ps_(e, SMI_PROF_LAYERNORM, stream);  // x += FFN-output slabs of the previous layer; h = LN1(x)
    HIP_TRY(launch_sum_layernorm(x, l ? parts : nullptr, ffn2_ks, ps, nullptr, 1, L.ln1_w.as<float>(), L.ln1_b.as<float>(),
                                 c.ln_eps, h, M, d, stream, tm, x16, pf_on ? L.w_1.p : nullptr, (size_t)f * d * 2, sf16)); }
    { ProfScope ps_(e, SMI_PROF_GEMM_QKV, stream);
    HIP_TRY(launch_gemm_tn(EPI_BIAS_F16 | io_tm, h, L.w_qkv.as<f16>(), L.b_qkv.as<float>(), qkv, M, 3 * d, d, 3 * d, stream)); }
    { ProfScope ps_(e, SMI_PROF_ATTENTION, stream);
    HIP_TRY(launch_attention(qkv, d_cu, ctx, n, max_len, d, c.num_heads, stream, tm ? 3 : 0)); }
    { ProfScope ps_(e, SMI_PROF_GEMM_OUT, stream);
    HIP_TRY(launch_gemm_tn_splitk(ctx, L.w_o.as<f16>(), L.b_o.as<float>(), parts, M, d, d, out_ks, stream, tm, sf16)); }
    { ProfScope ps_(e, SMI_PROF_LAYERNORM, stream);  // x += attention-output slabs; h = LN2(x)
    HIP_TRY(launch_sum_layernorm(x, parts, out_ks, ps, nullptr, 1, L.ln2_w.as<float>(), L.ln2_b.as<float>(), c.ln_eps, h, M, d,
                                 stream, tm, x16, pf_on ? L.w_2.p : nullptr, (size_t)f * d * 2, sf16)); }
    { ProfScope ps_(e, SMI_PROF_GEMM_FFN1, stream);
    HIP_TRY(launch_gemm_tn(EPI_RELU_F16 | io_tm, h, L.w_1.as<f16>(), L.b_1.as<float>(), ffn, M, f, d, f, stream)); }
    { ProfScope ps_(e, SMI_PROF_GEMM_FFN2, stream);
    HIP_TRY(launch_gemm_tn_splitk(ffn, L.w_2.as<f16>(), L.b_2.as<float>(), parts, M, d, f, ffn2_ks, stream, tm, sf16));
    if (l + 1 == c.num_layers)  // the last layer's slabs meet the stream before the final LayerNorm + pooling
      HIP_TRY(launch_fold_residual(x, x16, parts, ffn2_ks, ps, ps, stream, sf16)); }
  }
  for (int l = 0; l < c.num_layers && !sb; ++l) {
    Layer& L = e->layers[l];
    if (lnfold) {
      ProfScope ps(e, SMI_PROF_GEMM_QKV, stream);
      fold_cons.c1 = L.c1_qkv.as<float>();
      HIP_TRY(launch_gemm_tn(EPI_BIAS_F16 | io_tm, (const f16*)x, L.w_qkv_f.as<f16>(), L.c2_qkv.as<float>(), qkv, M, 3 * d,
                             d, 3 * d, stream, nullptr, &fold_cons));
    } else {
    { ProfScope ps(e, SMI_PROF_LAYERNORM, stream);
    HIP_TRY(launch_layernorm(x, L.ln1_w.as<float>(), L.ln1_b.as<float>(), c.ln_eps, h, M, d, stream, tm, x16, x_tm)); }
    { ProfScope ps(e, SMI_PROF_GEMM_QKV, stream);
    HIP_TRY(launch_gemm_tn(EPI_BIAS_F16 | io_tm, h, L.w_qkv.as<f16>(), L.b_qkv.as<float>(), qkv, M, 3 * d,
                           d, 3 * d, stream)); }
    }
    { ProfScope ps(e, SMI_PROF_ATTENTION, stream);
    HIP_TRY(launch_attention(qkv, d_cu, ctx, n, max_len, d, c.num_heads, stream, tm ? 3 : 0)); }
    { ProfScope ps(e, SMI_PROF_GEMM_OUT, stream);
    HIP_TRY(launch_gemm_tn(epi_resid | in_tm | x_out_tm, ctx, L.w_o.as<f16>(), L.b_o.as<float>(), x, M, d, d, d,
                           stream, nullptr, lnfold ? &fold_prod : nullptr)); }
    if (lnfold) {
      ProfScope ps(e, SMI_PROF_GEMM_FFN1, stream);
      fold_cons.c1 = L.c1_1.as<float>();
      HIP_TRY(launch_gemm_tn(EPI_RELU_F16 | io_tm, (const f16*)x, L.w_1_f.as<f16>(), L.c2_1.as<float>(), ffn, M, f, d, f,
                             stream, nullptr, &fold_cons));
    } else {
    { ProfScope ps(e, SMI_PROF_LAYERNORM, stream);
    HIP_TRY(launch_layernorm(x, L.ln2_w.as<float>(), L.ln2_b.as<float>(), c.ln_eps, h, M, d, stream, tm, x16, x_tm)); }
    { ProfScope ps(e, SMI_PROF_GEMM_FFN1, stream);
    HIP_TRY(launch_gemm_tn(EPI_RELU_F16 | io_tm, h, L.w_1.as<f16>(), L.b_1.as<float>(), ffn, M, f, d, f,
                           stream)); }
    }
    { ProfScope ps(e, SMI_PROF_GEMM_FFN2, stream);
    if (ffn2_ks > 1) {
      HIP_TRY(launch_gemm_tn_splitk(ffn, L.w_2.as<f16>(), L.b_2.as<float>(), e->parts.as<float>(), M, d, f, ffn2_ks,
                                    stream, tm));
      HIP_TRY(launch_fold_residual(x, x16, e->parts.as<float>(), ffn2_ks, (size_t)M * d, (size_t)M * d, stream));
    } else {
      HIP_TRY(launch_gemm_tn(epi_resid | in_tm | x_out_tm, ffn, L.w_2.as<f16>(), L.b_2.as<float>(), x, M, d, f, d,
                             stream, nullptr, lnfold ? &fold_prod : nullptr));
    } }
  }
  { ProfScope ps(e, SMI_PROF_LN_POOL, stream);
  HIP_TRY(launch_ln_pool(x, e->lnf_w.as<float>(), e->lnf_b.as<float>(), c.ln_eps, d_cu, out_emb,
                         out_dtype == SMI_F32, out_encoded, n, s, d, c.pooling, stream, x16, x_tm)); }
  return SMI_OK;
}

int smi_text_encoder_status(smi_text_encoder* e, void* stream) {
  if (!e) return fail(SMI_ERR_INVALID_ARG, "null argument");
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  if (e->bad_ids && *(volatile int32_t*)e->bad_ids) {
    *e->bad_ids = 0;
    return fail(SMI_ERR_INVALID_ARG, "token ids outside [0, %lld) reached the encoder (vocabulary / tokenizer mismatch)",
                (long long)e->cfg.vocab_size);
  }
  return SMI_OK;
}

int smi_text_encoder_set_profiling(smi_text_encoder* e, int32_t enable) {
  if (!e) return fail(SMI_ERR_INVALID_ARG, "null argument");
  e->profiling = enable != 0;
  return SMI_OK;
}

int smi_text_encoder_read_profile(smi_text_encoder* e, double* ms, int64_t* launches) {
  if (!e || !ms || !launches) return fail(SMI_ERR_INVALID_ARG, "null argument");
  for (auto& r : e->prof) {
    HIP_TRY(hipEventSynchronize(r.b));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.slot] += t;
    launches[r.slot] += 1;
  }
  e->prof.clear();
  e->ev_used = 0;
  return SMI_OK;
}

// ------------------------------------------------------------------- xsim
int64_t smi_xsim_padded_rows(int64_t rows) { return (rows + 255) / 256 * 256; }

int smi_xsim_normalize(const void* src, int32_t src_dtype, int64_t rows, int32_t d, void* dst,
                       void* stream) {
  if (!src || !dst || rows <= 0 || d <= 0) return fail(SMI_ERR_INVALID_ARG, "bad argument");
  if (src_dtype != SMI_F32 && src_dtype != SMI_F16) return fail(SMI_ERR_INVALID_ARG, "bad dtype");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(launch_l2_normalize(src, src_dtype == SMI_F32, (f16*)dst, rows, d, (hipStream_t)stream));
  return SMI_OK;
}

int64_t smi_xsim_workspace_bytes(int64_t nx, int64_t ny, int32_t k, int32_t d) {
  if (nx <= 0 || ny <= 0 || k < 1 || k > 8 || d <= 0) return 0;
  return (int64_t)xsim_workspace_bytes(smi_xsim_padded_rows(nx), smi_xsim_padded_rows(ny), k, d);
}

int smi_xsim_topk(const void* xn, int64_t nx, const void* yn, int64_t ny, int32_t d, int32_t k,
                  int64_t y_off, int32_t* idx, float* score, void* ws, int64_t ws_bytes, void* stream) {
  if (!xn || !yn || !idx || !score || !ws) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (k < 1 || k > 8) return fail(SMI_ERR_UNSUPPORTED, "k=%d outside [1,8]", k);
  if (d <= 0 || d % 64) return fail(SMI_ERR_UNSUPPORTED, "d=%d must be a multiple of 64", d);
  if (nx <= 0 || ny <= 0) return fail(SMI_ERR_INVALID_ARG, "empty input");
  if (ws_bytes < smi_xsim_workspace_bytes(nx, ny, k, d))
    return fail(SMI_ERR_INVALID_ARG, "workspace of %lld bytes, smi_xsim_workspace_bytes(nx, ny, k, d) = %lld",
                (long long)ws_bytes, (long long)smi_xsim_workspace_bytes(nx, ny, k, d));
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(launch_xsim_topk((const f16*)xn, nx, smi_xsim_padded_rows(nx), (const f16*)yn, ny,
                           smi_xsim_padded_rows(ny), d, k, y_off, idx, score, ws, (hipStream_t)stream));
  return SMI_OK;
}

int smi_xsim_merge_topk(const float* part_scores, const int32_t* part_idx, int32_t parts, int64_t n, int32_t k,
                        float* out_scores, int32_t* out_idx, void* stream) {
  if (!part_scores || !out_scores) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (parts < 1 || parts > 64) return fail(SMI_ERR_UNSUPPORTED, "parts=%d outside [1,64]", parts);
  if (k < 1 || k > 8) return fail(SMI_ERR_UNSUPPORTED, "k=%d outside [1,8]", k);
  if (n <= 0) return fail(SMI_ERR_INVALID_ARG, "empty input");
  if (out_idx && !part_idx) return fail(SMI_ERR_INVALID_ARG, "out_idx needs part_idx");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(launch_topk_merge(part_scores, part_idx, parts, n, k, out_scores, out_idx, (hipStream_t)stream));
  return SMI_OK;
}

int smi_xsim_margin_select(const float* fwd_scores, const int32_t* fwd_idx, int64_t nx, int32_t k,
                           const float* bwd_scores, int64_t ny, int32_t margin, int64_t x_index_offset,
                           int32_t* pred_idx, float* pred_margin, int32_t* err_count, void* stream) {
  if (!fwd_scores || !fwd_idx || !pred_idx) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (k < 1 || k > 8) return fail(SMI_ERR_UNSUPPORTED, "k=%d outside [1,8]", k);
  if (margin < SMI_MARGIN_RATIO || margin > SMI_MARGIN_COSINE) return fail(SMI_ERR_INVALID_ARG, "unknown margin %d", margin);
  if (nx <= 0) return fail(SMI_ERR_INVALID_ARG, "empty input");
  if (margin != SMI_MARGIN_COSINE && (!bwd_scores || ny <= 0))
    return fail(SMI_ERR_INVALID_ARG, "margin scoring needs the backward (y-side) neighbour scores");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(launch_margin_select(fwd_scores, fwd_idx, nx, k, bwd_scores, ny, margin, x_index_offset, pred_idx,
                               pred_margin, err_count, (hipStream_t)stream));
  return SMI_OK;
}

// -------------------------------------------------------- building blocks
int smi_pack_tile_major(const void* src, void* dst, int32_t rows, int32_t k, int32_t inverse,
                        void* stream) {
  if (!src || !dst || src == dst) return fail(SMI_ERR_INVALID_ARG, "bad argument");
  if (rows <= 0 || rows % 256 || k <= 0 || k % 32)
    return fail(SMI_ERR_UNSUPPORTED, "tile-major needs rows %% 256 == 0 and k %% 32 == 0 (rows=%d k=%d)", rows, k);
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(launch_pack_tile_major((const f16*)src, (f16*)dst, rows, k, inverse, (hipStream_t)stream));
  return SMI_OK;
}

int smi_gemm_tn(int32_t epi, const void* x, const void* w, const float* bias, void* out, int32_t m,
                int32_t n, int32_t k, int32_t ldo, void* stream) {
  if (!x || !w || !out) return fail(SMI_ERR_INVALID_ARG, "null argument");
  const int e = epi & 0xff, sel = (epi >> 8) & 0xf;
  const bool in_tm = epi & GEMM_IN_TM, out_tm = epi & GEMM_OUT_TM;
  if (epi < 0 || (epi & ~(0xfff | GEMM_IN_TM | GEMM_OUT_TM)) || e > 9 || sel > 2 || m <= 0 || m % 128 ||
      n <= 0 || n % 128 || k <= 0 || k % 64 || ldo < (e == 6 ? n / 2 : n) || (sel == 2 && (m % 256 || n % 256)))
    return fail(SMI_ERR_UNSUPPORTED, "gemm shape m=%d n=%d k=%d epi=%d ldo=%d", m, n, k, epi, ldo);
  if (in_tm && (m % 256 || n % 256 || (out_tm ? (e != 0 && e != 1 && e != 5 && e != 6 && e != 8 && e != 9) : (e != 0 && e != 2 && e != 3 && e != 4 && e != 6 && e != 8 && e != 9))))
    return fail(SMI_ERR_UNSUPPORTED, "tile-major gemm: m=%d n=%d epi=%d", m, n, epi);
  if (out_tm && (!in_tm || ldo != (e == 6 ? n / 2 : n)))
    return fail(SMI_ERR_UNSUPPORTED, "tile-major output needs tile-major inputs and ldo == n (n / 2 for the GLU epilogue)");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  {
    const hipError_t he = launch_gemm_tn(epi, (const f16*)x, (const f16*)w, bias, out, m, n, k, ldo, (hipStream_t)stream);
    // combinations only one engine implements (the GLU epilogue with a tile-major output: the 4-wave engine, from its tile
    // threshold up, with a bias) are refused, not mis-computed
    if (he == hipErrorInvalidValue)
      return fail(SMI_ERR_UNSUPPORTED, "gemm m=%d n=%d k=%d epi=%d: no engine takes this combination", m, n, k, epi);
    HIP_TRY(he);
  }
  return SMI_OK;
}

int smi_gemm_tn_splitk(const void* x, const void* w, const float* bias, void* parts, int32_t m, int32_t n, int32_t k,
                       int32_t ksplit, int32_t in_tm, int32_t slab_dtype, void* stream) {
  if (!x || !w || !parts) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (slab_dtype != SMI_F16 && slab_dtype != SMI_F32) return fail(SMI_ERR_INVALID_ARG, "slab dtype %d: SMI_F16 or SMI_F32", slab_dtype);
  if (m <= 0 || m % 128 || n <= 0 || n % 128 || k <= 0 || k % 64 || ksplit < 1 || ksplit > 16 ||
      (in_tm && (m % 256 || n % 256)))
    return fail(SMI_ERR_UNSUPPORTED, "split-K gemm shape m=%d n=%d k=%d ksplit=%d", m, n, k, ksplit);
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  const hipError_t e = launch_gemm_tn_splitk((const f16*)x, (const f16*)w, bias, parts, m, n, k, ksplit, (hipStream_t)stream,
                                             in_tm ? 1 : 0, slab_dtype == SMI_F16);
  if (e == hipErrorInvalidValue) return fail(SMI_ERR_UNSUPPORTED, "split-K gemm: k=%d does not split into %d parts", k, ksplit);
  HIP_TRY(e);
  return SMI_OK;
}

int smi_gemm_tn_tile_stats(const void* x, const void* w, void* out, int32_t m, int32_t n, int32_t k, float scale, int32_t valid_n,
                           float* tile_max, float* tile_sum, void* stream) {
  if (!x || !w || !out || !tile_max || !tile_sum) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (m <= 0 || m % 256 || n <= 0 || n % 256 || k <= 0 || k % 64 || !(scale > 0.f) || valid_n <= 0 || valid_n > n)
    return fail(SMI_ERR_UNSUPPORTED, "tile-statistics gemm m=%d n=%d k=%d scale=%g valid_n=%d", m, n, k, (double)scale, valid_n);
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  const GemmTileStats st{tile_max, tile_sum, scale, valid_n};
  HIP_TRY(launch_gemm_tn(EPI_BIAS_F16 | GEMM_IN_TM | GEMM_OUT_TM, (const f16*)x, (const f16*)w, nullptr, out, m, n, k, n,
                         (hipStream_t)stream, &st));
  return SMI_OK;
}

int smi_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!src || !dst))) return fail(SMI_ERR_INVALID_ARG, "bad argument");
  if (src_dtype < SMI_F32 || src_dtype > SMI_BF16 || dst_dtype < SMI_F32 || dst_dtype > SMI_BF16)
    return fail(SMI_ERR_INVALID_ARG, "bad dtype %d -> %d", src_dtype, dst_dtype);
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(launch_cast(src, src_dtype, dst, dst_dtype, (size_t)n, (hipStream_t)stream));
  return SMI_OK;
}

int smi_layernorm(const float* x, const float* w, const float* b, float eps, void* out, int32_t rows,
                  int32_t d, int32_t tile_major, void* stream) {
  if (!x || !w || !b || !out || rows <= 0) return fail(SMI_ERR_INVALID_ARG, "bad argument");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  hipError_t e = launch_layernorm(x, w, b, eps, (f16*)out, rows, d, (hipStream_t)stream, tile_major != 0);
  if (e == hipErrorInvalidValue) return fail(SMI_ERR_UNSUPPORTED, "layernorm d=%d unsupported", d);
  HIP_TRY(e);
  return SMI_OK;
}

int smi_attention(const void* qkv, const int32_t* cu, void* ctx, int32_t n, int32_t max_len,
                  int32_t d, int32_t heads, int32_t tile_major, void* stream) {
  if (!qkv || !cu || !ctx) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (heads <= 0 || d != heads * 64) return fail(SMI_ERR_UNSUPPORTED, "head_dim must be 64");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(launch_attention((const f16*)qkv, cu, (f16*)ctx, n, max_len, d, heads, (hipStream_t)stream,
                           tile_major & 3));
  return SMI_OK;
}

int smi_relpos_attention(const void* qkv, const int32_t* cu, const void* rp, int32_t rp_zero, int32_t rp_rows, const float* u_bias,
                         const float* v_bias, void* ctx, int32_t n, int32_t max_len, int32_t d, int32_t heads, int32_t tile_major,
                         void* stream) {
  if (!qkv || !cu || !rp || !u_bias || !v_bias || !ctx) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (heads <= 0 || d != heads * 64) return fail(SMI_ERR_UNSUPPORTED, "head_dim must be 64");
  if (n <= 0 || max_len <= 0 || rp_rows <= 0 || rp_zero < 0 || rp_zero >= rp_rows || tile_major < 0 || tile_major > 3)
    return fail(SMI_ERR_INVALID_ARG, "n=%d max_len=%d rp_zero=%d rp_rows=%d tile_major=%d", n, max_len, rp_zero, rp_rows, tile_major);
  if ((tile_major & 2) && !relpos_attention_reads_tile_major())
    return fail(SMI_ERR_UNSUPPORTED, "tile-major q | k | v needs the LDS-ring kernel (SPEECH_RP_LDS, SPEECH_QKV_TM)");
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(launch_relpos_attention((const f16*)qkv, cu, (const f16*)rp, rp_zero, rp_rows, u_bias, v_bias, (f16*)ctx, n, max_len, d,
                                  heads, (hipStream_t)stream, tile_major & 1, (tile_major >> 1) & 1));
  return SMI_OK;
}


}  // extern "C"
