// C-ABI implementation of the text decoder / beam search (see include/sonar_mi355.h).
// Host runtime: weight packing, workspace, and the per-step launch schedule
// (reference op order: sonar/models/sonar_text/factory.py:261-307, pre-LN decoder layers,
// final LayerNorm, tied projection; generation control: fairseq2 BeamSearchSeq2SeqGenerator).
#include <cmath>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "api_common.hpp"

using namespace smi;
using namespace smi_host;

namespace {

struct DecLayer {
  DevBuf ln1_w, ln1_b, w_qkv, b_qkv, w_o, b_o;
  DevBuf wc_v, bc_v, wc_o, bc_o;  // encoder_decoder_attn value / output projections
  DevBuf ln3_w, ln3_b, w_1, b_1, w_2, b_2;
};

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

constexpr int kMaxParts = 16;  // split-K slabs the `parts` buffer holds

}  // namespace

// Per-call workspace of ONE decode chain (grow-only).  smi_text_decoder_generate may split a batch into several
// chains of whole sentences: they share nothing but the weights, so each owns its activations, KV cache, beam state
// and stream, and their launch sequences overlap on the GPU.
// A/B switches of the decode step, read from the environment when a generate / sample / logits call starts:
// SMI_DEC_KS_OUT, SMI_DEC_KS_FFN (split-K parts of the attention-output / FFN-output projections; 0 = automatic,
// KS_OUT 1 = no split), SMI_DEC_FFN1_ENGINE (-1 automatic, 0 the GEMM's own choice, 1 = 128x128, 2 = 256x256),
// SMI_DEC_LOGITS_GRID (persistent workgroups of a chained call's logits GEMM; 0 = all CUs)
struct DecTuning {
  int ks_out = 0, ks_ffn = 0, ffn1_engine = -1, logits_grid = 0;
  void read() {
    ks_out = tune(TUNE_DEC_KS_OUT, 0);
    ks_ffn = tune(TUNE_DEC_KS_FFN, 0);
    ffn1_engine = tune(TUNE_DEC_FFN1_ENGINE, -1);
    logits_grid = tune(TUNE_DEC_LOGITS_GRID, 0);
  }
};

struct DecWork {
  DecTuning tuning;
  DevBuf x, h, ctx, ffn, logits, kv, cc, cvtmp, emb16, parts;
  DevBuf tok, cum, parent, new_tok, new_cum, nactive, done, ndone, fin_count, fin_len, fin_score, fin_tok;
  DevBuf anc[2], hist[2];
  DevBuf pmax, psum, pval, pidx;
  DevBuf tile_max, tile_sum;  // logits-GEMM tile statistics [rows_pad][vocab_pad / 256]
  int kv_positions = 0;       // positions per layer in the current kv allocation
  bool chained = false;       // this call runs next to other chains: per-launch tile choices differ (decoder_step)
  hipStream_t stream = nullptr;  // chains of a split call run on streams of their own
  hipEvent_t done_ev = nullptr;
};

constexpr int kMaxChains = 4;

struct smi_text_decoder {
  smi_text_decoder_config cfg;
  int64_t vocab_pad = 0;
  DevBuf embed, pos, lnf_w, lnf_b;
  DevBuf embed_tm;  // tile-major copy of the (padded) table for the tied output projection
  std::vector<DecLayer> layers;
  DecWork ws[kMaxChains];  // ws[0]: the single-chain calls (logits, sample, unsplit generate)
  hipEvent_t fork_ev = nullptr;
  DevBuf margins;       // [n][2] decision margins of the last generate() call
  int margins_n = 0;
  int chains = 0;       // smi_text_decoder_set_chains: 0 = SMI_DEC_CHAINS / the default
  int beam_logits_f16 = 0;  // smi_text_decoder_set_beam_logits_dtype: the beam search's logits are stored in fp16
  int beam_slab_f16 = 0;    // smi_text_decoder_set_slab_dtype: the beam search's split-K partial sums are stored in fp16
  int64_t weight_bytes = 0;
  int ffn_tile_major = 0;  // FFN weights stored tile-major (d, f multiples of 256)
  // Generic-dimension mode (flex.hip): head_dim != 64 or dimensions the MFMA engines do not tile for (the reference's
  // `toy` arch, config.py:232-255).  fp32 weights, activations and KV cache; the beam search / sampling kernels are
  // shared with the fast mode (they only see logits rows and tile statistics).
  bool flex = false;
  size_t act_bytes() const { return flex ? 4 : 2; }  // element size of h / ctx / ffn / the KV cache
  ~smi_text_decoder() {
    for (auto& w : ws) {
      if (w.stream) (void)hipStreamDestroy(w.stream);
      if (w.done_ev) (void)hipEventDestroy(w.done_ev);
    }
    if (fork_ev) (void)hipEventDestroy(fork_ev);
  }
};

namespace {

// SMI_OK and *flex = false: the MFMA engines cover the shape; *flex = true: the generic-dimension kernels do
int check_dec_cfg(const smi_text_decoder_config& c, bool* flex) {
  if (c.model_dim <= 0 || c.num_heads <= 0 || c.model_dim % c.num_heads)
    return fail(SMI_ERR_INVALID_ARG, "model_dim %d must be a positive multiple of num_heads %d", c.model_dim, c.num_heads);
  if (c.ffn_inner_dim <= 0 || c.input_dim <= 0)
    return fail(SMI_ERR_INVALID_ARG, "bad ffn_inner_dim %d / input_dim %d", c.ffn_inner_dim, c.input_dim);
  if (c.num_layers < 0 || c.vocab_size <= 16 || c.max_seq_len <= 1 || c.pos_offset < 0)
    return fail(SMI_ERR_INVALID_ARG, "bad num_layers/vocab_size/max_seq_len/pos_offset");
  const bool fast = c.model_dim == c.num_heads * 64 && c.model_dim % 256 == 0 &&
                    (c.model_dim / 256 <= 4 || c.model_dim == 2048) && c.ffn_inner_dim % 128 == 0 && c.input_dim % 64 == 0;
  if (!fast && c.model_dim / c.num_heads > 256)
    return fail(SMI_ERR_UNSUPPORTED, "head_dim %d > 256 is not covered", c.model_dim / c.num_heads);
  *flex = !fast;
  return SMI_OK;
}

// per-sentence cross-attention constants cc[l][s] = W_o (W_v e_s + b_v) + b_o  (fp32 [L][n_pad][d])
int compute_cross_constants(smi_text_decoder* D, DecWork& S, const void* emb, int emb_dtype, int n, int n_pad,
                            hipStream_t stream) {
  const int d = D->cfg.model_dim, ci = D->cfg.input_dim;  // the conditioning vector may be wider / narrower than the model
  HIP_TRY(S.cc.reserve((size_t)D->cfg.num_layers * n_pad * d * 4));
  if (D->flex) {
    HIP_TRY(S.emb16.reserve((size_t)n * ci * 4));  // fp32 copy of the sentence vectors
    HIP_TRY(S.cvtmp.reserve((size_t)n * d * 4));
    if (emb_dtype == SMI_F32)
      HIP_TRY(hipMemcpyAsync(S.emb16.p, emb, (size_t)n * ci * 4, hipMemcpyDeviceToDevice, stream));
    else
      HIP_TRY(launch_f16_to_f32((const f16*)emb, S.emb16.as<float>(), (size_t)n * ci, stream));
    for (int l = 0; l < D->cfg.num_layers; ++l) {
      DecLayer& L = D->layers[l];
      HIP_TRY(launch_flex_linear(S.emb16.as<float>(), ci, L.wc_v.as<float>(), L.bc_v.as<float>(), S.cvtmp.as<float>(), d,
                                 n, d, ci, 0, nullptr, 0, stream));
      HIP_TRY(launch_flex_linear(S.cvtmp.as<float>(), d, L.wc_o.as<float>(), L.bc_o.as<float>(),
                                 S.cc.as<float>() + (size_t)l * n_pad * d, d, n, d, d, 0, nullptr, 0, stream));
    }
    return SMI_OK;
  }
  HIP_TRY(S.emb16.reserve((size_t)n_pad * ci * 2));
  HIP_TRY(S.cvtmp.reserve((size_t)n_pad * d * 2));
  HIP_TRY(hipMemsetAsync(S.emb16.p, 0, (size_t)n_pad * ci * 2, stream));
  if (emb_dtype == SMI_F32)
    HIP_TRY(launch_f32_to_f16((const float*)emb, S.emb16.as<f16>(), (size_t)n * ci, stream));
  else
    HIP_TRY(hipMemcpyAsync(S.emb16.p, emb, (size_t)n * ci * 2, hipMemcpyDeviceToDevice, stream));
  HIP_TRY(hipMemsetAsync(S.cc.p, 0, (size_t)D->cfg.num_layers * n_pad * d * 4, stream));
  for (int l = 0; l < D->cfg.num_layers; ++l) {
    DecLayer& L = D->layers[l];
    HIP_TRY(launch_gemm_tn(EPI_BIAS_F16, S.emb16.as<f16>(), L.wc_v.as<f16>(), L.bc_v.as<float>(),
                           S.cvtmp.p, n_pad, d, ci, d, stream));
    HIP_TRY(launch_gemm_tn(EPI_RESID_F32, S.cvtmp.as<f16>(), L.wc_o.as<f16>(), L.bc_o.as<float>(),
                           S.cc.as<float>() + (size_t)l * n_pad * d, n_pad, d, d, d, stream));
  }
  return SMI_OK;
}

// The same step on the generic-dimension kernels (flex.hip): fp32 throughout, one launch per reference module.
int flex_decoder_step(smi_text_decoder* D, DecWork& S, int rows, int rows_pad, int group, int n_pad, int pos,
                      const int32_t* anc, int anc_stride, hipStream_t stream, float stats_scale) {
  const smi_text_decoder_config& c = D->cfg;
  const int d = c.model_dim, f = c.ffn_inner_dim;
  float* x = S.x.as<float>();
  float* h = S.h.as<float>();
  float* ctx = S.ctx.as<float>();
  float* ffn = S.ffn.as<float>();
  const size_t slab = (size_t)rows_pad * 3 * d;
  const int P = S.kv_positions;
  HIP_TRY(launch_flex_embed(nullptr, S.tok.as<int32_t>(), D->embed.as<float>(), D->pos.as<float>(), c.embed_scale, x, rows, d,
                            1, c.pos_offset, pos, c.vocab_size, nullptr, stream));
  for (int l = 0; l < c.num_layers; ++l) {
    DecLayer& L = D->layers[l];
    float* kvl = S.kv.as<float>() + (size_t)l * P * slab;
    HIP_TRY(launch_flex_layernorm(x, L.ln1_w.as<float>(), L.ln1_b.as<float>(), c.ln_eps, h, rows, d, stream));
    HIP_TRY(launch_flex_linear(h, d, L.w_qkv.as<float>(), L.b_qkv.as<float>(), kvl + (size_t)pos * slab, 3 * d, rows, 3 * d,
                               d, 0, nullptr, 0, stream));
    HIP_TRY(launch_flex_dec_attention(kvl, anc, anc_stride, ctx, rows, rows_pad, d, c.num_heads, pos, stream));
    HIP_TRY(launch_flex_linear(ctx, d, L.w_o.as<float>(), L.b_o.as<float>(), x, d, rows, d, d, 0, x, d, stream));
    HIP_TRY(launch_flex_add_rows(x, S.cc.as<float>() + (size_t)l * n_pad * d, rows, d, group, stream));
    HIP_TRY(launch_flex_layernorm(x, L.ln3_w.as<float>(), L.ln3_b.as<float>(), c.ln_eps, h, rows, d, stream));
    HIP_TRY(launch_flex_linear(h, d, L.w_1.as<float>(), L.b_1.as<float>(), ffn, f, rows, f, d, 1, nullptr, 0, stream));
    HIP_TRY(launch_flex_linear(ffn, f, L.w_2.as<float>(), L.b_2.as<float>(), x, d, rows, d, f, 0, x, d, stream));
  }
  HIP_TRY(launch_flex_layernorm(x, D->lnf_w.as<float>(), D->lnf_b.as<float>(), c.ln_eps, h, rows, d, stream));
  HIP_TRY(launch_flex_linear(h, d, D->embed.as<float>(), nullptr, S.logits.as<float>(), (int)D->vocab_pad, rows,
                             (int)c.vocab_size, d, 0, nullptr, 0, stream));
  if (stats_scale > 0.f)
    HIP_TRY(launch_flex_tile_stats(S.logits.as<float>(), (int)D->vocab_pad, rows, (int)c.vocab_size, stats_scale,
                                   S.tile_max.as<float>(), S.tile_sum.as<float>(), rows_pad, stream));
  return SMI_OK;
}

// one decoder step at position `pos` for `rows` rows (rows_pad GEMM rows); logits -> S.logits
// logits_f16 (beam search of an fp16 model, round 4): the logits GEMM stores fp16 in the tile-major layout (the reference's
// fp16 model produces fp16 logits too: fairseq2 up-casts them inside log_softmax) and takes the tile statistics of the
// rounded values; needs the tile-major table copy.  Halves the 1.3 GB the 256 x 5-row step wrote per position.
int decoder_step(smi_text_decoder* D, DecWork& S, int rows, int rows_pad, int group, int n_pad, int pos,
                 const int32_t* anc, int anc_stride, hipStream_t stream, float stats_scale = 0.f, int logits_f16 = 0,
                 int slab_f16 = 0) {
  if (D->flex) return flex_decoder_step(D, S, rows, rows_pad, group, n_pad, pos, anc, anc_stride, stream, stats_scale);
  const smi_text_decoder_config& c = D->cfg;
  const int d = c.model_dim, f = c.ffn_inner_dim;
  float* x = S.x.as<float>();
  f16* h = S.h.as<f16>();
  f16* ctx = S.ctx.as<f16>();
  f16* ffn = S.ffn.as<f16>();
  const size_t slab = (size_t)rows_pad * 3 * d;  // elements per (layer, pos)
  const int P = S.kv_positions;
  void* parts = S.parts.p;
  const size_t part_stride = (size_t)rows_pad * d;  // elements
  // fp16 split-K slabs in the fp16 model's beam search (smi_text_decoder_set_slab_dtype; its own setting since round 5: the
  // reference rounds every sublayer output to fp16; here each partial is rounded once -- saturating at +-65504, never inf --,
  // the sum is formed in fp32 and added to the fp32 stream): half the 42 MB the FFN output projection wrote and the next
  // kernel read per layer at 1280 rows.  Tuning switch DEC_SLAB_F16 = 0 / 1 overrides (A/B runs).
  const int sf16 = slab_f16;
  // Split-K parts of the two N = d projections.  FFN output (K = f): 8 parts of 32 K slices = 160 lone units at
  // 1280 rows; 10 / 12 parts (unequal K ranges, 200 / 240 units) were traced and are NOT faster -- a unit's K loop
  // shrinks 20.0 -> 15.3 us but its 256 KiB slab store grows 4.3 -> 6.4 us (the slab writes run at the chip's
  // ~9.8 TB/s write path either way) and the fold reads 4 more slabs (profiles/r03_experiments.txt).
  // Attention output (K = d): as many parts as keep every 128x128 unit on a CU of its own (2 at 1280 rows: 160 units
  // on the lone-tile ring engine); SMI_DEC_KS_OUT overrides for A/B runs, 1 = no split, residual epilogue.
  const DecTuning& tu = S.tuning;
  // (an override the slab buffer or the K split cannot take falls back to the automatic choice)
  const bool ks_out_ok = tu.ks_out >= 1 && tu.ks_out <= kMaxParts && d % (64 * tu.ks_out) == 0;
  const int ks_out = ks_out_ok ? tu.ks_out : gemm_splitk_parts(rows_pad, d, d, kMaxParts);
  const int ks_ffn = tu.ks_ffn > 0 && tu.ks_ffn <= kMaxParts && f % (256 * tu.ks_ffn) == 0
                         ? tu.ks_ffn : (f % 512 == 0 ? 8 : (f % 256 == 0 ? 4 : 1));
  // A chain of a split call (S.chained) has too few FFN-inner tiles for the automatic engine choice (3 x 32 at 768
  // rows < the 144-tile crossover measured for ONE chain on an idle chip), but its neighbours fill the other CUs:
  // keep it on the 256x256 engine.  Its logits GEMM -- the one launch of a step that wants the whole chip -- may
  // leave CUs to the other chains (tu.logits_grid).
  const int ffn1_engine = tu.ffn1_engine >= 0 ? tu.ffn1_engine
                                              : (S.chained && rows_pad % 256 == 0 && f % 256 == 0 ? 2 : 0);
  const int logits_grid_env = tu.logits_grid;
  HIP_TRY(launch_dec_embed(S.tok.as<int32_t>(), D->embed.as<f16>(),
                           D->pos.as<float>() + (size_t)(pos + c.pos_offset) * d, c.embed_scale, x, rows, d,
                           c.vocab_size, stream));
  for (int l = 0; l < c.num_layers; ++l) {
    DecLayer& L = D->layers[l];
    f16* kvl = S.kv.as<f16>() + (size_t)l * P * slab;
    // x += FFN-out slabs of the previous layer (split-K), then LN1
    // (small batches: the CUs the row kernels leave idle read the layer's FFN matrices ahead, common.hpp: prefetch_range)
    HIP_TRY(launch_sum_layernorm(x, l ? parts : nullptr, ks_ffn, part_stride, nullptr, 1, L.ln1_w.as<float>(),
                                 L.ln1_b.as<float>(), c.ln_eps, h, rows_pad, d, stream, 0, 0, L.w_1.p, (size_t)f * d * 2, sf16));
    HIP_TRY(launch_gemm_tn(EPI_BIAS_F16, h, L.w_qkv.as<f16>(), L.b_qkv.as<float>(), kvl + (size_t)pos * slab,
                           rows_pad, 3 * d, d, 3 * d, stream));
    HIP_TRY(launch_dec_attention(kvl, anc, anc_stride, ctx, rows, rows_pad, d, c.num_heads, pos, stream));
    // the two N = d projections have too few tiles to fill 256 CUs at decode batch sizes:
    // split K into fp32 slabs that the next fused sum+LayerNorm folds into the residual stream
    if (ks_out == 1)  // no split: the projection adds into the fp32 residual stream itself
      HIP_TRY(launch_gemm_tn(EPI_RESID_F32, ctx, L.w_o.as<f16>(), L.b_o.as<float>(), x, rows_pad, d, d, d, stream));
    else
      HIP_TRY(launch_gemm_tn_splitk(ctx, L.w_o.as<f16>(), L.b_o.as<float>(), parts, rows_pad, d, d, ks_out, stream, 0, sf16));
    // the FFN runs on tile-major operands (common.hpp): LN output, hidden activation and both weights
    const int tm = D->ffn_tile_major;
    HIP_TRY(launch_sum_layernorm(x, ks_out == 1 ? nullptr : parts, ks_out, part_stride,
                                 S.cc.as<float>() + (size_t)l * n_pad * d, group, L.ln3_w.as<float>(), L.ln3_b.as<float>(),
                                 c.ln_eps, h, rows, d, stream, tm, 0, L.w_2.p, (size_t)f * d * 2, sf16));
    HIP_TRY(launch_gemm_tn(EPI_RELU_F16 | (tm ? GEMM_IN_TM | GEMM_OUT_TM : 0) | (ffn1_engine << 8), h, L.w_1.as<f16>(),
                           L.b_1.as<float>(), ffn, rows_pad, f, d, f, stream));
    HIP_TRY(launch_gemm_tn_splitk(ffn, L.w_2.as<f16>(), L.b_2.as<float>(), parts, rows_pad, d, f, ks_ffn, stream, tm, sf16));
  }
  const int ltm = D->embed_tm.p != nullptr;  // the tied projection reads a tile-major copy of the table
  HIP_TRY(launch_sum_layernorm(x, c.num_layers ? parts : nullptr, ks_ffn, part_stride, nullptr, 1,
                               D->lnf_w.as<float>(), D->lnf_b.as<float>(), c.ln_eps, h, rows_pad, d, stream, ltm, 0, nullptr, 0,
                               sf16));
  // beam search (stats_scale = 1 / temperature > 0): the GEMM also leaves per-tile softmax
  // statistics, so the candidate selection never re-reads the 1 MB logits rows
  GemmTileStats st{S.tile_max.as<float>(), S.tile_sum.as<float>(), stats_scale, (int)c.vocab_size};
  if (S.chained && logits_grid_env > 0) set_gemm_grid_cap(logits_grid_env);
  const hipError_t le =
      logits_f16 && ltm && stats_scale > 0.f
          ? launch_gemm_tn(EPI_BIAS_F16 | GEMM_IN_TM | GEMM_OUT_TM, h, D->embed_tm.as<f16>(), nullptr, S.logits.p, rows_pad,
                           (int)D->vocab_pad, d, (int)D->vocab_pad, stream, &st)
          : launch_gemm_tn(EPI_STORE_F32 | (ltm ? GEMM_IN_TM : 0), h, (ltm ? D->embed_tm : D->embed).as<f16>(), nullptr,
                           S.logits.p, rows_pad, (int)D->vocab_pad, d, (int)D->vocab_pad, stream,
                           stats_scale > 0.f ? &st : nullptr);
  set_gemm_grid_cap(0);
  HIP_TRY(le);
  return SMI_OK;
}

int ensure_step_workspace(smi_text_decoder* D, DecWork& S, int rows_pad, int positions, hipStream_t stream) {
  const smi_text_decoder_config& c = D->cfg;
  const size_t d = c.model_dim, f = c.ffn_inner_dim;
  const size_t before = S.x.bytes + S.h.bytes + S.ctx.bytes + S.ffn.bytes;
  const size_t es = D->act_bytes();
  S.tuning.read();
  HIP_TRY(S.x.reserve((size_t)rows_pad * d * 4));
  HIP_TRY(S.h.reserve((size_t)rows_pad * d * es));
  HIP_TRY(S.ctx.reserve((size_t)rows_pad * d * es));
  HIP_TRY(S.ffn.reserve((size_t)rows_pad * f * es));
  HIP_TRY(S.logits.reserve((size_t)rows_pad * D->vocab_pad * 4));
  HIP_TRY(S.parts.reserve((size_t)kMaxParts * rows_pad * d * 4));
  // kv cache for this call: [layers][positions][rows_pad][3d] (q|k|v slabs written by the QKV GEMM).
  // `positions` may be smaller than the generation cap: grow_kv() extends the cache when a call
  // actually decodes that far (the cap is max_seq_len = 512 for sentence vectors, fairseq2's
  // a * source_len + b rule; typical outputs stop after a few dozen tokens).
  {
    const size_t per_pos = (size_t)c.num_layers * rows_pad * 3 * d * es;
    const int have = per_pos ? (int)std::min<size_t>(S.kv.bytes / per_pos, (size_t)c.max_seq_len) : 0;
    if (have >= positions) {
      S.kv_positions = have;
    } else {
      HIP_TRY(S.kv.reserve(per_pos * positions));
      S.kv_positions = positions;
    }
  }
  if (S.x.bytes + S.h.bytes + S.ctx.bytes + S.ffn.bytes != before) {
    // tile-padding rows are read by the GEMMs: keep them finite
    // (stream-ordered: a chain's stream does not synchronise with the null stream a plain hipMemset runs on)
    HIP_TRY(hipMemsetAsync(S.x.p, 0, S.x.bytes, stream));
    HIP_TRY(hipMemsetAsync(S.h.p, 0, S.h.bytes, stream));
    HIP_TRY(hipMemsetAsync(S.ctx.p, 0, S.ctx.bytes, stream));
    HIP_TRY(hipMemsetAsync(S.ffn.p, 0, S.ffn.bytes, stream));
  }
  return SMI_OK;
}

// the decode loop reached the end of the kv allocation: move the slabs written so far into a cache
// with room for `positions` positions per layer (layout [layer][position][rows_pad][3d])
int grow_kv(smi_text_decoder* D, DecWork& S, int rows_pad, int positions, hipStream_t stream) {
  const smi_text_decoder_config& c = D->cfg;
  const size_t slab = (size_t)rows_pad * 3 * c.model_dim * D->act_bytes();  // bytes per (layer, position)
  const int old_p = S.kv_positions;
  if (positions <= old_p) return SMI_OK;
  DevBuf bigger;
  HIP_TRY(bigger.alloc((size_t)c.num_layers * positions * slab));
  for (int l = 0; l < c.num_layers; ++l)
    HIP_TRY(hipMemcpyAsync((char*)bigger.p + (size_t)l * positions * slab, (char*)S.kv.p + (size_t)l * old_p * slab,
                           (size_t)old_p * slab, hipMemcpyDeviceToDevice, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  S.kv = std::move(bigger);
  S.kv_positions = positions;
  return SMI_OK;
}

constexpr int kKvInitialPositions = 160;

// Number of independent chains for a beam-search call (smi_text_decoder_set_chains / SMI_DEC_CHAINS override).
// Measured on the `basic` decoder, beam 5, ms per step (profiles/r04_experiments.txt, experiment 1):
//   sentences (rows)   1 chain   2 chains   3 chains   4 chains
//   256  (1280)          3.88      4.16-4.25    4.97
//   512  (2560)          7.71      6.23
//   768  (3840)         10.03      9.68       9.18
//   1024 (5120)         13.57     13.22      12.61      13.22
// 1280 rows are 160 lone FFN tiles -- one round on 256 CUs -- and splitting them only makes every co-running launch 20-30 %
// slower (kernel trace: 65 % of the time two kernels in flight, each longer than alone); from 2560 rows on the FFN tiles need
// a second round and chains of ~1280 rows win, up to three of them (a fourth costs more in contention than it hides).
int decode_chains(const smi_text_decoder* D, int n, int beam) {
  const int env = tune(TUNE_DEC_CHAINS, 0);
  if (D->flex) return 1;
  const int64_t rows_pad = round_up((int64_t)n * beam, 256);
  const int automatic = rows_pad <= 2048 ? 1 : (int)std::min<int64_t>(3, (rows_pad + 1279) / 1280);
  int g = D->chains > 0 ? D->chains : (env > 0 ? env : automatic);
  g = std::min(g, kMaxChains);
  // every chain keeps at least two 256-row tiles of hypotheses: below that a chain's launches are all fixed cost
  while (g > 1 && (int64_t)((n + g - 1) / g) * beam < 384) --g;
  return std::max(g, 1);
}

// One decode chain: beam search for the n sentences of `emb` on workspace S and stream `stream` (the whole call, or
// one sentence group of a split call).  margins: device [n][2].
int generate_chain(smi_text_decoder* D, DecWork& S, const void* emb, int emb_dtype, int n, const int64_t* prompt,
                   int prompt_len, const smi_beam_search_params* bp, int32_t* out_tokens, int32_t* out_lens,
                   float* out_scores, float* margins, hipStream_t stream) {
  const smi_text_decoder_config& c = D->cfg;
  const int beam = bp->beam_size;
  const int max_len = bp->max_seq_len, min_len = bp->min_seq_len;

  const int rows = n * beam;
  const int rows_pad = (int)round_up(rows, 256), n_pad = (int)round_up(n, 256);
  const int stride = c.max_seq_len + 1;
  const int k2 = 2 * beam;
  if (int rc = ensure_step_workspace(D, S, rows_pad, std::min(max_len, kKvInitialPositions), stream)) return rc;
  HIP_TRY(S.tok.reserve((size_t)rows_pad * 4));
  HIP_TRY(S.cum.reserve((size_t)rows * 4));
  HIP_TRY(S.parent.reserve((size_t)rows * 4));
  HIP_TRY(S.new_tok.reserve((size_t)rows * 4));
  HIP_TRY(S.new_cum.reserve((size_t)rows * 4));
  HIP_TRY(S.nactive.reserve((size_t)n * 4));
  HIP_TRY(S.done.reserve((size_t)n * 4));
  HIP_TRY(S.ndone.reserve(4));
  HIP_TRY(S.fin_count.reserve((size_t)n * 4));
  HIP_TRY(S.fin_len.reserve((size_t)rows * 4));
  HIP_TRY(S.fin_score.reserve((size_t)rows * 4));
  HIP_TRY(S.fin_tok.reserve((size_t)rows * stride * 4));
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(S.anc[i].reserve((size_t)rows_pad * stride * 4));
    HIP_TRY(S.hist[i].reserve((size_t)rows_pad * stride * 4));
  }
  const int ntiles = (int)(D->vocab_pad / 256);
  HIP_TRY(S.pmax.reserve((size_t)rows * 4));
  HIP_TRY(S.psum.reserve((size_t)rows * 4));
  HIP_TRY(S.pval.reserve((size_t)rows * kVocabScanK2Max * 4));
  HIP_TRY(S.pidx.reserve((size_t)rows * kVocabScanK2Max * 4));
  HIP_TRY(S.tile_max.reserve((size_t)rows_pad * ntiles * 4));
  HIP_TRY(S.tile_sum.reserve((size_t)rows_pad * ntiles * 4));

  if (int rc = compute_cross_constants(D, S, emb, emb_dtype, n, n_pad, stream)) return rc;
  HIP_TRY(launch_beam_init(S.tok.as<int32_t>(), S.cum.as<float>(), S.nactive.as<int32_t>(),
                           S.done.as<int32_t>(), S.ndone.as<int32_t>(), S.fin_count.as<int32_t>(),
                           S.hist[0].as<int32_t>(), S.anc[0].as<int32_t>(), margins, rows, n, stride,
                           (int)prompt[0], stream));
  const float inv_temp = 1.0f / bp->temperature;
  // fp16 logits: the handle's setting (SMI_DEC_LOGITS_F16 = 0 / 1 overrides it: A/B runs), MFMA path with the tile-major table
  const int lf_env = tune(TUNE_DEC_LOGITS_F16, -1);
  const int logits_f16 = !D->flex && D->embed_tm.p != nullptr && (lf_env >= 0 ? lf_env != 0 : D->beam_logits_f16 != 0);
  const int sf_env = tune(TUNE_DEC_SLAB_F16, -1);
  const int slab_f16 = !D->flex && (sf_env >= 0 ? sf_env != 0 : D->beam_slab_f16 != 0);

  // everything one decode step enqueues (position pos; ancestry/history buffer pos & 1)
  auto enqueue_step = [&](int pos, hipStream_t s) -> int {
    const int cur = pos & 1, step_nr = pos + 1;
    if (int rc = decoder_step(D, S, rows, rows_pad, beam, n_pad, pos, S.anc[cur].as<int32_t>(), stride, s, inv_temp,
                              logits_f16, slab_f16))
      return rc;
    const bool forced_prompt = step_nr < prompt_len;
    const bool force_eos = !forced_prompt && step_nr == max_len - 1;
    // forced steps need only the softmax normaliser (the candidate is a given token): k2 = 0
    const bool free_step = !forced_prompt && !force_eos;
    HIP_TRY(launch_vocab_select(S.logits.as<float>(), (int)D->vocab_pad, logits_f16, rows, (int)c.vocab_size,
                                S.tile_max.as<float>(), S.tile_sum.as<float>(), ntiles, rows_pad, free_step ? k2 : 0, inv_temp,
                                c.pad_idx, c.eos_idx, c.unk_idx, free_step ? bp->unk_penalty : 0.f,
                                free_step && step_nr < min_len ? 1 : 0, S.pmax.as<float>(), S.psum.as<float>(),
                                S.pval.as<float>(), S.pidx.as<int>(), s));
    BeamStepArgs a{};
    a.tok = S.tok.as<int32_t>(); a.cum = S.cum.as<float>(); a.nactive = S.nactive.as<int32_t>();
    a.done = S.done.as<int32_t>(); a.ndone = S.ndone.as<int32_t>();
    a.parent = S.parent.as<int32_t>(); a.new_tok = S.new_tok.as<int32_t>(); a.new_cum = S.new_cum.as<float>();
    a.hist = S.hist[cur].as<int32_t>(); a.fin_tok = S.fin_tok.as<int32_t>(); a.fin_len = S.fin_len.as<int32_t>();
    a.fin_score = S.fin_score.as<float>(); a.fin_count = S.fin_count.as<int32_t>();
    a.margins = margins;
    a.logits = S.logits.as<float>(); a.ldl = (int)D->vocab_pad; a.logits_f16_tm = logits_f16;
    a.pmax = S.pmax.as<float>(); a.psum = S.psum.as<float>(); a.pval = S.pval.as<float>(); a.pidx = S.pidx.as<int>();
    a.nchunks = 1; a.n = n; a.beam = beam; a.k2 = k2; a.pos = pos; a.prompt_len = prompt_len;
    a.forced_tok = forced_prompt ? (int)prompt[step_nr] : -1; a.max_len = max_len;
    a.inv_temp = inv_temp; a.len_penalty = bp->len_penalty; a.normalize = bp->normalize_scores;
    a.eos_idx = c.eos_idx; a.hist_stride = stride;
    HIP_TRY(launch_beam_step(a, s));
    HIP_TRY(launch_beam_reorder(S.parent.as<int32_t>(), S.new_tok.as<int32_t>(), S.new_cum.as<float>(),
                                S.anc[cur].as<int32_t>(), S.anc[cur ^ 1].as<int32_t>(), S.hist[cur].as<int32_t>(),
                                S.hist[cur ^ 1].as<int32_t>(), S.tok.as<int32_t>(), S.cum.as<float>(), rows, stride,
                                pos, s));
    return SMI_OK;
  };

  // (A hipGraph cache of this step -- one captured graph per position, replayed on later calls -- was
  // measured at 256 x beam 5 and 16 x beam 5: 447.1 vs 447.3 ms and 185.4 vs 184.9 ms per 65 steps.  The
  // step is bound by the GPU front end's dependent-dispatch latency of ~180 short kernels, not by host
  // launch cost, so plain launches stay; fewer, fatter kernels are the lever.  DESIGN.md 3.4.)
  for (int pos = 0; pos + 1 < max_len; ++pos) {
    const int step_nr = pos + 1;
    if (pos >= S.kv_positions)
      if (int rc = grow_kv(D, S, rows_pad, std::min(max_len, 2 * S.kv_positions), stream)) return rc;
    if (int rc = enqueue_step(pos, stream)) return rc;
    // every 8 steps: has every sentence collected its `beam` hypotheses?
    const bool force_eos = step_nr >= prompt_len && step_nr == max_len - 1;
    if ((step_nr & 7) == 0 || force_eos) {
      int32_t nd = 0;
      HIP_TRY(hipMemcpyAsync(&nd, S.ndone.p, 4, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      if (nd >= n) break;
    }
  }
  HIP_TRY(launch_beam_output(S.fin_tok.as<int32_t>(), S.fin_len.as<int32_t>(), S.fin_score.as<float>(),
                             S.fin_count.as<int32_t>(), n, beam, stride, max_len, out_tokens, out_lens, out_scores,
                             margins, stream));
  return SMI_OK;
}

}  // namespace

extern "C" {

int smi_text_decoder_create(const smi_text_decoder_config* cfg, const smi_text_decoder_weights* w,
                            smi_text_decoder** out) {
  if (!cfg || !w || !out) return fail(SMI_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  bool flex = false;
  if (int rc = check_dec_cfg(*cfg, &flex)) return rc;
  if (!have_device()) return fail(SMI_ERR_NO_DEVICE, "no HIP device visible");
  if (cfg->num_layers > 0 && !w->layers) return fail(SMI_ERR_INVALID_ARG, "null layers");
  smi_text_decoder* D = new smi_text_decoder();
  D->cfg = *cfg;
  D->flex = flex;
  D->vocab_pad = round_up(cfg->vocab_size, 256);
  const int64_t d = cfg->model_dim, f = cfg->ffn_inner_dim, ci = cfg->input_dim;
  int rc = SMI_OK;
  // flex mode keeps every tensor fp32 (upload converts); the MFMA mode wants fp16 matrices
  auto up = [&](const smi_tensor& t, int64_t numel, bool f16w, DevBuf& dst, const char* name, int64_t pad = 0) {
    if (rc == SMI_OK) rc = upload(t, numel, f16w && !flex, dst, name, pad);
  };
  // the tied output projection multiplies by the embedding table: pad it to 256-row tiles
  up(w->embed, cfg->vocab_size * d, true, D->embed, "decoder_frontend.embed.weight", D->vocab_pad * d);
  if (rc == SMI_OK && !flex && d % 256 == 0) {  // [vocab_pad][d] -> tile-major (common.hpp), ~0.5 GB for NLLB
    rc = upload(w->embed, cfg->vocab_size * d, true, D->embed_tm, "decoder_frontend.embed.weight", D->vocab_pad * d);
    if (rc == SMI_OK) rc = to_tile_major(D->embed_tm, (int)D->vocab_pad, (int)d);
  }
  up(w->pos_table, (int64_t)(cfg->max_seq_len + cfg->pos_offset) * d, false, D->pos, "pos_table");
  up(w->final_layer_norm_w, d, false, D->lnf_w, "decoder.layer_norm.weight");
  up(w->final_layer_norm_b, d, false, D->lnf_b, "decoder.layer_norm.bias");
  D->ffn_tile_major = !flex && d % 256 == 0 && f % 256 == 0;
  D->layers.resize(cfg->num_layers);
  for (int l = 0; l < cfg->num_layers && rc == SMI_OK; ++l) {
    const smi_text_decoder_layer& s = w->layers[l];
    DecLayer& L = D->layers[l];
    up(s.self_attn_layer_norm_w, d, false, L.ln1_w, "self_attn_layer_norm.weight");
    up(s.self_attn_layer_norm_b, d, false, L.ln1_b, "self_attn_layer_norm.bias");
    up(s.ffn_layer_norm_w, d, false, L.ln3_w, "ffn_layer_norm.weight");
    up(s.ffn_layer_norm_b, d, false, L.ln3_b, "ffn_layer_norm.bias");
    up(s.out_w, d * d, true, L.w_o, "self_attn.output_proj.weight");
    up(s.out_b, d, false, L.b_o, "self_attn.output_proj.bias");
    up(s.cross_v_w, d * ci, true, L.wc_v, "encoder_decoder_attn.v_proj.weight");  // [model_dim, input_dim]
    up(s.cross_v_b, d, false, L.bc_v, "encoder_decoder_attn.v_proj.bias");
    up(s.cross_out_w, d * d, true, L.wc_o, "encoder_decoder_attn.output_proj.weight");
    up(s.cross_out_b, d, false, L.bc_o, "encoder_decoder_attn.output_proj.bias");
    up(s.ffn_inner_w, f * d, true, L.w_1, "ffn.inner_proj.weight");
    up(s.ffn_inner_b, f, false, L.b_1, "ffn.inner_proj.bias");
    up(s.ffn_out_w, d * f, true, L.w_2, "ffn.output_proj.weight");
    up(s.ffn_out_b, d, false, L.b_2, "ffn.output_proj.bias");
    if (rc == SMI_OK && D->ffn_tile_major) {
      rc = to_tile_major(L.w_1, (int)f, (int)d);
      if (rc == SMI_OK) rc = to_tile_major(L.w_2, (int)d, (int)f);
    }
    if (rc == SMI_OK) {
      DevBuf tq, tk, tv, bq, bk, bv;
      up(s.q_w, d * d, true, tq, "self_attn.q_proj.weight");
      up(s.k_w, d * d, true, tk, "self_attn.k_proj.weight");
      up(s.v_w, d * d, true, tv, "self_attn.v_proj.weight");
      up(s.q_b, d, false, bq, "self_attn.q_proj.bias");
      up(s.k_b, d, false, bk, "self_attn.k_proj.bias");
      up(s.v_b, d, false, bv, "self_attn.v_proj.bias");
      if (rc == SMI_OK) {
        const size_t wes = flex ? 4 : 2;
        hipError_t he = L.w_qkv.alloc((size_t)3 * d * d * wes);
        if (he == hipSuccess) he = L.b_qkv.alloc((size_t)3 * d * 4);
        const size_t wb = (size_t)d * d * wes, bb = (size_t)d * 4;
        if (he == hipSuccess) he = hipMemcpy(L.w_qkv.p, tq.p, wb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy((char*)L.w_qkv.p + wb, tk.p, wb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy((char*)L.w_qkv.p + 2 * wb, tv.p, wb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy(L.b_qkv.p, bq.p, bb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy((char*)L.b_qkv.p + bb, bk.p, bb, hipMemcpyDeviceToDevice);
        if (he == hipSuccess) he = hipMemcpy((char*)L.b_qkv.p + 2 * bb, bv.p, bb, hipMemcpyDeviceToDevice);
        if (he != hipSuccess)
          rc = fail(he == hipErrorOutOfMemory ? SMI_ERR_OOM : SMI_ERR_HIP, "packing qkv: %s", hipGetErrorString(he));
      }
    }
  }
  if (rc != SMI_OK) {
    delete D;
    return rc;
  }
  D->weight_bytes = (int64_t)(D->embed.bytes + D->embed_tm.bytes + D->pos.bytes);
  for (auto& L : D->layers)
    D->weight_bytes += (int64_t)(L.w_qkv.bytes + L.w_o.bytes + L.wc_v.bytes + L.wc_o.bytes + L.w_1.bytes + L.w_2.bytes);
  *out = D;
  return SMI_OK;
}

void smi_text_decoder_destroy(smi_text_decoder* dec) {
  if (!dec) return;
  (void)hipDeviceSynchronize();
  delete dec;
}

int smi_text_decoder_logits(smi_text_decoder* D, const void* emb, int32_t emb_dtype, int32_t n,
                            const int64_t* prev_tokens, int32_t t, float* out_logits, void* stream_v) {
  if (!D || !emb || !prev_tokens || !out_logits) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (n <= 0 || t <= 0) return fail(SMI_ERR_INVALID_ARG, "empty input");
  if (t > D->cfg.max_seq_len) return fail(SMI_ERR_INVALID_ARG, "t=%d exceeds max_seq_len %d", t, D->cfg.max_seq_len);
  if (emb_dtype != SMI_F32 && emb_dtype != SMI_F16) return fail(SMI_ERR_INVALID_ARG, "bad emb dtype");
  hipStream_t stream = (hipStream_t)stream_v;
  DecWork& S = D->ws[0];
  S.chained = false;
  const int rows_pad = (int)round_up(n, 256), n_pad = rows_pad;
  if (int rc = ensure_step_workspace(D, S, rows_pad, t, stream)) return rc;
  HIP_TRY(S.tok.reserve((size_t)rows_pad * 4));
  // teacher forcing, one hypothesis per sentence: the ancestry is the identity and is never read
  // for j < pos only through anc[r][j] = r
  const int stride = D->cfg.max_seq_len + 1;
  HIP_TRY(S.anc[0].reserve((size_t)rows_pad * stride * 4));
  {
    std::vector<int32_t> ident((size_t)n * stride);
    for (int r = 0; r < n; ++r)
      for (int j = 0; j < stride; ++j) ident[(size_t)r * stride + j] = r;
    HIP_TRY(hipMemcpyAsync(S.anc[0].p, ident.data(), ident.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
  }
  if (int rc = compute_cross_constants(D, S, emb, emb_dtype, n, n_pad, stream)) return rc;
  for (int pos = 0; pos < t; ++pos) {
    HIP_TRY(launch_gather_tokens(prev_tokens, t, pos, S.tok.as<int32_t>(), n, stream));
    if (int rc = decoder_step(D, S, n, rows_pad, 1, n_pad, pos, S.anc[0].as<int32_t>(), stride, stream)) return rc;
    HIP_TRY(hipMemcpy2DAsync(out_logits + (size_t)pos * D->cfg.vocab_size, (size_t)t * D->cfg.vocab_size * 4,
                             S.logits.p, (size_t)D->vocab_pad * 4, (size_t)D->cfg.vocab_size * 4, n,
                             hipMemcpyDeviceToDevice, stream));
  }
  return SMI_OK;
}

int smi_text_decoder_generate(smi_text_decoder* D, const void* emb, int32_t emb_dtype, int32_t n,
                              const int64_t* prompt, int32_t prompt_len, const smi_beam_search_params* bp,
                              int32_t* out_tokens, int32_t* out_lens, float* out_scores, void* stream_v) {
  if (!D || !emb || !prompt || !bp || !out_tokens || !out_lens || !out_scores)
    return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (n <= 0 || prompt_len <= 0) return fail(SMI_ERR_INVALID_ARG, "empty input");
  if (emb_dtype != SMI_F32 && emb_dtype != SMI_F16) return fail(SMI_ERR_INVALID_ARG, "bad emb dtype");
  const smi_text_decoder_config& c = D->cfg;
  const int beam = bp->beam_size;
  if (beam < 1 || beam > 8) return fail(SMI_ERR_UNSUPPORTED, "beam_size %d outside [1,8]", beam);
  if (2 * beam >= c.vocab_size) return fail(SMI_ERR_UNSUPPORTED, "vocabulary too small for beam %d", beam);
  const int max_len = bp->max_seq_len;
  if (max_len > c.max_seq_len || max_len <= prompt_len)
    return fail(SMI_ERR_INVALID_ARG, "max_seq_len %d must be in (prompt_len %d, model max %d]", max_len, prompt_len,
                c.max_seq_len);
  if (!(bp->temperature > 0.f)) return fail(SMI_ERR_INVALID_ARG, "temperature must be positive");
  for (int i = 0; i < prompt_len; ++i)
    if (prompt[i] < 0 || prompt[i] >= c.vocab_size) return fail(SMI_ERR_INVALID_ARG, "prompt token out of range");
  hipStream_t stream = (hipStream_t)stream_v;
  HIP_TRY(D->margins.reserve((size_t)n * 2 * 4));
  D->margins_n = n;
  float* margins = D->margins.as<float>();

  // Independent chains.  Sentences do not interact, and at decode batch sizes a step is ~170
  // dependent launches of lone tiles: ~8.5 us of every launch is fixed cost (dispatch, pipeline fill, epilogue) and the
  // two FFN projections occupy 160 of 256 CUs.  Two or three sentence groups, each with its own workspace, KV cache,
  // beam state, stream and host thread, put their launch gaps and idle CUs under each other's K loops; the row-count
  // bound kernels (logits GEMM, single-query attention, slab folds) shrink with the group.  Beam bookkeeping is per
  // sentence, so the hypotheses are those of the single chain up to the fp32 summation order of the split-K slabs
  // (the number of K parts follows the group's row count).
  const int chains = decode_chains(D, n, beam);
  if (chains <= 1) {
    D->ws[0].chained = false;
    return generate_chain(D, D->ws[0], emb, emb_dtype, n, prompt, prompt_len, bp, out_tokens, out_lens, out_scores,
                          margins, stream);
  }
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (!D->fork_ev) HIP_TRY(hipEventCreateWithFlags(&D->fork_ev, hipEventDisableTiming));
  for (int g = 0; g < chains; ++g) {
    DecWork& S = D->ws[g];
    if (!S.stream) HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    if (!S.done_ev) HIP_TRY(hipEventCreateWithFlags(&S.done_ev, hipEventDisableTiming));
  }
  // fork: every chain starts behind the work already queued on the caller's stream (the embeddings' producer)
  HIP_TRY(hipEventRecord(D->fork_ev, stream));
  const size_t emb_row = (size_t)c.input_dim * (emb_dtype == SMI_F32 ? 4 : 2);
  const int per = (n + chains - 1) / chains;
  int rcs[kMaxChains] = {};
  std::string errs[kMaxChains];
  std::thread workers[kMaxChains];
  for (int g = 0; g < chains; ++g) {
    const int s0 = g * per, ng = std::min(per, n - s0);
    if (ng <= 0) continue;
    workers[g] = std::thread([&, g, s0, ng] {
      DecWork& S = D->ws[g];
      S.chained = true;
      int rc = SMI_OK;
      hipError_t he = hipSetDevice(dev);
      if (he == hipSuccess) he = hipStreamWaitEvent(S.stream, D->fork_ev, 0);
      if (he != hipSuccess) rc = fail(SMI_ERR_HIP, "chain %d set-up: %s", g, hipGetErrorString(he));
      if (rc == SMI_OK)
        rc = generate_chain(D, S, (const char*)emb + (size_t)s0 * emb_row, emb_dtype, ng, prompt, prompt_len, bp,
                            out_tokens + (size_t)s0 * beam * max_len, out_lens + (size_t)s0 * beam,
                            out_scores + (size_t)s0 * beam, margins + 2 * (size_t)s0, S.stream);
      if (rc == SMI_OK && (he = hipEventRecord(S.done_ev, S.stream)) != hipSuccess)
        rc = fail(SMI_ERR_HIP, "chain %d: %s", g, hipGetErrorString(he));
      rcs[g] = rc;
      if (rc != SMI_OK) errs[g] = last_error();  // thread-local: hand the text to the calling thread
    });
  }
  for (int g = 0; g < chains; ++g)
    if (workers[g].joinable()) workers[g].join();
  for (int g = 0; g < chains; ++g)
    if (rcs[g] != SMI_OK) {
      for (int h = 0; h < chains; ++h)
        if (D->ws[h].stream) (void)hipStreamSynchronize(D->ws[h].stream);
      return fail(rcs[g], "%s", errs[g].c_str());
    }
  // join: the caller's stream continues behind every chain
  for (int g = 0; g < chains; ++g)
    if (g * per < n)
      HIP_TRY(hipStreamWaitEvent(stream, D->ws[g].done_ev, 0));
  return SMI_OK;
}

int smi_text_decoder_set_chains(smi_text_decoder* D, int32_t chains) {
  if (!D) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (chains < 0 || chains > kMaxChains) return fail(SMI_ERR_INVALID_ARG, "chains %d outside [0, %d]", chains, kMaxChains);
  D->chains = chains;
  return SMI_OK;
}

int smi_text_decoder_set_beam_logits_dtype(smi_text_decoder* D, int32_t dtype) {
  if (!D) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (dtype != SMI_F16 && dtype != SMI_F32) return fail(SMI_ERR_INVALID_ARG, "dtype %d: SMI_F16 or SMI_F32", dtype);
  D->beam_logits_f16 = dtype == SMI_F16;
  return SMI_OK;
}

int smi_text_decoder_set_slab_dtype(smi_text_decoder* D, int32_t dtype) {
  if (!D) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (dtype != SMI_F16 && dtype != SMI_F32) return fail(SMI_ERR_INVALID_ARG, "dtype %d: SMI_F16 or SMI_F32", dtype);
  D->beam_slab_f16 = dtype == SMI_F16;
  return SMI_OK;
}

int smi_text_decoder_last_margins(smi_text_decoder* D, float* out_margins, int32_t n, void* stream_v) {
  if (!D || !out_margins) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (n <= 0 || n != D->margins_n || !D->margins.p)
    return fail(SMI_ERR_INVALID_ARG, "n=%d does not match the last generate() call (%d sentences)", n, D->margins_n);
  HIP_TRY(hipMemcpyAsync(out_margins, D->margins.p, (size_t)n * 2 * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream_v));
  return SMI_OK;
}

int smi_text_decoder_sample(smi_text_decoder* D, const void* emb, int32_t emb_dtype, int32_t n,
                            const int64_t* prompt, int32_t prompt_len, const smi_sampling_params* sp,
                            int32_t* out_tokens, int32_t* out_lens, float* out_scores, void* stream_v) {
  if (!D || !emb || !prompt || !sp || !out_tokens || !out_lens || !out_scores)
    return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (n <= 0 || prompt_len <= 0) return fail(SMI_ERR_INVALID_ARG, "empty input");
  if (emb_dtype != SMI_F32 && emb_dtype != SMI_F16) return fail(SMI_ERR_INVALID_ARG, "bad emb dtype");
  const smi_text_decoder_config& c = D->cfg;
  if (sp->sampler == SMI_SAMPLER_TOP_K) {
    if (sp->top_k < 1) return fail(SMI_ERR_INVALID_ARG, "top_k must be >= 1");
  } else if (sp->sampler == SMI_SAMPLER_TOP_P) {
    if (!(sp->top_p > 0.f && sp->top_p <= 1.f)) return fail(SMI_ERR_INVALID_ARG, "top_p must be in (0, 1]");
  } else {
    return fail(SMI_ERR_INVALID_ARG, "unknown sampler %d", sp->sampler);
  }
  if (c.vocab_size > (1 << 18))
    return fail(SMI_ERR_UNSUPPORTED, "vocab %d: sampling covers up to 2^18 tokens", (int)c.vocab_size);
  const int max_len = sp->max_seq_len, min_len = sp->min_seq_len;
  if (max_len > c.max_seq_len || max_len <= prompt_len)
    return fail(SMI_ERR_INVALID_ARG, "max_seq_len %d must be in (prompt_len %d, model max %d]", max_len, prompt_len,
                c.max_seq_len);
  if (!(sp->temperature > 0.f)) return fail(SMI_ERR_INVALID_ARG, "temperature must be positive");
  for (int i = 0; i < prompt_len; ++i)
    if (prompt[i] < 0 || prompt[i] >= c.vocab_size) return fail(SMI_ERR_INVALID_ARG, "prompt token out of range");
  hipStream_t stream = (hipStream_t)stream_v;

  // one hypothesis per sentence (fairseq2 num_gens = 1): rows = sentences, identity ancestry
  DecWork& S = D->ws[0];
  S.chained = false;
  const int rows_pad = (int)round_up(n, 256), n_pad = rows_pad;
  const int stride = c.max_seq_len + 1;
  if (int rc = ensure_step_workspace(D, S, rows_pad, std::min(max_len, kKvInitialPositions), stream)) return rc;
  HIP_TRY(S.tok.reserve((size_t)rows_pad * 4));
  HIP_TRY(S.cum.reserve((size_t)n * 4));
  HIP_TRY(S.done.reserve((size_t)n * 4));
  HIP_TRY(S.ndone.reserve(4));
  HIP_TRY(S.new_tok.reserve((size_t)n * 4));
  HIP_TRY(S.new_cum.reserve((size_t)n * 4));
  HIP_TRY(S.anc[0].reserve((size_t)rows_pad * stride * 4));
  {
    std::vector<int32_t> ident((size_t)n * stride), first((size_t)n, (int32_t)prompt[0]);
    for (int r = 0; r < n; ++r)
      for (int j = 0; j < stride; ++j) ident[(size_t)r * stride + j] = r;
    HIP_TRY(hipMemcpyAsync(S.anc[0].p, ident.data(), ident.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(S.tok.p, first.data(), first.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
  }
  HIP_TRY(hipMemsetAsync(S.cum.p, 0, (size_t)n * 4, stream));
  HIP_TRY(hipMemsetAsync(S.done.p, 0, (size_t)n * 4, stream));
  HIP_TRY(hipMemsetAsync(S.ndone.p, 0, 4, stream));
  HIP_TRY(hipMemsetAsync(out_tokens, 0xff, (size_t)n * max_len * 4, stream));
  HIP_TRY(hipMemsetAsync(out_lens, 0, (size_t)n * 4, stream));
  HIP_TRY(hipMemsetAsync(out_scores, 0, (size_t)n * 4, stream));
  if (int rc = compute_cross_constants(D, S, emb, emb_dtype, n, n_pad, stream)) return rc;

  for (int pos = 0; pos + 1 < max_len; ++pos) {
    const int step_nr = pos + 1;
    if (pos >= S.kv_positions)
      if (int rc = grow_kv(D, S, rows_pad, std::min(max_len, 2 * S.kv_positions), stream)) return rc;
    if (int rc = decoder_step(D, S, n, rows_pad, 1, n_pad, pos, S.anc[0].as<int32_t>(), stride, stream)) return rc;
    const bool forced_prompt = step_nr < prompt_len;
    const bool force_eos = !forced_prompt && step_nr == max_len - 1;
    SampleRowsArgs a{};
    a.logits = S.logits.as<float>(); a.ld = D->vocab_pad; a.rows = n; a.vocab = (int)c.vocab_size;
    a.inv_temp = 1.0f / sp->temperature; a.pad_idx = c.pad_idx; a.eos_idx = c.eos_idx;
    a.block_eos = !forced_prompt && !force_eos && step_nr < min_len;
    a.unk_idx = c.unk_idx; a.unk_penalty = sp->unk_penalty;
    a.forced_tok = forced_prompt ? (int)prompt[step_nr] : (force_eos ? c.eos_idx : -1);
    a.mode = sp->sampler; a.top_k = sp->top_k; a.top_p = sp->top_p; a.z = nullptr; a.seed = sp->seed; a.step = step_nr;
    a.done = S.done.as<int32_t>(); a.out_tok = S.new_tok.as<int32_t>(); a.out_logp = S.new_cum.as<float>();
    HIP_TRY(launch_sample_rows(a, stream));
    SampleUpdateArgs u{};
    u.samp_tok = S.new_tok.as<int32_t>(); u.samp_logp = S.new_cum.as<float>(); u.tok = S.tok.as<int32_t>();
    u.cum = S.cum.as<float>(); u.done = S.done.as<int32_t>(); u.ndone = S.ndone.as<int32_t>();
    u.out_tokens = out_tokens; u.out_lens = out_lens; u.out_scores = out_scores; u.n = n; u.out_stride = max_len;
    u.pos = pos; u.prompt_len = prompt_len; u.eos_idx = c.eos_idx; u.normalize = sp->normalize_scores;
    u.len_penalty = sp->len_penalty;
    HIP_TRY(launch_sample_update(u, stream));
    if ((step_nr & 7) == 0 && !force_eos) {
      int32_t nd = 0;
      HIP_TRY(hipMemcpyAsync(&nd, S.ndone.p, 4, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      if (nd >= n) break;
    }
  }
  return SMI_OK;
}

}  // extern "C"
