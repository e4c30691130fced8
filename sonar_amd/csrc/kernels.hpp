// Internal launch API of the HIP kernels (host side).  Everything here is
// stream-ordered and never synchronises the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "tuning.hpp"

namespace smi {
// compute units of the current device (per-device atomic cache; gemm.hip)
int num_cus();
// one-time per-DEVICE initialisation flag (function attributes, device-resident constants): a process
// may drive several GPUs, so a plain `static bool` would skip the set-up on every device but the first
struct DeviceOnce {
  std::atomic<unsigned long long> mask{0};
  static int dev() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 63;
  }
  bool done() const { return (mask.load(std::memory_order_acquire) >> dev()) & 1ull; }
  void set() { mask.fetch_or(1ull << dev(), std::memory_order_release); }
};
// ---- generic-dimension kernels (flex.hip): fp32 activations and weights, any model_dim / head_dim <= 256 ----
// Y[M,N] = act(X[M,K] . W[N,K]^T + bias) (+ R); act 0 none, 1 relu; R may alias Y
hipError_t launch_flex_linear(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M,
                              int N, int K, int act, const float* R, int ldr, hipStream_t stream);
hipError_t launch_flex_layernorm(const float* x, const float* w, const float* b, float eps, float* y, int rows, int d,
                                 hipStream_t stream);
// x[r] = table[id[r]] * scale (+ pe[pos + pos_offset]); ids64 ([n, s] batch, pos = r % s, fixed_pos = -1) or ids32
// (one token per row at position fixed_pos); out-of-range ids raise *bad
hipError_t launch_flex_embed(const int64_t* ids64, const int32_t* ids32, const float* table, const float* pe,
                             float scale, float* x, int rows, int d, int s, int pos_offset, int fixed_pos,
                             int64_t vocab, int32_t* bad, hipStream_t stream);
// out[b, i] = softmax(q[b, i] . k[b, :klens[b]]^T / sqrt(hd)) . v per head; q rows b * sq + i, k / v rows b * sk + j
hipError_t launch_flex_attention(const float* q, int ldq, const float* k, const float* v, int ldk, float* out, int ldo,
                                 int items, int sq, int sk, const int32_t* klens, int heads, int hd, int causal,
                                 hipStream_t stream);
hipError_t launch_flex_dec_attention(const float* kv, const int32_t* anc, int anc_stride, float* ctx, int rows,
                                     int rows_pad, int d, int heads, int pos, hipStream_t stream);
hipError_t launch_flex_pool(const float* x, const int32_t* lens, int pooling, float* out, int n, int s, int d,
                            hipStream_t stream);
hipError_t launch_flex_tile_stats(const float* logits, int ld, int rows, int vocab, float scale, float* tile_max,
                                  float* tile_sum, int stat_rows, hipStream_t stream);
hipError_t launch_flex_add_rows(float* x, const float* c, int rows, int d, int group, hipStream_t stream);
hipError_t launch_flex_store_encoded(const float* x, const int32_t* lens, int n, int s, int d, void* out, bool out_f16,
                                     hipStream_t stream);

}  // namespace smi

namespace smi {

typedef _Float16 f16;

enum GemmEpilogue {
  EPI_BIAS_F16 = 0, EPI_RELU_F16 = 1, EPI_RESID_F32 = 2, EPI_STORE_F32 = 3,
  EPI_RESID_HALF_F32 = 4, EPI_SILU_F16 = 5, EPI_GLU_F16 = 6, EPI_TANH_F16 = 7, EPI_RESID_F16 = 8,
  EPI_RESID_HALF_F16 = 9
};

// layout flags OR-ed into epi_sel (tile-major layout: common.hpp tm_offset)
constexpr int GEMM_IN_TM = 1 << 12;   // X and W are tile-major (M, N % 256 == 0)
constexpr int GEMM_OUT_TM = 1 << 13;  // fp16 output is tile-major with K = N (needs GEMM_IN_TM, ldo == N)

// Optional per-(row, 256-column tile) softmax statistics of an EPI_STORE_F32 GEMM on the 256x256
// engine (the decoder's logits GEMM): tile_max[N/256][M] = max_n v, tile_sum[N/256][M] =
// sum_n exp(v - max) with v = C[m][n] * scale over the columns n < valid_n of the tile
// (tile-major so that a tile's 256 rows are one coalesced 1 KiB store).
struct GemmTileStats {
  float* tile_max;
  float* tile_sum;
  float scale;
  int valid_n;
};

// LayerNorm folded into the GEMMs around it (256x256 engine, tile-major fp16 residual stream).
// With h = LN(x) = (x - mean) * rstd * g + b,   h . W^T = rstd * (x . (W (.) g)^T - mean * c1) + c2,
// c1[n] = sum_k g[k] W[n][k], c2[n] = sum_k b[k] W[n][k] + bias[n]: the consuming GEMM multiplies the residual stream x
// ITSELF by the pre-scaled weights and applies the row statistics in its epilogue; the producing (residual) GEMM leaves
// the per-row (sum, sum of squares) of the stream it has just written.  No LayerNorm launch, no `h` round trip.
//  * producer (EPI_RESID_F16, tile-major stream): part_out[N/256][M] <- partial sums of the new rows over the tile's columns
//  * consumer (EPI_BIAS_F16 / EPI_RELU_F16, tile-major in/out): X = the stream, W = the pre-scaled weights, bias = c2,
//    part_in[nparts][M] = the partial sums of x's rows, c1 as above
struct GemmLnFold {
  float2* part_out;
  const float2* part_in;
  const float* c1;
  int nparts;
  float inv_k;  // 1 / (row width of the stream)
  float eps;
  // centered != 0: the pre-scaled weight rows were centred (W'' = W (.) g - c1 / K), which moves the "- mean * c1" term
  // into the GEMM itself (sum_k x_k W''_nk = sum_k (x_k - mean) (W (.) g)_nk): the epilogue is out = rstd * acc + c2.
  // Costs: the fp16 rounding of W'' leaves a residue r_n = sum_k W''_nk (~4.5e-3 of a weight's magnitude at K = 1024)
  // that multiplies the row mean; relative to the signal that is ~1.4e-4 * |mean| / std.
  int centered;
};

// C = X[M,K] * W[N,K]^T (+bias, epilogue).  M%128==0, N%128==0, K%64==0.
hipError_t launch_gemm_tn(int epi_sel, const f16* X, const f16* W, const float* bias, void* out, int M,
                          int N, int K, int ldo, hipStream_t stream, const GemmTileStats* stats = nullptr,
                          const GemmLnFold* fold = nullptr);

// The 4-wave 256x256 engine (gemm_v2.hip): tile-major fp16 in / out, epi in {bias, relu, silu}, optional LayerNorm-fold
// consumer.  gemm_v2_fits: the launch qualifies (shape, switches); launch_gemm_tn routes to it by itself.
bool gemm_v2_fits(int epi, int M, int N, int K, const float* bias, const GemmLnFold* fold);
hipError_t launch_gemm_v2(int epi, const f16* X, const f16* W, const float* bias, f16* out, int M, int N, int K,
                          hipStream_t stream, const GemmLnFold* fold);

// ... and the decoder's logits projection with the fused softmax statistics (fp16 tile-major logits, no bias, stats->scale > 0)
bool gemm_v2_stats_fits(int M, int N, int K, const GemmTileStats* stats);
hipError_t launch_gemm_v2_stats(const f16* X, const f16* W, f16* out, int M, int N, int K, hipStream_t stream,
                                const GemmTileStats* stats, int grid_cap);

// 160x256 lone units of the 4-wave engine (gemm_v2_lone.hip): M % 1280 == 0 rows, tile-major operands, every unit on its own CU.
// mode 1 / 2: out = tile-major fp16 relu(X W^T + bias) / X W^T + bias (ksplit 1); mode 0: out = row-major fp16 split-K slabs
// [ksplit][M][N].  (The 160x256 name is historic: units are 128 / 160 / 192 rows.)
bool gemm_v2_lone_fits(int M, int N, int K, int ksplit);
hipError_t launch_gemm_v2_lone(int mode, const f16* X, const f16* W, const float* bias, void* out, int M, int N, int K, int ksplit,
                               hipStream_t stream);

// in_tm: X and W tile-major (common.hpp; M, N % 256 == 0)
// slab_f16: fp16 slabs [ksplit][M][N] instead of fp32 ones (the consumers take the same flag)
hipError_t launch_gemm_tn_splitk(const f16* X, const f16* W, const float* bias, void* parts, int M,
                                 int N, int K, int ksplit, hipStream_t stream, int in_tm = 0, int slab_f16 = 0);
// cap the persistent grid of the calling thread's next 256x256-engine launches (0 = no cap)
void set_gemm_grid_cap(int workgroups);
// number of K parts for a decode-time split-K projection (gemm.hip): every unit on its own CU, <= max_parts
int gemm_splitk_parts(int M, int N, int K, int max_parts);

// x[row(n,p), :] = E[ids[n*S+p], :] * scale + PE[p + pos_offset, :]   (packed rows)
// x_f16: the residual stream x is fp16 (SMI_ENC_FP16_RESIDUAL) instead of fp32
hipError_t launch_embed_pack(const int64_t* ids, const int32_t* cu_seqlens, const f16* table,
                             const float* pos_table, float scale, int pos_offset, void* x, int N,
                             int S, int max_len, int d, int64_t vocab, hipStream_t stream, int x_f16 = 0,
                             int32_t* bad = nullptr, int x_tm = 0);

// h[r,:] = f16(LN(x[r,:]) * w + b)
// out_tm: h is written tile-major (rows rounded up to 256 must be allocated)
hipError_t launch_layernorm(const void* x, const float* w, const float* b, float eps, f16* h,
                            int rows, int d, hipStream_t stream, int out_tm = 0, int x_f16 = 0, int x_tm = 0);
// x_tm (embed_pack, layernorm, ln_pool): the fp16 residual stream x is itself tile-major (common.hpp), so that
// the residual GEMM epilogues can read-modify-write it straight from the accumulators (GEMM_OUT_TM + EPI_RESID_F16)
// dst (tile-major) <- src (row-major [rows][K]), rows % 256 == 0, K % 32 == 0; and the inverse
hipError_t launch_pack_tile_major(const f16* src, f16* dst, int rows, int K, int inverse, hipStream_t stream);

// Final LN + masked pooling (mean) over each sentence's packed rows.
// out: [N, d] in fp16 or fp32; encoded (optional): [N, S, d] same dtype, pads zeroed.
hipError_t launch_ln_pool(const void* x, const float* w, const float* b, float eps,
                          const int32_t* cu_seqlens, void* out, int out_is_f32, void* encoded, int N,
                          int S, int d, int pooling, hipStream_t stream, int x_f16 = 0, int x_tm = 0);

// Self-attention over packed rows. qkv: [T, 3*d] (q | k | v), ctx: [T, d].  head_dim 64.
// ctx_tm bit 0: ctx is written tile-major; bit 1: qkv is read tile-major (K = 3d).
hipError_t launch_attention(const f16* qkv, const int32_t* cu_seqlens, f16* ctx, int N, int max_len,
                            int d, int heads, hipStream_t stream, int ctx_tm = 0);

// dst_f16[i] = f16(src_f32[i])
hipError_t launch_f32_to_f16(const float* src, f16* dst, size_t n, hipStream_t stream);
// element-wise cast, dtypes 0 = fp32, 1 = fp16, 2 = bf16 (smi_dtype)
hipError_t launch_cast(const void* src, int src_dtype, void* dst, int dst_dtype, size_t n, hipStream_t stream);
// x[i] += sum_z parts[z * part_elems + i], i < n (n % 8 == 0); x fp16 or fp32 (rowops.hip)
// LayerNorm-fold helpers (rowops.hip): Wf = f16(W (.) g), c1 = row sums of Wf, c2 = W . b + bias  (W [N][K] row-major)
// centered: Wf = f16(W (.) g - c1 / K) (row-centred), c1 = the rounding residue sum_k Wf[n][k]
hipError_t launch_ln_fold_prep(const f16* W, const float* g, const float* b, const float* bias, f16* Wf, float* c1,
                               float* c2, int N, int K, int centered, hipStream_t stream);
// part[0][r] = (sum, sum of squares) of row r of the tile-major fp16 stream x [M][d]; part[1..nparts-1][r] = 0
hipError_t launch_row_stats_tm(const f16* x_tm, float2* part, int M, int d, int nparts, hipStream_t stream);
hipError_t launch_fold_residual(void* x, int x_f16, const void* parts, int nparts, size_t part_elems, size_t n,
                                hipStream_t stream, int parts_f16 = 0);
// dst_f32[i] = float(src_f16[i])
hipError_t launch_f16_to_f32(const f16* src, float* dst, size_t n, hipStream_t stream);

// xsim: row-normalise then mine top-k cosine neighbours of each X row among Y rows.
hipError_t launch_l2_normalize(const void* src, int src_is_f32, f16* dst, int64_t rows, int d,
                               hipStream_t stream);
size_t xsim_workspace_bytes(int64_t nx_pad, int64_t ny_pad, int k, int d);
hipError_t launch_xsim_topk(const f16* Xn, int64_t nx, int64_t nx_pad, const f16* Yn, int64_t ny,
                            int64_t ny_pad, int d, int k, int64_t y_index_offset, int32_t* idx,
                            float* score, void* workspace, hipStream_t stream);
hipError_t launch_topk_merge(const float* part_scores, const int32_t* part_idx, int parts, int64_t n, int k,
                             float* out_scores, int32_t* out_idx, hipStream_t stream);
hipError_t launch_margin_select(const float* fwd_scores, const int32_t* fwd_idx, int64_t nx, int k,
                                const float* bwd_scores, int64_t ny, int kind, int64_t x_off, int32_t* pred,
                                float* pred_margin, int32_t* err_count, hipStream_t stream);

// ---- decoder / beam search (decoder.hip) ----
hipError_t launch_dec_embed(const int32_t* tok, const f16* table, const float* pe_row, float scale,
                            float* x, int rows, int d, int64_t vocab, hipStream_t stream);
// x[r] += sum_z parts[z][r] (+ c[r / group] if c); h[r] = LN(x[r])   (parts/c may be null)
// pf / pf_bytes: weights of a later GEMM that the CUs the row work leaves idle read ahead (common.hpp: prefetch_range)
// parts_f16: the slabs are fp16 (part_stride counts ELEMENTS either way)
hipError_t launch_sum_layernorm(void* x, const void* parts, int nparts, size_t part_stride,
                                const float* c, int group, const float* w, const float* b, float eps,
                                f16* h, int rows, int d, hipStream_t stream, int h_tm = 0, int x_f16 = 0,
                                const void* pf = nullptr, size_t pf_bytes = 0, int parts_f16 = 0);
hipError_t launch_dec_attention(const f16* kv, const int32_t* anc, int anc_stride, f16* ctx, int rows,
                                int rows_pad, int d, int heads, int pos, hipStream_t stream);
constexpr int kVocabScanK2Max = 16;
// Per row: softmax normaliser (pmax, psum) from the GEMM's tile statistics and the top-k2 candidates
// among the k2 best tiles + tile 0 (pval / pidx [rows][kVocabScanK2Max]), without re-reading the whole
// logits row.
// f16_tm: the logits are fp16 in the tile-major layout (K = ldl) instead of fp32 [rows][ldl]
hipError_t launch_vocab_select(const float* logits, int ldl, int f16_tm, int rows, int vocab, const float* tile_max,
                               const float* tile_sum, int ntiles, int stat_rows, int k2, float inv_temp, int pad_idx, int eos_idx,
                               int unk_idx, float unk_penalty, int block_eos, float* pmax, float* psum, float* pval,
                               int* pidx, hipStream_t stream);
struct BeamStepArgs {
  int32_t* tok; float* cum; int32_t* nactive; int32_t* done; int32_t* ndone;
  int32_t* parent; int32_t* new_tok; float* new_cum;
  const int32_t* hist; int32_t* fin_tok; int32_t* fin_len; float* fin_score; int32_t* fin_count;
  float* margins;  // [n][2] or null
  const float* logits; int ldl;
  int logits_f16_tm;  // logits are fp16 in the tile-major layout (K = ldl) instead of fp32 [rows][ldl]
  const float* pmax; const float* psum; const float* pval; const int* pidx; int nchunks;
  int n, beam, k2, pos, prompt_len, forced_tok, max_len;
  float inv_temp, len_penalty; int normalize, eos_idx, hist_stride;
};
hipError_t launch_beam_step(const BeamStepArgs& a, hipStream_t stream);
hipError_t launch_beam_reorder(const int32_t* parent, const int32_t* new_tok, const float* new_cum,
                               const int32_t* anc, int32_t* anc2, const int32_t* hist, int32_t* hist2,
                               int32_t* tok, float* cum, int rows, int stride, int pos,
                               hipStream_t stream);
hipError_t launch_beam_init(int32_t* tok, float* cum, int32_t* nactive, int32_t* done, int32_t* ndone,
                            int32_t* fin_count, int32_t* hist, int32_t* anc, float* margins, int rows, int n,
                            int stride, int first_tok, hipStream_t stream);
hipError_t launch_beam_output(const int32_t* fin_tok, const int32_t* fin_len, const float* fin_score,
                              const int32_t* fin_count, int n, int beam, int stride, int out_stride,
                              int32_t* out_tok, int32_t* out_len, float* out_score, float* margins,
                              hipStream_t stream);
hipError_t launch_gather_tokens(const int64_t* src, int src_stride, int col, int32_t* tok, int rows,
                                hipStream_t stream);

// ---- embedding heads: BLASER / MuTox MLPs (heads.hip) ----
// form 0: out = f16(x) (src only); 1 QE: [src, mt, src*mt, |mt-src|]; 2 COMET:
// [ref, mt, src*mt, ref*mt, |mt-src|, |mt-ref|]; inputs optionally L2-normalised first (F.normalize).
// out: [rows rounded up to 128][blocks * d] fp16, padding rows zero.
hipError_t launch_head_featurize(int form, const void* src, const void* mt, const void* ref, int in_is_f32,
                                 int rows, int d, int norm, f16* out, hipStream_t stream);
// out[r][o] = act(h[r,:] . w[o,:] + b[o]) for o < out_dim (<= 8); act 0 none, 1 tanh, 2 sigmoid
hipError_t launch_head_output(const f16* h, int ldh, const float* w, const float* b, int rows, int K, int out_dim,
                              int act, float* out, hipStream_t stream);

// ---- speech path (speech.hip) ----
hipError_t launch_fbank(const float* wave, int64_t nsamples, float scale, int standardize, const float* window,
                        const float* mel_w, const int* mel_range, float* out, hipStream_t stream);
hipError_t launch_stack_ln(const float* fb, int n, int t, int nb, const int32_t* cu, int max_len, const float* w,
                           const float* b, float eps, f16* out, int ldo, hipStream_t stream);
// x = LN1(x) in place; h = f16(w2 ? LN2(x) : x) (h may be null)
hipError_t launch_ln2(void* x, const float* w1, const float* b1, const float* w2, const float* b2, float eps,
                      f16* h, int rows, int d, hipStream_t stream, int out_tm = 0, int x_f16 = 0, int x_tm = 0);
hipError_t launch_fbank_batch(const float* waves, const int64_t* off_dev, int n, int tpad, float scale,
                              int standardize, const float* window, const float* mel_w, const int* mel_range,
                              float* out, hipStream_t stream);
hipError_t launch_relpos_attention(const f16* qkv, const int32_t* cu, const f16* rp, int rp_zero, int rp_rows,
                                   const float* u_bias, const float* v_bias, f16* ctx, int n, int max_len, int d,
                                   int heads, hipStream_t stream, int ctx_tm = 0, int qkv_tm = 0);
bool relpos_attention_reads_tile_major();  // tuning: the LDS-ring kernel is on and SPEECH_QKV_TM is not 0
hipError_t launch_dwconv_bn_silu(const f16* x, const int32_t* cu, const float* w, const float* scale,
                                 const float* shift, f16* y, int n, int max_len, int d, int ktaps,
                                 hipStream_t stream, int y_tm = 0, int x_tm = 0);
hipError_t launch_pool_attention(const f16* q, const f16* kv, const int32_t* cu, f16* ctx, int n, int d, int heads,
                                 hipStream_t stream);
hipError_t launch_broadcast_row(const float* row, float* x, int rows, int d, hipStream_t stream);


// ---- sampling generation (sampling.hip)
struct SampleRowsArgs {
  const float* logits;
  int64_t ld;
  int rows, vocab;
  float inv_temp;
  int pad_idx, eos_idx, block_eos;
  int unk_idx;        // with unk_penalty != 0: probs[unk] -= unk_penalty before the filter (never below 0: masked then)
  float unk_penalty;
  int forced_tok;  // >= 0: no draw, the token is given (prompt forcing, forced EOS)
  int mode, top_k;
  float top_p;
  const unsigned long long* z;          // [rows] random words, or null: hash(seed, row, step)
  unsigned long long seed;
  int step;
  const int32_t* done;   // [rows] rows to skip, may be null
  int32_t* out_tok;
  float* out_logp;
  unsigned long long* out_kept_mass;    // optional
  int32_t* out_kept_count;
};

struct SampleUpdateArgs {
  const int32_t* samp_tok;
  const float* samp_logp;
  int32_t* tok;   // next step's input token
  float* cum;
  int32_t* done;
  int32_t* ndone;
  int32_t* out_tokens;
  int32_t* out_lens;
  float* out_scores;
  int n, out_stride, pos, prompt_len, eos_idx, normalize;
  float len_penalty;
};
hipError_t launch_sample_rows(const SampleRowsArgs& a, hipStream_t stream);
hipError_t launch_sample_update(const SampleUpdateArgs& a, hipStream_t stream);

// ---- generic-dimension kernels (flex.hip): fp32 activations and weights, any model_dim / head_dim <= 256 ----
// Y[M,N] = act(X[M,K] . W[N,K]^T + bias) (+ R); act 0 none, 1 relu; R may alias Y
hipError_t launch_flex_linear(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M,
                              int N, int K, int act, const float* R, int ldr, hipStream_t stream);
hipError_t launch_flex_layernorm(const float* x, const float* w, const float* b, float eps, float* y, int rows, int d,
                                 hipStream_t stream);
// x[r] = table[id[r]] * scale (+ pe[pos + pos_offset]); ids64 ([n, s] batch, pos = r % s, fixed_pos = -1) or ids32
// (one token per row at position fixed_pos); out-of-range ids raise *bad
hipError_t launch_flex_embed(const int64_t* ids64, const int32_t* ids32, const float* table, const float* pe,
                             float scale, float* x, int rows, int d, int s, int pos_offset, int fixed_pos,
                             int64_t vocab, int32_t* bad, hipStream_t stream);
// out[b, i] = softmax(q[b, i] . k[b, :klens[b]]^T / sqrt(hd)) . v per head; q rows b * sq + i, k / v rows b * sk + j
hipError_t launch_flex_attention(const float* q, int ldq, const float* k, const float* v, int ldk, float* out, int ldo,
                                 int items, int sq, int sk, const int32_t* klens, int heads, int hd, int causal,
                                 hipStream_t stream);
hipError_t launch_flex_dec_attention(const float* kv, const int32_t* anc, int anc_stride, float* ctx, int rows,
                                     int rows_pad, int d, int heads, int pos, hipStream_t stream);
hipError_t launch_flex_pool(const float* x, const int32_t* lens, int pooling, float* out, int n, int s, int d,
                            hipStream_t stream);
hipError_t launch_flex_tile_stats(const float* logits, int ld, int rows, int vocab, float scale, float* tile_max,
                                  float* tile_sum, int stat_rows, hipStream_t stream);
hipError_t launch_flex_add_rows(float* x, const float* c, int rows, int d, int group, hipStream_t stream);
hipError_t launch_flex_store_encoded(const float* x, const int32_t* lens, int n, int s, int d, void* out, bool out_f16,
                                     hipStream_t stream);

}  // namespace smi
