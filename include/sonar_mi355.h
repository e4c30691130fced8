/*
 * sonar_mi355.h -- C ABI of the MI355X-native SONAR inference hot path.
 *
 * The reference (facebookresearch/SONAR v0.4.0) has no native code and no FFI:
 * its hot path is `model(batch)` inside the Python pipelines.  The entry points
 * below are what a binding for that path attaches to; each one cites the
 * reference interface it stands in for (paths relative to the reference repo).
 * INTEGRATION.md shows the ctypes stub a SONAR maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types cross this boundary;
 *  - every call returns SMI_OK (0) or a negative smi_status; smi_last_error()
 *    returns a thread-local human readable message for the last failure;
 *  - device buffers passed in are borrowed for the duration of the call (it is
 *    stream-ordered: the call enqueues work on `stream` and returns);
 *  - the engine owns its packed weights and workspace; the caller may free its
 *    own weight buffers as soon as *_create returns;
 *  - a handle is not re-entrant: use one handle per device and per thread;
 *  - nothing here falls back to the CPU: without a HIP device every compute
 *    entry point fails with SMI_ERR_NO_DEVICE.
 */
#ifndef SONAR_MI355_H
#define SONAR_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum smi_status {
  SMI_OK = 0,
  SMI_ERR_INVALID_ARG = -1,
  SMI_ERR_UNSUPPORTED = -2, /* shape/config outside what the kernels cover */
  SMI_ERR_NO_DEVICE = -3,
  SMI_ERR_OOM = -4,
  SMI_ERR_HIP = -5 /* a HIP runtime call failed; see smi_last_error() */
} smi_status;

/* SMI_BF16 exists at the BOUNDARY only (smi_cast): the engines take and return fp32 / fp16 tensors and compute with
 * fp16 operands and fp32 accumulation whatever a model's nominal dtype is.  A bf16 model (the reference's pipelines
 * accept any dtype, sonar/inference_pipelines/text.py:36-54,161-162) is served by casting: its weights are exactly
 * representable in fp16 over the normal range, its outputs are rounded to bf16 once, on the way out. */
typedef enum smi_dtype { SMI_F32 = 0, SMI_F16 = 1, SMI_BF16 = 2 } smi_dtype;
/* SMI_POOL_ATTENTION: the trainable pooler of sonar/nn/encoder_pooler.py:49-95 (factory.py:155-226) */
typedef enum smi_pooling { SMI_POOL_MEAN = 0, SMI_POOL_MAX = 1, SMI_POOL_LAST = 2, SMI_POOL_ATTENTION = 3 } smi_pooling;

/* A dense tensor handed to the engine at create time.  `data` may live in host
 * or device memory (`on_device`); fp32 or fp16. */
typedef struct smi_tensor {
  const void* data;
  int32_t dtype;     /* smi_dtype */
  int32_t on_device; /* 0 host, 1 device (current HIP device) */
  int64_t numel;
} smi_tensor;

/* Mirrors SonarTextEncoderConfig (sonar/models/sonar_text/config.py:14-85) for
 * the fields that affect the forward pass of the `basic`/`small` archs. */
typedef struct smi_text_encoder_config {
  int32_t model_dim;     /* 1024 (see "Shapes" below) */
  int32_t num_layers;    /* 24 */
  int32_t num_heads;     /* 16 */
  int32_t ffn_inner_dim; /* 8192; multiple of 128 */
  int64_t vocab_size;    /* 256206 */
  int32_t max_seq_len;   /* 514 = 512 + pad_idx + 1 (factory.py:56-59) */
  int32_t pos_offset;    /* 2 = pad_idx + 1 (_legacy_pad_idx, factory.py:88-92) */
  float embed_scale;     /* sqrt(model_dim), or 1 if no_scale_embedding */
  float ln_eps;          /* 1e-5 */
  int32_t pooling;       /* smi_pooling (config.py: pooling="mean") */
  int32_t flags;         /* SMI_ENC_* */
  /* Attention pooling (all 0 for the released models): width of the sentence vector (config.py `embedding_dim`,
   * 0 = model_dim) and the shape of the pooler's decoder layers (num_decoder_layers, num_decoder_attn_heads,
   * decoder_ffn_inner_dim or ffn_inner_dim; factory.py:190-226). */
  int32_t embedding_dim;
  int32_t pooler_layers;
  int32_t pooler_heads;
  int32_t pooler_ffn_dim;
} smi_text_encoder_config;
/* Shapes: the MFMA engines serve model_dim = num_heads * 64 in {256, 512, 768, 1024, 2048}, ffn_inner_dim % 128 == 0,
 * static pooling -- the released models.  EVERY OTHER shape the reference's factory accepts (any model_dim divisible by
 * num_heads with head_dim <= 256, attention pooling, the flags below other than FP16_RESIDUAL) runs on the library's
 * generic-dimension fp32 kernels (csrc/flex.hip), chosen inside smi_text_encoder_create. */

/* smi_text_encoder_config.flags */
/* Keep the residual stream in fp16 instead of fp32.  The reference's fp16 model does exactly this
 * (every tensor of the model is fp16, sonar/inference_pipelines/text.py:161-162 `.to(device, dtype)`);
 * the engine's default fp32 stream costs 2x the residual traffic and buys ~100x margin on the 1e-3
 * parity bound.  With the flag each residual add is one fp32 add rounded once to fp16. */
#define SMI_ENC_FP16_RESIDUAL 1
/* config.py `normalize_before`: the encoder stack ends in its own LayerNorm (StandardTransformerEncoder norm_order PRE,
 * factory.py:107-109, weights encoder_layer_norm_*) and the pooler's layers are pre-norm with a final LayerNorm. */
#define SMI_ENC_NORMALIZE_BEFORE 2
/* config.py `layernorm_embedding`: LayerNorm on the frontend output (weights embed_layer_norm_*). */
#define SMI_ENC_LAYERNORM_EMBEDDING 4
/* config.py `no_token_positional_embeddings`: pos_table is absent (data NULL).  (`learned_pos` needs no flag: the caller
 * passes encoder_frontend.pos_encoder.weight as pos_table with pos_offset 0.) */
#define SMI_ENC_NO_POSITIONS 8

/* Per-layer parameters, names as produced by the reference's checkpoint
 * conversion (sonar/models/sonar_text/handler.py:71-82).  Linear weights are
 * [out, in] row-major as in torch.nn.Linear. */
typedef struct smi_text_encoder_layer {
  smi_tensor self_attn_layer_norm_w, self_attn_layer_norm_b;
  smi_tensor q_w, q_b, k_w, k_b, v_w, v_b, out_w, out_b;
  smi_tensor ffn_layer_norm_w, ffn_layer_norm_b;
  smi_tensor ffn_inner_w, ffn_inner_b, ffn_out_w, ffn_out_b;
} smi_text_encoder_layer;

/* One decoder layer of the attention pooler (factory.py:199-218).  Its self-attention sees ONE token, so only the
 * value and output projections matter (softmax over one key is 1); the cross-attention reads the encoder output:
 * its k / v projections are [embedding_dim, model_dim]. */
typedef struct smi_text_pooler_layer {
  smi_tensor self_attn_layer_norm_w, self_attn_layer_norm_b;
  smi_tensor self_v_w, self_v_b, self_out_w, self_out_b;
  smi_tensor cross_layer_norm_w, cross_layer_norm_b;
  smi_tensor cross_q_w, cross_q_b, cross_k_w, cross_k_b, cross_v_w, cross_v_b, cross_out_w, cross_out_b;
  smi_tensor ffn_layer_norm_w, ffn_layer_norm_b;
  smi_tensor ffn_inner_w, ffn_inner_b, ffn_out_w, ffn_out_b;
} smi_text_pooler_layer;

typedef struct smi_text_encoder_weights {
  smi_tensor embed;     /* encoder_frontend.embed.weight [vocab, model_dim] */
  smi_tensor pos_table; /* position table [max_seq_len + pos_offset, model_dim] fp32 (sinusoidal or learned); data NULL =
                         * no_token_positional_embeddings */
  smi_tensor final_layer_norm_w, final_layer_norm_b; /* model-level layer_norm (factory.py:117) */
  const smi_text_encoder_layer* layers;              /* num_layers entries */
  /* the rest is read only when the configuration asks for it */
  smi_tensor encoder_layer_norm_w, encoder_layer_norm_b; /* encoder.layer_norm (SMI_ENC_NORMALIZE_BEFORE) */
  smi_tensor embed_layer_norm_w, embed_layer_norm_b;     /* encoder_frontend.layer_norm (SMI_ENC_LAYERNORM_EMBEDDING) */
  smi_tensor pooler_query;  /* [embedding_dim] fp32: pooler.decoder_frontend.embed.weight[bos] * sqrt(embedding_dim) +
                             * PE[0] -- the pooler's one input token after its frontend (encoder_pooler.py:78-92) */
  const smi_text_pooler_layer* pooler;                        /* pooler_layers entries */
  smi_tensor pooler_layer_norm_w, pooler_layer_norm_b;   /* pooler.decoder.layer_norm (SMI_ENC_NORMALIZE_BEFORE) */
  smi_tensor pooler_proj_w, pooler_proj_b;               /* pooler.projection_out [embedding_dim, embedding_dim] */
} smi_text_encoder_weights;

typedef struct smi_text_encoder smi_text_encoder; /* opaque */

/* Library / device ------------------------------------------------------- */
const char* smi_version(void);
/* ABI revision of this header: bumped whenever a struct grows, an argument list changes or a workspace formula
 * changes (round 2 = 2, round 3 = 3, ...).  A binding compares it with SMI_ABI_VERSION at load time and refuses a
 * library built from another revision (the structs carry no size field). */
#define SMI_ABI_VERSION 6
int smi_abi_version(void);
const char* smi_last_error(void);
/* Tuning registry (round 5).  Every A/B switch of the library -- engine-family thresholds, split-K part counts, storage
 * types, layout choices; the list with meanings and defaults is sonar_amd/csrc/tuning.hpp, enumerated here by
 * smi_tuning_name(0..) until it returns NULL -- is one process-wide integer that is set through these calls and read with
 * an atomic load.  THE LIBRARY NEVER READS THE ENVIRONMENT: a host that calls none of them runs the shipped defaults, and
 * results are a deterministic function of (inputs, batch row count, the switches set here) -- see INTEGRATION.md
 * "Numerics contract".  `name` with or without the "SMI_" prefix.  (The reference has no counterpart: its kernels are
 * ATen's, selected by torch's own heuristics.) */
int smi_tuning_set(const char* name, int32_t value);
int smi_tuning_unset(const char* name);
int smi_tuning_get(const char* name, int32_t* value, int32_t* is_set);
const char* smi_tuning_name(int32_t index);
/* Selects the HIP device for this thread (hipSetDevice). */
int smi_init(int device_id);
int smi_device_count(void);

/* Text encoder ------------------------------------------------------------
 * Stands in for: SonarTextEncoderFactory.create_model + checkpoint load
 * (sonar/models/sonar_text/factory.py:72-120, handler.py:52-94) and
 * SonarTextTransformerEncoderModel.forward (sonar/models/sonar_text/model.py:130-143)
 * as invoked by `.map(self.model)` in
 * TextToEmbeddingModelPipeline.predict (sonar/inference_pipelines/text.py:244). */
int smi_text_encoder_create(const smi_text_encoder_config* cfg, const smi_text_encoder_weights* w,
                            int64_t max_tokens_hint, smi_text_encoder** out);
void smi_text_encoder_destroy(smi_text_encoder* enc);

/* ids:      device int64 [n, s] right-padded token ids (SequenceBatch.seqs)
 * seq_lens: HOST int32 [n] valid lengths, or NULL when the batch is not ragged
 *           (PaddingMask is None, sonar/inference_pipelines/utils.py:18-21)
 * out_emb:  device [n, embedding_dim or model_dim] sentence_embeddings, out_dtype
 * out_encoded: optional device [n, s, model_dim] encoded_seqs (out_dtype), pads zeroed; may be NULL
 * stream:   hipStream_t (NULL = default stream) */
int smi_text_encoder_forward(smi_text_encoder* enc, const int64_t* ids, const int32_t* seq_lens,
                             int32_t n, int32_t s, void* out_emb, void* out_encoded,
                             int32_t out_dtype, void* stream);

/* Synchronises `stream` and reports what the device found while running the forward calls enqueued on
 * it: SMI_ERR_INVALID_ARG if a batch held token ids outside [0, vocab_size) -- the reference's
 * embedding lookup raises an IndexError there (tokenizer / model vocabulary mismatch); the engine never
 * reads outside the table and flags the batch instead.  The flag is reported (and cleared) ONLY here, after
 * the synchronisation -- smi_text_encoder_forward never refuses a later, valid batch because of it.
 * predict() calls this before it returns embeddings. */
int smi_text_encoder_status(smi_text_encoder* enc, void* stream);

/* Bytes of device memory currently held by the handle (weights + workspace). */
int64_t smi_text_encoder_device_bytes(const smi_text_encoder* enc);

/* Per-kernel timing with HIP events recorded on the caller's stream around every
 * launch of smi_text_encoder_forward (measurement aid for the roofline report; the
 * reference has no profiler).  Off by default. */
typedef enum smi_prof_slot {
  SMI_PROF_EMBED = 0,
  SMI_PROF_LAYERNORM = 1,
  SMI_PROF_GEMM_QKV = 2,
  SMI_PROF_ATTENTION = 3,
  SMI_PROF_GEMM_OUT = 4,
  SMI_PROF_GEMM_FFN1 = 5,
  SMI_PROF_GEMM_FFN2 = 6,
  SMI_PROF_LN_POOL = 7,
  SMI_PROF_SLOTS = 8
} smi_prof_slot;
int smi_text_encoder_set_profiling(smi_text_encoder* enc, int32_t enable);
/* Synchronises the recorded events, ADDS elapsed milliseconds and launch counts per
 * slot into ms[SMI_PROF_SLOTS] / launches[SMI_PROF_SLOTS], then clears the record. */
int smi_text_encoder_read_profile(smi_text_encoder* enc, double* ms, int64_t* launches);

/* Text decoder + beam search ---------------------------------------------------
 * Stands in for: SonarTextDecoderFactory.create_model + checkpoint load
 * (sonar/models/sonar_text/factory.py:229-315, handler.py:122-172), the
 * SonarEncoderDecoderModel.encode/decode/project calls of one generation step
 * (sonar/models/sonar_translation/model.py:48-78, DummyEncoderModel :81-95) and fairseq2's
 * BeamSearchSeq2SeqGenerator as driven by EmbeddingToTextModelPipeline.predict
 * (sonar/inference_pipelines/text.py:305-346). */
typedef struct smi_text_decoder_config {
  int32_t model_dim;     /* 1024; num_heads * 64 in {256, 512, 768, 1024, 2048} runs on the MFMA engines, any other
                          * multiple of num_heads with head_dim <= 256 (e.g. the `toy` arch) on the generic fp32 kernels */
  int32_t num_layers;    /* 24 */
  int32_t num_heads;     /* 16 */
  int32_t ffn_inner_dim; /* 8192 */
  int64_t vocab_size;    /* 256206 */
  int32_t max_seq_len;   /* 512: longest target sequence incl. prompt (config.py:197-219) */
  int32_t pos_offset;    /* 2 = model pad_idx + 1 (_legacy_pad_idx, factory.py:248-252) */
  int32_t input_dim;     /* conditioning vector dimension (config.py input_dim; model_dim when unset) */
  float embed_scale;     /* sqrt(model_dim) */
  float ln_eps;          /* 1e-5 */
  int32_t pad_idx, unk_idx, bos_idx, eos_idx; /* TOKENIZER ids: 0, 1, 2, 3 */
} smi_text_decoder_config;

/* encoder_decoder_attn q/k projections and its LayerNorm are not needed: the encoder output is
 * ONE vector, so the attention weight is 1 and the block reduces to W_o (W_v e + b_v) + b_o. */
typedef struct smi_text_decoder_layer {
  smi_tensor self_attn_layer_norm_w, self_attn_layer_norm_b;
  smi_tensor q_w, q_b, k_w, k_b, v_w, v_b, out_w, out_b;
  smi_tensor cross_v_w, cross_v_b, cross_out_w, cross_out_b;
  smi_tensor ffn_layer_norm_w, ffn_layer_norm_b;
  smi_tensor ffn_inner_w, ffn_inner_b, ffn_out_w, ffn_out_b;
} smi_text_decoder_layer;

typedef struct smi_text_decoder_weights {
  smi_tensor embed;     /* decoder_frontend.embed.weight [vocab, model_dim]; also the tied final_proj */
  smi_tensor pos_table; /* sinusoidal table [max_seq_len + pos_offset, model_dim] fp32 */
  smi_tensor final_layer_norm_w, final_layer_norm_b; /* decoder.layer_norm */
  const smi_text_decoder_layer* layers;
} smi_text_decoder_weights;

typedef struct smi_beam_search_params {
  int32_t beam_size;        /* 1..8 (fairseq2 default 5) */
  int32_t max_seq_len;      /* min(prompt_len + max_gen_len, model max): EOS is forced at max_seq_len-1 */
  int32_t min_seq_len;      /* prompt_len + min_gen_len: EOS blocked while step < min_seq_len */
  int32_t normalize_scores; /* 1: score / (len - 1)^len_penalty */
  float len_penalty;        /* 1.0 */
  float unk_penalty;        /* 0.0 */
  float temperature;        /* 1.0 */
  int32_t reserved;
} smi_beam_search_params;

typedef struct smi_text_decoder smi_text_decoder; /* opaque */

int smi_text_decoder_create(const smi_text_decoder_config* cfg, const smi_text_decoder_weights* w,
                            smi_text_decoder** out);
void smi_text_decoder_destroy(smi_text_decoder* dec);

/* Teacher-forced logits (the reference test's call, tests/integration_tests/test_text_sonar.py:61-105):
 * emb device [n, model_dim] (emb_dtype), prev_tokens device int64 [n, t], out_logits device fp32 [n, t, vocab]. */
int smi_text_decoder_logits(smi_text_decoder* dec, const void* emb, int32_t emb_dtype, int32_t n,
                            const int64_t* prev_tokens, int32_t t, float* out_logits, void* stream);

/* Beam search for n sentence embeddings.  prompt: HOST int64 [prompt_len] (= [</s>, __lang__]).
 * Outputs (device): out_tokens int32 [n, beam, max_seq_len] generated tokens after the prompt incl. the
 * final EOS, -1 padded; out_lens int32 [n, beam]; out_scores fp32 [n, beam]; hypotheses best first
 * (the reference decodes hypotheses[0]).  Synchronises `stream` every 8 steps to test for completion. */
int smi_text_decoder_generate(smi_text_decoder* dec, const void* emb, int32_t emb_dtype, int32_t n,
                              const int64_t* prompt, int32_t prompt_len, const smi_beam_search_params* params,
                              int32_t* out_tokens, int32_t* out_lens, float* out_scores, void* stream);

/* Decision margins of the LAST smi_text_decoder_generate call on this handle: out_margins device fp32
 * [n, 2].  [s][0] = the smallest gap, over all free decoding steps of sentence s, between neighbouring
 * entries of the sorted beam x vocab candidate list among the candidates the beam rules consumed plus
 * the first one they did not (log-prob units; for beam_size 1 this is the greedy top-1 / top-2 margin);
 * [s][1] = normalised score of the returned hypothesis minus the runner-up's (+inf if there is none).
 * "Exact token-id match for greedy decode" (BASELINE north_star) is tested as: every token equal to the
 * fp32 CPU oracle's unless [s][0] is below the epsilon stated in the test. */
int smi_text_decoder_last_margins(smi_text_decoder* dec, float* out_margins, int32_t n, void* stream);

/* Independent decode chains of smi_text_decoder_generate (round 4).  Sentences do not interact in
 * EmbeddingToTextModelPipeline.predict (sonar/inference_pipelines/text.py:329-346 decodes buckets of sentences), so a
 * large batch may run as `chains` sentence groups, each with its own workspace, KV cache, beam state, stream and host
 * thread: one group's per-launch fixed costs fall under the other's K loops.  0 = the engine's choice (tuning
 * switch DEC_CHAINS, else the built-in default), 1 = one chain, up to 4.  Hypotheses equal the single chain's up to the
 * fp32 summation order of split-K slabs. */
int smi_text_decoder_set_chains(smi_text_decoder* dec, int32_t chains);

/* Storage type of the logits inside smi_text_decoder_generate (round 4): SMI_F32 (default) or SMI_F16.  The reference's fp16
 * model produces fp16 logits (the tied final_proj is an fp16 Linear; fairseq2's beam search up-casts them inside
 * log_softmax, sonar/inference_pipelines/text.py:305-346 -> BeamSearchSeq2SeqGenerator), so SMI_F16 is what an fp16 model's
 * pipeline selects: the logits GEMM then rounds its fp32 accumulators to fp16 once, takes the softmax statistics of the
 * ROUNDED values and writes half the bytes (0.66 instead of 1.31 GB per position at 256 sentences x beam 5).
 * smi_text_decoder_logits and smi_text_decoder_sample keep fp32 logits and fp32 partial sums. */
int smi_text_decoder_set_beam_logits_dtype(smi_text_decoder* dec, int32_t dtype);

/* Storage type of the split-K PARTIAL sums of the attention-output and FFN-output projections inside
 * smi_text_decoder_generate: SMI_F32 (default) or SMI_F16 (round 4; its own setting since round 5, it used to follow the
 * logits dtype).  With SMI_F16 each partial is rounded to fp16 once -- SATURATING at +-65504 (MODE.FP16_OVFL), so a partial
 * outside fp16's range cannot turn a representable sum into inf --, the consumer widens, sums in fp32 in slab order and adds to
 * the fp32 residual stream.  The reference's fp16 model rounds the FULL sublayer output to fp16 once
 * (sonar/inference_pipelines/text.py:36-54 puts the whole model in fp16); rounding 2-8 partials instead bounds the error by
 * 2^-11 x the sum of the partials' magnitudes rather than of the result's -- the same order unless the K ranges cancel
 * (tests/test_gpu_kernels.py::test_splitk_f16_slabs_cancellation_and_saturation).  It halves the 42 MB per layer the FFN
 * output projection writes and the next kernel reads at 1 280 rows (-3.9 % of a C5 step).  An fp16 model's Python engine
 * (TextDecoderEngine(dtype=float16)) selects it; a C caller gets fp32 partial sums unless it asks. */
int smi_text_decoder_set_slab_dtype(smi_text_decoder* dec, int32_t dtype);

/* Sampling generation (sonar/inference_pipelines/text.py:315-320: a `sampler` makes predict() build
 * fairseq2's SamplingSeq2SeqGenerator instead of the beam search; one hypothesis per sentence).
 * Per step: probs = softmax(logits / temperature) in fp32, pad -> 0, EOS -> 0 before min_seq_len,
 * probs[unk] -= unk_penalty (a result <= 0 removes the token), EOS forced at max_seq_len - 1; TopKSampler keeps the k most probable tokens, TopPSampler the sorted
 * prefix whose exclusive cumulative probability stays <= p; one token is drawn from the renormalised
 * kept set; the step score is log(probs[token]).  The draw is a counter-based hash of
 * (seed, sentence, step): a call is reproducible, and independent of the batch it runs in. */
#define SMI_SAMPLER_TOP_K 0
#define SMI_SAMPLER_TOP_P 1
typedef struct smi_sampling_params {
  int32_t sampler;          /* SMI_SAMPLER_TOP_K / SMI_SAMPLER_TOP_P */
  int32_t top_k;            /* TopKSampler(k), k >= 1 */
  float top_p;              /* TopPSampler(p), 0 < p <= 1 */
  float temperature;        /* 1.0 */
  int32_t max_seq_len;      /* as smi_beam_search_params */
  int32_t min_seq_len;
  int32_t normalize_scores; /* 1: score / (len - 1)^len_penalty */
  float len_penalty;        /* 1.0 */
  uint64_t seed;
  float unk_penalty;        /* 0.0; subtracted from the PROBABILITY of the UNK token (fairseq2's sampling generator) */
} smi_sampling_params;

/* Outputs (device): out_tokens int32 [n, max_seq_len] generated tokens after the prompt incl. the final
 * EOS, -1 padded; out_lens int32 [n]; out_scores fp32 [n].  prompt: HOST int64 [prompt_len]. */
int smi_text_decoder_sample(smi_text_decoder* dec, const void* emb, int32_t emb_dtype, int32_t n,
                            const int64_t* prompt, int32_t prompt_len, const smi_sampling_params* params,
                            int32_t* out_tokens, int32_t* out_lens, float* out_scores, void* stream);

/* The filter + draw of one sampling step on given logits (device fp32 [rows, ld], ld % 4 == 0,
 * ld >= vocab rounded up to 4, vocab <= 2^18), exposed for the parity tests: z device uint64 [rows]
 * random words (the draw is floor(z * kept_mass / 2^64) into the kept mass); outputs device:
 * out_token int32 [rows], out_logprob fp32 [rows], and optionally the kept set's size and its mass in
 * Q40 fixed point relative to exp(max scaled logit). */
int smi_sample_rows(const float* logits, int64_t ld, int32_t rows, int32_t vocab, int32_t sampler, int32_t top_k,
                    float top_p, float temperature, int32_t pad_idx, int32_t eos_idx, int32_t block_eos,
                    int32_t unk_idx, float unk_penalty, const uint64_t* z, int32_t* out_token, float* out_logprob,
                    uint64_t* out_kept_mass, int32_t* out_kept_count, void* stream);

/* Speech encoder -------------------------------------------------------------------
 * Stands in for: WaveformToFbankConverter(num_mel_bins=80, waveform_scale=2**15,
 * standardize=True) (sonar/inference_pipelines/speech.py:283-290), SonarSpeechEncoderFactory +
 * checkpoint load (sonar/models/sonar_speech/factory.py:53-152, handler.py:46-110) and
 * SonarSpeechEncoderModel.forward (sonar/models/sonar_speech/model.py:59-77) as invoked by
 * SpeechToEmbeddingModelPipeline.predict (speech.py:431-474). */
typedef struct smi_speech_encoder_config {
  int32_t model_dim;        /* 1024 = num_heads*64 */
  int32_t num_layers;       /* 24 conformer blocks */
  int32_t num_heads;        /* 16 */
  int32_t ffn_inner_dim;    /* 4096 */
  int32_t conv_kernel;      /* 31 (7 also built, for tests) */
  int32_t num_mel_bins;     /* 80; two frames are stacked -> feature_dim 160 */
  int32_t pooler_layers;    /* 3 ("english") / 6 ("non_english") */
  int32_t pooler_heads;     /* 16 */
  int32_t pooler_ffn_dim;   /* 4096 */
  int32_t pooler_vocab;     /* rows of the pooler embedding (= model_dim, factory.py:94-100) */
  int32_t bos_idx;          /* 2 */
  int32_t max_frames;       /* largest number of STACKED frames per clip (rel-pos table), e.g. 4096 */
  float ln_eps, bn_eps;     /* 1e-5, 1e-5 */
  int32_t flags;            /* SMI_ENC_FP16_RESIDUAL: fp16 residual stream, as the reference's `.half()` model */
  int32_t reserved;
} smi_speech_encoder_config;

typedef struct smi_conformer_layer {
  smi_tensor ffn1_layer_norm_w, ffn1_layer_norm_b, ffn1_inner_w, ffn1_inner_b, ffn1_out_w, ffn1_out_b;
  smi_tensor self_attn_layer_norm_w, self_attn_layer_norm_b;
  smi_tensor q_w, q_b, k_w, k_b, v_w, v_b, out_w, out_b;
  smi_tensor r_proj_w, u_bias, v_bias;             /* self_attn.sdpa.* */
  smi_tensor conv_layer_norm_w, conv_layer_norm_b;
  smi_tensor pointwise_conv1_w;                    /* [2d, d] (kernel size 1 squeezed), no bias */
  smi_tensor depthwise_conv_w;                     /* [d, k] */
  smi_tensor batch_norm_w, batch_norm_b, batch_norm_mean, batch_norm_var;
  smi_tensor pointwise_conv2_w;                    /* [d, d] */
  smi_tensor ffn2_layer_norm_w, ffn2_layer_norm_b, ffn2_inner_w, ffn2_inner_b, ffn2_out_w, ffn2_out_b;
  smi_tensor layer_norm_w, layer_norm_b;           /* final LayerNorm of the block */
} smi_conformer_layer;

/* POST-norm decoder layer of the attention pooler.  The self-attention runs over ONE token, so
 * its q/k projections cannot influence the output and are not needed. */
typedef struct smi_pooler_layer {
  smi_tensor self_v_w, self_v_b, self_out_w, self_out_b, self_attn_layer_norm_w, self_attn_layer_norm_b;
  smi_tensor cross_q_w, cross_q_b, cross_k_w, cross_k_b, cross_v_w, cross_v_b, cross_out_w, cross_out_b;
  smi_tensor cross_layer_norm_w, cross_layer_norm_b;
  smi_tensor ffn_inner_w, ffn_inner_b, ffn_out_w, ffn_out_b, ffn_layer_norm_w, ffn_layer_norm_b;
} smi_pooler_layer;

typedef struct smi_speech_encoder_weights {
  smi_tensor post_extract_layer_norm_w, post_extract_layer_norm_b; /* [2*num_mel_bins] */
  smi_tensor model_dim_proj_w, model_dim_proj_b;                   /* [d, 2*num_mel_bins], [d] */
  smi_tensor layer_norm_w, layer_norm_b;                           /* model-level LN (handler.py:102-108) */
  smi_tensor pooler_embed;                                         /* [pooler_vocab, d] */
  smi_tensor pooler_projection_out_w;                              /* [d, d], no bias */
  const smi_conformer_layer* layers;
  const smi_pooler_layer* pooler;
} smi_speech_encoder_weights;

typedef struct smi_speech_encoder smi_speech_encoder; /* opaque */

int smi_speech_encoder_create(const smi_speech_encoder_config* cfg, const smi_speech_encoder_weights* w,
                              smi_speech_encoder** out);
void smi_speech_encoder_destroy(smi_speech_encoder* enc);

/* fbank: device fp32 [n, t, num_mel_bins] zero-padded, t even (Collater pad_to_multiple=2,
 * speech.py:444); fbank_lens: HOST int32 [n] frames per clip or NULL; out_emb device [n, model_dim]. */
int smi_speech_encoder_forward(smi_speech_encoder* enc, const float* fbank, const int32_t* fbank_lens,
                               int32_t n, int32_t t, void* out_emb, int32_t out_dtype, void* stream);

/* Kaldi-compatible log-mel filterbank of ONE 16 kHz clip: wave device fp32 [nsamples] in [-1, 1],
 * out device fp32 [smi_fbank_num_frames(nsamples), 80].  25 ms / 10 ms frames, povey window,
 * pre-emphasis 0.97, DC removal, 512-point FFT, 80 mel bins from 20 Hz, log power, snip_edges. */
int64_t smi_fbank_num_frames(int64_t nsamples);
int smi_fbank(const float* wave, int64_t nsamples, float waveform_scale, int32_t standardize, float* out,
              void* stream);

/* The same filterbank for a whole batch in one launch: waves device fp32, the clips back to back;
 * offsets HOST int64 [n + 1] (clip i = samples offsets[i] .. offsets[i+1]); out device fp32
 * [n, tpad, 80], tpad >= every clip's frame count, rows past a clip's frames are set to 0 (the
 * reference's Collater pad value, speech.py:444).  Stands in for the per-file
 * WaveformToFbankConverter map + Collater of SpeechToEmbeddingModelPipeline.predict (speech.py:431-452). */
int smi_fbank_batch(const float* waves, const int64_t* offsets, int32_t n, float waveform_scale, int32_t standardize,
                    float* out, int64_t tpad, void* stream);

/* xsim mining ---------------------------------------------------------------
 * Stands in for the similarity search the reference performs as
 * F.normalize(x) @ F.normalize(y).T (tests/integration_tests/test_text_sonar.py:42-53)
 * and that xsim (README.md:5) evaluates: for each of the nx rows of X return the
 * k (<= 8) most cosine-similar rows of Y, best first.
 *
 * smi_xsim_normalize: dst = f16 row-normalised copy of src, padded with zero
 *   rows to a multiple of 256 rows (dst must hold smi_xsim_padded_rows(rows)*d f16).
 * smi_xsim_topk: Xn/Yn are such normalised, padded matrices.  idx [nx,k] int32
 *   (row index in Y plus y_index_offset; -1 if fewer than k candidates),
 *   score [nx,k] fp32.  workspace: smi_xsim_workspace_bytes() bytes of device memory (the per-chunk partial
 *   lists and, for k <= 4, tile-major copies of both matrices: the mining kernel streams those); workspace_bytes =
 *   what the caller allocated -- a buffer smaller than the formula of THIS library is refused, not overrun. */
int64_t smi_xsim_padded_rows(int64_t rows);
int smi_xsim_normalize(const void* src, int32_t src_dtype, int64_t rows, int32_t d, void* dst_f16,
                       void* stream);
int64_t smi_xsim_workspace_bytes(int64_t nx, int64_t ny, int32_t k, int32_t d);
int smi_xsim_topk(const void* xn_f16, int64_t nx, const void* yn_f16, int64_t ny, int32_t d,
                  int32_t k, int64_t y_index_offset, int32_t* idx, float* score, void* workspace,
                  int64_t workspace_bytes, void* stream);

/* k-way merge of `parts` per-shard top-k lists (device fp32 / int32 [parts, n, k], each sorted as
 * smi_xsim_topk returns them) into the k best of their union, same total order (score desc, index asc).
 * Folds the per-rank partial y-side neighbour lists of the sharded margin scoring; part_idx /
 * out_idx may be NULL when only the scores are needed (the margin uses the neighbour MEAN). */
int smi_xsim_merge_topk(const float* part_scores, const int32_t* part_idx, int32_t parts, int64_t n, int32_t k,
                        float* out_scores, int32_t* out_idx, void* stream);

/* Margin re-scoring of the k-NN candidates, LASER's xsim (facebookresearch/LASER source/xsim.py,
 * _score_margin / _score_knn; un-vendored, restated in oracle/xsim.py):
 *   score(i, j) = margin(cos(x_i, y_j), (mean_k cos(x_i, NN_k(x_i)) + mean_k cos(y_j, NN_k(y_j))) / 2)
 * fwd_scores / fwd_idx: device [nx, k] from smi_xsim_topk(X, Y, k) (indices into the rows of bwd_scores);
 * bwd_scores: device [ny, k] from smi_xsim_topk(Y, X, k) (or its cross-rank merge).  pred_idx[i] = the
 * candidate with the best score (first on ties), pred_margin[i] (nullable) its score; err_count (nullable,
 * device int32, ACCUMULATED) += #rows with pred_idx[i] != i + x_index_offset (aligned pairs). */
#define SMI_MARGIN_RATIO 0    /* a / b */
#define SMI_MARGIN_DISTANCE 1 /* a - b */
#define SMI_MARGIN_COSINE 2   /* a (bwd_scores unused) */
int smi_xsim_margin_select(const float* fwd_scores, const int32_t* fwd_idx, int64_t nx, int32_t k,
                           const float* bwd_scores, int64_t ny, int32_t margin, int64_t x_index_offset,
                           int32_t* pred_idx, float* pred_margin, int32_t* err_count, void* stream);

/* Embedding heads: BLASER / MuTox ---------------------------------------------
 * A small MLP over (features of) sentence embeddings.  Replaces
 *   BlaserModel.forward = F.normalize -> featurize_input -> mlp   sonar/models/blaser/model.py:82-125
 *   MutoxClassifier.forward = model_all (+ sigmoid)               sonar/models/mutox/model.py:18-24,
 *                                                                 sonar/models/mutox/factory.py:15-38
 * Hidden layers need in % 64 == 0 and out % 128 == 0 (MFMA GEMM tiles); the output layer has 1..8 units. */
typedef struct smi_mlp_head smi_mlp_head; /* opaque */
typedef struct smi_mlp_head_config {
  int32_t input_dim;  /* feature width: 6*d (COMET), 4*d (QE), d (MuTox) */
  int32_t n_layers;   /* Linear layers including the output layer, 1..8 */
  int32_t hidden_act; /* 0 ReLU, 1 tanh */
  int32_t out_act;    /* 0 none, 1 tanh (BLASER output_act), 2 sigmoid (MuTox output_prob) */
} smi_mlp_head_config;
typedef struct smi_mlp_head_layer {
  smi_tensor w; /* [out_dim, in_dim], nn.Linear layout */
  smi_tensor b; /* [out_dim] */
  int32_t out_dim;
  int32_t reserved;
} smi_mlp_head_layer;
int smi_mlp_head_create(const smi_mlp_head_config* cfg, const smi_mlp_head_layer* layers, smi_mlp_head** out);
void smi_mlp_head_destroy(smi_mlp_head* head);
/* features: form 0 f16(src); 1 QE [src, mt, src*mt, |mt-src|]; 2 COMET [ref, mt, src*mt, ref*mt,
 * |mt-src|, |mt-ref|] (model.py:95-125), inputs L2-normalised first when norm_emb (model.py:89-93).
 * src/mt/ref: device [rows, d] of `dtype`; out: device f16 [(rows+127)/128*128, blocks*d], pad rows zeroed. */
int smi_head_featurize(int32_t form, const void* src, const void* mt, const void* ref, int32_t dtype, int32_t rows,
                       int32_t d, int32_t norm_emb, void* out_f16, void* stream);
/* x: device f16 [(rows+127)/128*128, input_dim]; out: device fp32 [rows, out_dim];
 * out_act -1 = the configured one, else 0 / 1 / 2 as above. */
int smi_mlp_head_forward(smi_mlp_head* head, const void* x_f16, int32_t rows, int32_t out_act, float* out,
                         void* stream);

/* Host input path ------------------------------------------------------------
 * Replaces the fairseq2n C++ DataPipeline stages between the tokenizer and the model,
 * sonar/inference_pipelines/text.py:226-247 (`.map(truncate)`, `.dynamic_bucket(...)`,
 * `Collater(pad_value)`), fused with the NLLB id assembly of the token encoder
 * ([prefix] piece+1 ... [suffix]).  Pure host code (threads), writes into the caller's
 * (pinned) staging buffer.  SentencePiece segmentation itself is done by the caller with the
 * sentencepiece library's multi-threaded batch encode, as `pieces` (flat int32) + `piece_offsets`. */
int smi_host_token_lengths(const int64_t* piece_offsets, int64_t n, int32_t n_prefix, int32_t n_suffix,
                           int32_t max_seq_len, int32_t* out_lens, int64_t* n_truncated);
int smi_host_dynamic_bucket(const int32_t* lens, int64_t n, int64_t threshold, int32_t max_num,
                            int32_t min_num, int64_t* bounds, int64_t* n_buckets, int64_t* n_open);
int smi_host_collate_nllb(const int32_t* pieces, const int64_t* piece_offsets, const int32_t* lens,
                          int64_t first, int64_t n, const int64_t* prefix, int32_t n_prefix,
                          const int64_t* suffix, int32_t n_suffix, int32_t piece_shift, int64_t pad_value,
                          int64_t* out_ids, int32_t row_stride, int32_t num_threads);

/* WAV decoding on the host (the reference: fairseq2n AudioDecoder over libsndfile,
 * sonar/inference_pipelines/speech.py:292-308).  RIFF/WAVE with PCM 8/16/24/32-bit or IEEE float
 * 32/64-bit samples (incl. WAVE_FORMAT_EXTENSIBLE).  `bytes` is the file image.  smi_host_wav_decode
 * writes float32 [frames, channels] (channel-last, integer PCM scaled by 2^-(bits-1)). */
int smi_host_wav_info(const uint8_t* bytes, int64_t nbytes, int32_t* channels, int32_t* sample_rate, int64_t* frames);
int smi_host_wav_decode(const uint8_t* bytes, int64_t nbytes, float* out, int64_t frames, int32_t channels);
/* The same pair for any covered container, sniffed from the file image: RIFF/WAVE as above, or a native FLAC stream
 * (RFC 9639: every block-size / sample-size code, constant / verbatim / fixed / LPC subframes, Rice and Rice2
 * residuals, wasted bits, left-side / side-right / mid-side stereo, 4-32 bits per sample, header CRC-8 and frame
 * CRC-16 verified; an ID3v2 tag in front is skipped).  Ogg encapsulation returns SMI_ERR_UNSUPPORTED. */
int smi_host_audio_info(const uint8_t* bytes, int64_t nbytes, int32_t* channels, int32_t* sample_rate, int64_t* frames);
int smi_host_audio_decode(const uint8_t* bytes, int64_t nbytes, float* out, int64_t frames, int32_t channels);

/* Building blocks (exported for the parity tests and microbenchmarks) ------ */
/* TILE-MAJOR operand layout (SMI_GEMM_IN_TM / SMI_GEMM_OUT_TM, `tile_major` arguments): a K-major
 * fp16 matrix A[rows][k] (rows % 256 == 0, k % 32 == 0) stored as 16-KiB blocks, block
 * (r/256, c/32) at element offset ((r/256)*(k/32) + c/32) * 8192, and inside a block element
 * (rr = r%256, cc = c%32) at rr*32 + (((cc/8) ^ s(rr)) << 3) + cc%8 with s(rr) = q ^ ((q&1)<<1),
 * q = (rr>>2)&3 (i.e. 0,3,2,1 for q = 0..3: the bank swizzle of the 16x16x32 fragment reads) -- the LDS image of the
 * 256x256 tile engine, so one K slice of a tile is one linear 16 KiB read.  The encoder keeps
 * every GEMM operand (weights, LayerNorm / attention / FFN-inner outputs) in this layout. */
#define SMI_GEMM_IN_TM (1 << 12)  /* x and w are tile-major (m, n % 256 == 0) */
#define SMI_GEMM_OUT_TM (1 << 13) /* f16 output tile-major, as the next GEMM's x (needs IN_TM, ldo == n); with epilogues 8 / 9: the
                                   * f16 residual stream that is read-modified-written is tile-major */
/* dst <- tile-major(src) (inverse == 0) or dst <- row-major(src) (inverse != 0); f16, device. */
int smi_pack_tile_major(const void* src_f16, void* dst_f16, int32_t rows, int32_t k, int32_t inverse,
                        void* stream);
/* out = epilogue(X[m,k] . W[n,k]^T + bias[n]); epi & 0xff: 0 f16 out, 1 f16 ReLU out,
 * 2 fp32 residual accumulate (out += ...), 3 fp32 store, 4 fp32 residual += 0.5 * (...),
 * 5 f16 SiLU out, 6 f16 GLU out (n/2 wide), 7 f16 tanh out, 8 f16 residual accumulate
 * (out_f16 = f16(float(out_f16) + ...), one rounding), 9 the same with 0.5 * (...) (bias may be NULL);
 * (epi >> 8) & 0xf selects the tile engine: 0 auto, 1 128x128, 2 256x256 (needs m,n % 256 == 0);
 * layout flags SMI_GEMM_IN_TM (epilogues 0, 2, 3, 4, 6, 8, 9) and SMI_GEMM_IN_TM|SMI_GEMM_OUT_TM (0, 1, 5, 8, 9; 6 with ldo == n/2, a bias
 * and enough 256x256 tiles for the 4-wave engine -- SMI_ERR_UNSUPPORTED otherwise).
 * m%128==0, n%128==0, k%64==0. */
int smi_gemm_tn(int32_t epi, const void* x_f16, const void* w_f16, const float* bias, void* out,
                int32_t m, int32_t n, int32_t k, int32_t ldo, void* stream);
/* Split-K form of the same product: parts[z][m][n] (slab_dtype SMI_F32 or SMI_F16, row-major, z < ksplit) = X[:, Kz] . W[:, Kz]^T
 * (+ bias in part 0); the consumer sums the slabs (the decode step's and the small-batch encoder's N = model_dim
 * projections).  fp16 slabs saturate at +-65504.  in_tm: x and w tile-major. */
int smi_gemm_tn_splitk(const void* x_f16, const void* w_f16, const float* bias, void* parts, int32_t m, int32_t n, int32_t k,
                       int32_t ksplit, int32_t in_tm, int32_t slab_dtype, void* stream);
/* The decoder's logits projection with its fused softmax statistics (exported for tests; TiedProjection + the beam search's
 * log_softmax, sonar/models/sonar_text/factory.py:300-315, sonar/inference_pipelines/text.py:305-346): x, w and the f16 output
 * tile-major, no bias (m, n % 256 == 0, k % 64 == 0), scale > 0; per (256-column tile t, row r), over the columns c < valid_n of
 * the tile: tile_max[t*m + r] = max_c scale*v, tile_sum[t*m + r] = sum_c exp(scale*v - tile_max), v = the ROUNDED f16 logit. */
int smi_gemm_tn_tile_stats(const void* x_f16_tm, const void* w_f16_tm, void* out_f16_tm, int32_t m, int32_t n, int32_t k,
                           float scale, int32_t valid_n, float* tile_max, float* tile_sum, void* stream);
/* dst[i] = (dst_dtype) src[i] for n elements of DEVICE memory (dtypes: smi_dtype incl. SMI_BF16; fp32 -> bf16 rounds to
 * nearest even).  The bf16 side of the pipelines' `dtype=` argument: `model.to(device, dtype)` / the embeddings returned by
 * TextToEmbeddingModelPipeline.predict (sonar/inference_pipelines/text.py:161-162, 262-268). */
int smi_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n, void* stream);
/* out = f16(LN(x) * w + b); tile_major != 0: out in the tile-major layout ((rows+255)/256*256 rows allocated) */
int smi_layernorm(const float* x, const float* w, const float* b, float eps, void* out_f16,
                  int32_t rows, int32_t d, int32_t tile_major, void* stream);
/* qkv: f16 [t, 3*d] packed rows; cu_seqlens: device int32 [n+1]; ctx: f16 [t, d];
 * tile_major bit 0: ctx written tile-major, bit 1: qkv read tile-major (k = 3*d)
 * ((t+255)/256*256 rows allocated for a tile-major buffer) */
int smi_attention(const void* qkv_f16, const int32_t* cu_seqlens, void* ctx_f16, int32_t n,
                  int32_t max_len, int32_t d, int32_t heads, int32_t tile_major, void* stream);

/* The conformer's relative-position self-attention (exported for tests; fairseq2 RelativePositionSDPA as configured by
 * sonar/models/sonar_speech/factory.py via the w2v-BERT encoder: Shaw-style scores with the learned u / v biases):
 * scores[i][j] = ((q_i + u) . k_j + (q_i + v) . rp[rp_zero + i - j]) / 8 over the frames j of the clip, softmax, times V.
 * qkv: f16 [t, 3*d] packed rows (q | k | v); cu_seqlens: device int32 [n+1]; rp: f16 [rp_rows, d], the projected relative-position
 * table, row rp_zero = distance 0 (rows outside the table are clamped); u_bias / v_bias: fp32 [d]; ctx: f16 [t, d].
 * tile_major bit 0: ctx written tile-major, bit 1: qkv read tile-major (k = 3*d) ((t+255)/256*256 rows allocated for a
 * tile-major buffer).  head_dim = 64. */
int smi_relpos_attention(const void* qkv_f16, const int32_t* cu_seqlens, const void* rp_f16, int32_t rp_zero, int32_t rp_rows,
                         const float* u_bias, const float* v_bias, void* ctx_f16, int32_t n, int32_t max_len, int32_t d,
                         int32_t heads, int32_t tile_major, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SONAR_MI355_H */
