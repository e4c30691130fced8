// Host input path of the text pipelines: the stages the reference runs in fairseq2n's C++
// DataPipeline between the tokenizer and the model (sonar/inference_pipelines/text.py:226-247):
// token-id assembly (NLLB id layout), truncation, dynamic bucketing and right-padded collation.
// SentencePiece itself stays in its own library (multi-threaded batch encode); everything after it
// is integer / byte work done here by a few host threads, writing straight into the caller's
// (pinned) staging buffer so the H2D copy can be asynchronous.  No device code in this file.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "api_common.hpp"

using namespace smi_host;

namespace {

// run fn(begin, end) over [0, n) on up to `threads` host threads
template <typename F>
void parallel_rows(int64_t n, int threads, F fn) {
  const int t = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n / 64));
  if (t <= 1) {
    fn((int64_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t per = (n + t - 1) / t;
  for (int i = 0; i < t; ++i) {
    const int64_t b = i * per, e = std::min(n, b + per);
    if (b >= e) break;
    pool.emplace_back([=] { fn(b, e); });
  }
  for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

// Token length of every sequence after [prefix] pieces [suffix] assembly and truncation to
// max_seq_len (<= 0: no limit) -- text.py:213-219,232-233: the cut happens AFTER the suffix is
// appended, so an over-long input loses its EOS.  Returns the number of truncated sequences.
int smi_host_token_lengths(const int64_t* piece_offsets, int64_t n, int32_t n_prefix, int32_t n_suffix,
                           int32_t max_seq_len, int32_t* out_lens, int64_t* n_truncated) {
  if (!piece_offsets || !out_lens || n < 0 || n_prefix < 0 || n_suffix < 0)
    return fail(SMI_ERR_INVALID_ARG, "bad argument");
  int64_t cut = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t np = piece_offsets[i + 1] - piece_offsets[i];
    if (np < 0) return fail(SMI_ERR_INVALID_ARG, "piece_offsets must be non-decreasing");
    int64_t len = np + n_prefix + n_suffix;
    if (max_seq_len > 0 && len > max_seq_len) {
      len = max_seq_len;
      ++cut;
    }
    if (len > 0x7fffffff) return fail(SMI_ERR_UNSUPPORTED, "sequence too long");
    out_lens[i] = (int32_t)len;
  }
  if (n_truncated) *n_truncated = cut;
  return SMI_OK;
}

// fairseq2 `.dynamic_bucket(threshold, cost_fn=len, min_num_examples, max_num_examples,
// drop_remainder=False)` (text.py:234-240) over a stream of lengths: a bucket closes once its
// summed length reaches `threshold` (and it holds >= min_num) or it holds max_num sequences.
// bounds[0..*n_buckets] are the bucket boundaries; `*n_open` = sequences of the trailing bucket
// that has NOT closed yet (the caller carries them into the next chunk, or emits them at the end).
int smi_host_dynamic_bucket(const int32_t* lens, int64_t n, int64_t threshold, int32_t max_num,
                            int32_t min_num, int64_t* bounds, int64_t* n_buckets, int64_t* n_open) {
  if (!lens || !bounds || !n_buckets || !n_open || n < 0 || threshold <= 0 || max_num <= 0 || min_num < 1)
    return fail(SMI_ERR_INVALID_ARG, "bad argument");
  int64_t nb = 0, start = 0, cost = 0;
  bounds[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    cost += lens[i];
    const int64_t cnt = i + 1 - start;
    if ((cost >= threshold && cnt >= min_num) || cnt >= max_num) {
      bounds[++nb] = i + 1;
      start = i + 1;
      cost = 0;
    }
  }
  *n_buckets = nb;
  *n_open = n - start;
  return SMI_OK;
}

// Collater(pad_value) (text.py:241) fused with the id assembly: row i of out_ids [n, row_stride]
// = ([prefix] (piece + piece_shift)... [suffix])[:lens[i]] right-padded with pad_value.
// lens from smi_host_token_lengths; row_stride >= max(lens).  `first` selects the sequence range
// [first, first + n) of the piece arrays (one bucket of a larger chunk).
int smi_host_collate_nllb(const int32_t* pieces, const int64_t* piece_offsets, const int32_t* lens,
                          int64_t first, int64_t n, const int64_t* prefix, int32_t n_prefix,
                          const int64_t* suffix, int32_t n_suffix, int32_t piece_shift, int64_t pad_value,
                          int64_t* out_ids, int32_t row_stride, int32_t num_threads) {
  if (!piece_offsets || !lens || !out_ids || n < 0 || first < 0 || n_prefix < 0 || n_suffix < 0 ||
      (n_prefix && !prefix) || (n_suffix && !suffix) || row_stride < 0)
    return fail(SMI_ERR_INVALID_ARG, "bad argument");
  for (int64_t i = first; i < first + n; ++i)
    if (lens[i] > row_stride) return fail(SMI_ERR_INVALID_ARG, "row_stride %d < sequence length %d", row_stride, lens[i]);
  if (!pieces && piece_offsets[first + n] != piece_offsets[first]) return fail(SMI_ERR_INVALID_ARG, "null pieces");
  parallel_rows(n, num_threads, [=](int64_t b, int64_t e) {
    for (int64_t r = b; r < e; ++r) {
      const int64_t i = first + r;
      int64_t* row = out_ids + r * (int64_t)row_stride;
      const int32_t len = lens[i];
      const int32_t* p = pieces + piece_offsets[i];
      const int64_t np = piece_offsets[i + 1] - piece_offsets[i];
      int32_t w = 0;
      for (int32_t k = 0; k < n_prefix && w < len; ++k) row[w++] = prefix[k];
      for (int64_t k = 0; k < np && w < len; ++k) row[w++] = (int64_t)p[k] + piece_shift;
      for (int32_t k = 0; k < n_suffix && w < len; ++k) row[w++] = suffix[k];
      for (; w < row_stride; ++w) row[w] = pad_value;
    }
  });
  return SMI_OK;
}

}  // extern "C"
