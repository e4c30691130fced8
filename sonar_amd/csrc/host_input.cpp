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

// ---- RIFF / WAVE decoding (the reference decodes audio with fairseq2n's libsndfile AudioDecoder,
// sonar/inference_pipelines/speech.py:292-308; this covers the WAV container: PCM 8/16/24/32-bit,
// IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE, any channel count).  Pure byte work on the host.
namespace {
struct WavFmt {
  int format = 0, channels = 0, bits = 0, block_align = 0;
  int64_t rate = 0, data_off = -1, data_bytes = 0;
};
inline uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int parse_wav(const uint8_t* b, int64_t n, WavFmt& f) {
  if (n < 12 || memcmp(b, "RIFF", 4) || memcmp(b + 8, "WAVE", 4)) return fail(SMI_ERR_INVALID_ARG, "not a RIFF/WAVE file");
  int64_t p = 12;
  bool have_fmt = false;
  while (p + 8 <= n) {
    const uint32_t sz = rd32(b + p + 4);
    const uint8_t* c = b + p + 8;
    if (!memcmp(b + p, "fmt ", 4)) {
      if (sz < 16 || p + 8 + 16 > n) return fail(SMI_ERR_INVALID_ARG, "truncated fmt chunk");
      f.format = rd16(c);
      f.channels = rd16(c + 2);
      f.rate = rd32(c + 4);
      f.block_align = rd16(c + 12);
      f.bits = rd16(c + 14);
      if (f.format == 0xFFFE) {  // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
        if (sz < 40 || p + 8 + 40 > n) return fail(SMI_ERR_INVALID_ARG, "truncated extensible fmt chunk");
        f.format = rd16(c + 24);
      }
      have_fmt = true;
    } else if (!memcmp(b + p, "data", 4)) {
      f.data_off = p + 8;
      f.data_bytes = std::min<int64_t>(sz, n - (p + 8));  // streamed files may carry a bogus size
      break;
    }
    p += 8 + (int64_t)sz + (sz & 1);
  }
  if (!have_fmt || f.data_off < 0) return fail(SMI_ERR_INVALID_ARG, "WAV file without fmt/data chunk");
  if (f.channels <= 0 || f.bits <= 0 || f.bits % 8) return fail(SMI_ERR_UNSUPPORTED, "WAV: %d channels, %d bits", f.channels, f.bits);
  const bool pcm = f.format == 1 && (f.bits == 8 || f.bits == 16 || f.bits == 24 || f.bits == 32);
  const bool flt = f.format == 3 && (f.bits == 32 || f.bits == 64);
  if (!pcm && !flt) return fail(SMI_ERR_UNSUPPORTED, "WAV format tag %d with %d bits is not covered (PCM / IEEE float only)", f.format, f.bits);
  const int frame_bytes = f.channels * (f.bits / 8);
  if (f.block_align < frame_bytes) f.block_align = frame_bytes;
  return SMI_OK;
}
}  // namespace

// channels / sample_rate / frames of a WAV file image
int smi_host_wav_info(const uint8_t* bytes, int64_t nbytes, int32_t* channels, int32_t* sample_rate, int64_t* frames) {
  if (!bytes || !channels || !sample_rate || !frames) return fail(SMI_ERR_INVALID_ARG, "null argument");
  WavFmt f;
  if (int rc = parse_wav(bytes, nbytes, f)) return rc;
  *channels = f.channels;
  *sample_rate = (int32_t)f.rate;
  *frames = f.data_bytes / f.block_align;
  return SMI_OK;
}

// out: float32 [frames, channels] (file order, channel-last as fairseq2's AudioDecoder returns it),
// integer PCM scaled to [-1, 1) by 2^-(bits-1) as libsndfile does for float reads.
int smi_host_wav_decode(const uint8_t* bytes, int64_t nbytes, float* out, int64_t frames, int32_t channels) {
  if (!bytes || !out) return fail(SMI_ERR_INVALID_ARG, "null argument");
  WavFmt f;
  if (int rc = parse_wav(bytes, nbytes, f)) return rc;
  if (channels != f.channels || frames != f.data_bytes / f.block_align)
    return fail(SMI_ERR_INVALID_ARG, "frames/channels do not match smi_host_wav_info");
  const int bps = f.bits / 8;
  const uint8_t* d = bytes + f.data_off;
  for (int64_t i = 0; i < frames; ++i) {
    const uint8_t* fr = d + i * f.block_align;
    for (int c = 0; c < channels; ++c) {
      const uint8_t* s = fr + c * bps;
      float v;
      if (f.format == 3) {
        if (bps == 4) {
          memcpy(&v, s, 4);
        } else {
          double dv;
          memcpy(&dv, s, 8);
          v = (float)dv;
        }
      } else if (bps == 1) {
        v = ((int)s[0] - 128) * (1.0f / 128.0f);
      } else if (bps == 2) {
        v = (int16_t)rd16(s) * (1.0f / 32768.0f);
      } else if (bps == 3) {
        const int32_t x = (int32_t)((uint32_t)s[0] << 8 | (uint32_t)s[1] << 16 | (uint32_t)s[2] << 24) >> 8;
        v = x * (1.0f / 8388608.0f);
      } else {
        v = (float)((double)(int32_t)rd32(s) * (1.0 / 2147483648.0));
      }
      out[i * channels + c] = v;
    }
  }
  return SMI_OK;
}

}  // extern "C"
