// Host audio decoding beyond RIFF/WAVE: native FLAC streams (RFC 9639).
// The reference decodes audio files with fairseq2n's AudioDecoder over libsndfile
// (sonar/inference_pipelines/speech.py:292-308), which reads FLAC next to WAV; libsndfile is not part of
// this stack, so the container is decoded here: STREAMINFO, frame headers (fixed / variable block size, every
// block-size / sample-rate / sample-size code), the four subframe types (constant, verbatim, fixed predictors of
// order 0-4, LPC of order 1-32), Rice / Rice2 residuals with escape partitions, wasted bits, the three stereo
// decorrelations, CRC-8 of every frame header and CRC-16 of every frame.  Samples are returned as libsndfile's
// float reads return them: integer / 2^(bits-1), channel-last.  Ogg encapsulation is not covered.
// Pure byte / integer work on the host; no device code in this file.
#include <cstdint>
#include <cstring>
#include <vector>

#include "api_common.hpp"

using namespace smi_host;

namespace {

struct Crc {
  uint8_t t8[256];
  uint16_t t16[256];
  Crc() {
    for (int i = 0; i < 256; ++i) {
      uint8_t c = (uint8_t)i;
      for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);  // x^8 + x^2 + x + 1
      t8[i] = c;
      uint16_t d = (uint16_t)(i << 8);
      for (int b = 0; b < 8; ++b) d = (uint16_t)((d & 0x8000) ? (d << 1) ^ 0x8005 : d << 1);  // x^16 + x^15 + x^2 + 1
      t16[i] = d;
    }
  }
  uint8_t crc8(const uint8_t* p, int64_t n) const {
    uint8_t c = 0;
    for (int64_t i = 0; i < n; ++i) c = t8[c ^ p[i]];
    return c;
  }
  uint16_t crc16(const uint8_t* p, int64_t n) const {
    uint16_t c = 0;
    for (int64_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ t16[(c >> 8) ^ p[i]]);
    return c;
  }
};
const Crc& crc() {
  static const Crc c;
  return c;
}

// MSB-first bit reader over a byte range; reads past the end set `bad`
struct BitReader {
  const uint8_t* p;
  int64_t nbits, pos = 0;
  bool bad = false;
  BitReader(const uint8_t* p_, int64_t nbytes) : p(p_), nbits(nbytes * 8) {}
  uint64_t read(int n) {  // 0 <= n <= 57
    if (n == 0) return 0;
    if (pos + n > nbits) {
      bad = true;
      pos = nbits;
      return 0;
    }
    uint64_t v = 0;
    int64_t byte = pos >> 3;
    const int off = (int)(pos & 7);
    const int need = (off + n + 7) >> 3;
    for (int i = 0; i < need; ++i) v = (v << 8) | p[byte + i];
    v >>= need * 8 - off - n;
    pos += n;
    return v & ((n == 64) ? ~0ull : ((1ull << n) - 1));
  }
  int64_t read_signed(int n) {  // two's complement, 1 <= n <= 33 (32-bit audio side channel)
    if (n == 0) return 0;
    uint64_t v;
    if (n > 32) {
      v = read(n - 32) << 32;
      v |= read(32);
    } else {
      v = read(n);
    }
    const uint64_t sign = 1ull << (n - 1);
    return (int64_t)((v ^ sign) - sign);
  }
  // number of 0 bits in front of the next 1 bit (which is consumed)
  uint32_t read_unary() {
    uint32_t q = 0;
    while (true) {
      if (pos >= nbits) {
        bad = true;
        return q;
      }
      const int off = (int)(pos & 7);
      const uint8_t rest = (uint8_t)(p[pos >> 3] << off);  // remaining bits of the byte, left aligned
      if (rest) {
        const int z = __builtin_clz((unsigned)rest) - 24;
        q += z;
        pos += z + 1;
        return q;
      }
      q += 8 - off;
      pos += 8 - off;
    }
  }
  void align() { pos = (pos + 7) & ~7ll; }
};

struct FlacInfo {
  int channels = 0, bps = 0, min_block = 0, max_block = 0;
  int64_t rate = 0, total = 0, first_frame = 0;
};

int flac_header(const uint8_t* b, int64_t n, FlacInfo& f) {
  int64_t p = 0;
  if (n >= 10 && !memcmp(b, "ID3", 3)) {  // an ID3v2 tag in front of the stream: 10-byte header + syncsafe size
    p = 10 + (((int64_t)(b[6] & 0x7f) << 21) | ((b[7] & 0x7f) << 14) | ((b[8] & 0x7f) << 7) | (b[9] & 0x7f));
    if (b[5] & 0x10) p += 10;  // footer
  }
  if (p + 4 > n || memcmp(b + p, "fLaC", 4)) return fail(SMI_ERR_INVALID_ARG, "not a FLAC stream");
  p += 4;
  bool have_info = false;
  while (true) {
    if (p + 4 > n) return fail(SMI_ERR_INVALID_ARG, "FLAC: truncated metadata");
    const bool last = b[p] & 0x80;
    const int type = b[p] & 0x7f;
    const int64_t len = ((int64_t)b[p + 1] << 16) | (b[p + 2] << 8) | b[p + 3];
    p += 4;
    if (p + len > n) return fail(SMI_ERR_INVALID_ARG, "FLAC: truncated metadata block");
    if (type == 0) {
      if (len < 34) return fail(SMI_ERR_INVALID_ARG, "FLAC: short STREAMINFO");
      const uint8_t* s = b + p;
      f.min_block = (s[0] << 8) | s[1];
      f.max_block = (s[2] << 8) | s[3];
      f.rate = ((int64_t)s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      f.channels = ((s[12] >> 1) & 7) + 1;
      f.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      f.total = ((int64_t)(s[13] & 0xf) << 32) | ((int64_t)s[14] << 24) | (s[15] << 16) | (s[16] << 8) | s[17];
      have_info = true;
    } else if (type == 127) {
      return fail(SMI_ERR_INVALID_ARG, "FLAC: invalid metadata block type");
    }
    p += len;
    if (last) break;
  }
  if (!have_info) return fail(SMI_ERR_INVALID_ARG, "FLAC: no STREAMINFO block");
  if (f.bps < 4 || f.bps > 32 || f.rate <= 0) return fail(SMI_ERR_UNSUPPORTED, "FLAC: %d bits, %lld Hz", f.bps, (long long)f.rate);
  f.first_frame = p;
  // STREAMINFO's 36-bit sample count comes from an untrusted header and sizes the caller's output buffer: refuse a
  // count the stream cannot hold.  The smallest frame (5-byte header + 1 CRC-8, one constant subframe per channel,
  // CRC-16) is >= 9 bytes and carries at most 65535 samples per channel.
  if (f.total > ((n - p) / 9 + 1) * 65535)
    return fail(SMI_ERR_INVALID_ARG, "FLAC: STREAMINFO claims %lld samples, the stream has %lld bytes of frames",
                (long long)f.total, (long long)(n - p));
  return SMI_OK;
}

// residual of one subframe into r[order .. block)
int flac_residual(BitReader& br, int64_t* r, int block, int order) {
  const int method = (int)br.read(2);
  if (method > 1) return fail(SMI_ERR_INVALID_ARG, "FLAC: reserved residual coding method");
  const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
  const int porder = (int)br.read(4);
  const int parts = 1 << porder;
  if (porder > 0 && ((block >> porder) << porder) != block) return fail(SMI_ERR_INVALID_ARG, "FLAC: partition order does not divide the block");
  if ((block >> porder) < order && porder > 0) return fail(SMI_ERR_INVALID_ARG, "FLAC: predictor order exceeds the first partition");
  int i = order;
  for (int pt = 0; pt < parts; ++pt) {
    const int count = (porder == 0 ? block : block >> porder) - (pt == 0 ? order : 0);
    if (count < 0) return fail(SMI_ERR_INVALID_ARG, "FLAC: negative partition size");
    const int k = (int)br.read(pbits);
    if (k == esc) {
      const int nb = (int)br.read(5);
      for (int j = 0; j < count; ++j) r[i++] = nb ? br.read_signed(nb) : 0;
    } else {
      for (int j = 0; j < count; ++j) {
        const uint64_t q = br.read_unary();
        const uint64_t u = (q << k) | br.read(k);
        r[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
      }
    }
    if (br.bad) return fail(SMI_ERR_INVALID_ARG, "FLAC: truncated residual");
  }
  return SMI_OK;
}

int flac_subframe(BitReader& br, int64_t* s, int block, int bps) {
  if (br.read(1)) return fail(SMI_ERR_INVALID_ARG, "FLAC: subframe padding bit set");
  const int type = (int)br.read(6);
  int wasted = 0;
  if (br.read(1)) wasted = (int)br.read_unary() + 1;
  bps -= wasted;
  if (bps <= 0) return fail(SMI_ERR_INVALID_ARG, "FLAC: wasted bits exceed the sample size");
  if (type == 0) {  // constant
    const int64_t v = br.read_signed(bps);
    for (int i = 0; i < block; ++i) s[i] = v;
  } else if (type == 1) {  // verbatim
    for (int i = 0; i < block; ++i) s[i] = br.read_signed(bps);
  } else if (type >= 8 && type <= 12) {  // fixed predictor, order type - 8
    const int order = type - 8;
    if (order > block) return fail(SMI_ERR_INVALID_ARG, "FLAC: predictor order exceeds the block");
    for (int i = 0; i < order; ++i) s[i] = br.read_signed(bps);
    if (int rc = flac_residual(br, s, block, order)) return rc;
    // (unsigned arithmetic: a crafted residual may overflow 64 bits before the frame CRC is checked; wrapping is
    //  defined for uint64_t, the garbage is rejected by the CRC-16 / range checks afterwards)
    auto U = [&](int i) { return (uint64_t)s[i]; };
    switch (order) {
      case 1: for (int i = 1; i < block; ++i) s[i] = (int64_t)(U(i) + U(i - 1)); break;
      case 2: for (int i = 2; i < block; ++i) s[i] = (int64_t)(U(i) + 2 * U(i - 1) - U(i - 2)); break;
      case 3: for (int i = 3; i < block; ++i) s[i] = (int64_t)(U(i) + 3 * U(i - 1) - 3 * U(i - 2) + U(i - 3)); break;
      case 4: for (int i = 4; i < block; ++i) s[i] = (int64_t)(U(i) + 4 * U(i - 1) - 6 * U(i - 2) + 4 * U(i - 3) - U(i - 4)); break;
      default: break;
    }
  } else if (type >= 32) {  // LPC, order type - 31
    const int order = type - 31;
    if (order > block) return fail(SMI_ERR_INVALID_ARG, "FLAC: predictor order exceeds the block");
    for (int i = 0; i < order; ++i) s[i] = br.read_signed(bps);
    const int prec = (int)br.read(4) + 1;
    if (prec == 16) return fail(SMI_ERR_INVALID_ARG, "FLAC: invalid coefficient precision");
    const int shift = (int)br.read_signed(5);
    if (shift < 0) return fail(SMI_ERR_INVALID_ARG, "FLAC: negative prediction shift");
    int64_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.read_signed(prec);
    if (int rc = flac_residual(br, s, block, order)) return rc;
    for (int i = order; i < block; ++i) {
      uint64_t acc = 0;  // wrap-safe (see the fixed predictors)
      for (int j = 0; j < order; ++j) acc += (uint64_t)coef[j] * (uint64_t)s[i - 1 - j];
      s[i] = (int64_t)((uint64_t)s[i] + (uint64_t)((int64_t)acc >> shift));
    }
  } else {
    return fail(SMI_ERR_INVALID_ARG, "FLAC: reserved subframe type %d", type);
  }
  if (br.bad) return fail(SMI_ERR_INVALID_ARG, "FLAC: truncated subframe");
  if (wasted)
    for (int i = 0; i < block; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
  return SMI_OK;
}

// Decodes every frame; out (may be null: count only) receives float32 [frames, channels] up to cap frames.
int flac_decode_stream(const uint8_t* b, int64_t n, const FlacInfo& f, float* out, int64_t cap, int64_t* frames_out) {
  int64_t p = f.first_frame, done = 0;
  std::vector<int64_t> buf;
  const double scale = 1.0 / (double)(1ull << (f.bps - 1));
  while (p + 2 <= n && (f.total == 0 || done < f.total)) {
    if (!(b[p] == 0xff && (b[p + 1] & 0xfe) == 0xf8)) {
      if (n - p >= 3 && !memcmp(b + p, "TAG", 3)) break;  // an ID3v1 tag after the last frame
      return fail(SMI_ERR_INVALID_ARG, "FLAC: lost frame sync at byte %lld", (long long)p);
    }
    BitReader br(b + p, n - p);
    br.read(15);
    br.read(1);  // blocking strategy: only changes the meaning of the coded number, which is not needed here
    const int bs_code = (int)br.read(4), sr_code = (int)br.read(4), ch_code = (int)br.read(4), ss_code = (int)br.read(3);
    if (br.read(1)) return fail(SMI_ERR_INVALID_ARG, "FLAC: reserved frame header bit set");
    {  // UTF-8-style coded frame / sample number: 1-7 bytes
      const int first = (int)br.read(8);
      int extra = 0;
      if (first & 0x80) {
        while (extra < 7 && (first & (0x80 >> extra))) ++extra;
        if (extra < 2 || extra > 7) return fail(SMI_ERR_INVALID_ARG, "FLAC: bad coded number");
        extra -= 1;
      }
      for (int i = 0; i < extra; ++i)
        if ((br.read(8) & 0xc0) != 0x80) return fail(SMI_ERR_INVALID_ARG, "FLAC: bad coded number");
    }
    int block;
    if (bs_code == 0) return fail(SMI_ERR_INVALID_ARG, "FLAC: reserved block size code");
    else if (bs_code == 1) block = 192;
    else if (bs_code <= 5) block = 576 << (bs_code - 2);
    else if (bs_code == 6) block = (int)br.read(8) + 1;
    else if (bs_code == 7) block = (int)br.read(16) + 1;
    else block = 256 << (bs_code - 8);
    if (sr_code == 12) br.read(8);
    else if (sr_code == 13 || sr_code == 14) br.read(16);
    else if (sr_code == 15) return fail(SMI_ERR_INVALID_ARG, "FLAC: invalid sample rate code");
    static const int ss_bits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    int bps = ss_bits[ss_code];
    if (bps < 0) return fail(SMI_ERR_INVALID_ARG, "FLAC: reserved sample size code");
    if (bps == 0) bps = f.bps;
    if (bps != f.bps) return fail(SMI_ERR_UNSUPPORTED, "FLAC: sample size changes inside the stream");
    if (br.bad) return fail(SMI_ERR_INVALID_ARG, "FLAC: truncated frame header");
    const int64_t hdr_bytes = br.pos >> 3;
    const uint8_t c8 = (uint8_t)br.read(8);
    if (br.bad || crc().crc8(b + p, hdr_bytes) != c8) return fail(SMI_ERR_INVALID_ARG, "FLAC: frame header CRC mismatch at byte %lld", (long long)p);
    int channels;
    if (ch_code < 8) channels = ch_code + 1;
    else if (ch_code <= 10) channels = 2;
    else return fail(SMI_ERR_INVALID_ARG, "FLAC: reserved channel assignment");
    if (channels != f.channels) return fail(SMI_ERR_UNSUPPORTED, "FLAC: channel count changes inside the stream");
    buf.resize((size_t)channels * block);
    for (int c = 0; c < channels; ++c) {
      // the side channel of a decorrelated pair carries one more bit
      const int extra = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
      if (int rc = flac_subframe(br, buf.data() + (size_t)c * block, block, bps + extra)) return rc;
    }
    br.align();
    const int64_t body_bytes = br.pos >> 3;
    const uint16_t c16 = (uint16_t)br.read(16);
    if (br.bad) return fail(SMI_ERR_INVALID_ARG, "FLAC: truncated frame");
    if (crc().crc16(b + p, body_bytes) != c16) return fail(SMI_ERR_INVALID_ARG, "FLAC: frame CRC mismatch at byte %lld", (long long)p);
    int64_t* c0 = buf.data();
    int64_t* c1 = buf.data() + block;
    if (ch_code == 8) {  // left, side
      for (int i = 0; i < block; ++i) c1[i] = c0[i] - c1[i];
    } else if (ch_code == 9) {  // side, right
      for (int i = 0; i < block; ++i) c0[i] += c1[i];
    } else if (ch_code == 10) {  // mid, side
      for (int i = 0; i < block; ++i) {
        const int64_t side = c1[i];
        const int64_t mid = (int64_t)(((uint64_t)c0[i] << 1) | (uint64_t)(side & 1));
        c0[i] = (mid + side) >> 1;
        c1[i] = (mid - side) >> 1;
      }
    }
    int take = block;
    if (f.total && done + take > f.total) take = (int)(f.total - done);
    if (out) {
      if (done + take > cap) return fail(SMI_ERR_INVALID_ARG, "FLAC: more samples than smi_host_audio_info reported");
      for (int i = 0; i < take; ++i)
        for (int c = 0; c < channels; ++c) out[(done + i) * channels + c] = (float)((double)buf[(size_t)c * block + i] * scale);
    }
    done += take;
    p += br.pos >> 3;
  }
  if (f.total && done != f.total) return fail(SMI_ERR_INVALID_ARG, "FLAC: stream ends after %lld of %lld samples", (long long)done, (long long)f.total);
  *frames_out = done;
  return SMI_OK;
}

// native FLAC, optionally behind an ID3v2 tag (an MP3 file starts with the same tag: look for the marker after it)
bool is_flac(const uint8_t* b, int64_t n) {
  if (n >= 4 && !memcmp(b, "fLaC", 4)) return true;
  if (n >= 10 && !memcmp(b, "ID3", 3)) {
    int64_t p = 10 + (((int64_t)(b[6] & 0x7f) << 21) | ((b[7] & 0x7f) << 14) | ((b[8] & 0x7f) << 7) | (b[9] & 0x7f));
    if (b[5] & 0x10) p += 10;
    return p + 4 <= n && !memcmp(b + p, "fLaC", 4);
  }
  return false;
}

}  // namespace

extern "C" {

// Container-sniffing front of the audio decoders: RIFF/WAVE (host_input.cpp) or native FLAC.
int smi_host_audio_info(const uint8_t* bytes, int64_t nbytes, int32_t* channels, int32_t* sample_rate, int64_t* frames) {
  if (!bytes || !channels || !sample_rate || !frames) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (!is_flac(bytes, nbytes)) {
    if (nbytes >= 4 && !memcmp(bytes, "OggS", 4)) return fail(SMI_ERR_UNSUPPORTED, "Ogg containers are not covered (WAV and native FLAC are)");
    if (nbytes >= 3 && (!memcmp(bytes, "ID3", 3) || (bytes[0] == 0xff && (bytes[1] & 0xe0) == 0xe0)))
      return fail(SMI_ERR_UNSUPPORTED, "MPEG audio is not covered (WAV and native FLAC are)");
    return smi_host_wav_info(bytes, nbytes, channels, sample_rate, frames);
  }
  FlacInfo f;
  if (int rc = flac_header(bytes, nbytes, f)) return rc;
  *channels = f.channels;
  *sample_rate = (int32_t)f.rate;
  if (f.total) {
    *frames = f.total;
    return SMI_OK;
  }
  return flac_decode_stream(bytes, nbytes, f, nullptr, 0, frames);  // unknown length: count by decoding
}

int smi_host_audio_decode(const uint8_t* bytes, int64_t nbytes, float* out, int64_t frames, int32_t channels) {
  if (!bytes || !out) return fail(SMI_ERR_INVALID_ARG, "null argument");
  if (!is_flac(bytes, nbytes)) return smi_host_wav_decode(bytes, nbytes, out, frames, channels);
  FlacInfo f;
  if (int rc = flac_header(bytes, nbytes, f)) return rc;
  if (channels != f.channels) return fail(SMI_ERR_INVALID_ARG, "frames/channels do not match smi_host_audio_info");
  int64_t got = 0;
  if (int rc = flac_decode_stream(bytes, nbytes, f, out, frames, &got)) return rc;
  if (got != frames) return fail(SMI_ERR_INVALID_ARG, "frames/channels do not match smi_host_audio_info");
  return SMI_OK;
}

}  // extern "C"
