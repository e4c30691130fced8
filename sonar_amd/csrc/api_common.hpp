// Shared host-side helpers of the C-ABI translation units (api.hip, decoder_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <utility>

#include "../../include/sonar_mi355.h"
#include "kernels.hpp"

namespace smi_host {

using namespace smi;

std::string& last_error();

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return smi_host::fail(_e == hipErrorOutOfMemory ? SMI_ERR_OOM : SMI_ERR_HIP,        \
                            "%s failed: %s", #expr, hipGetErrorString(_e));               \
  } while (0)

inline bool have_device() {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
    o.p = nullptr;
    o.bytes = 0;
  }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      bytes = o.bytes;
      o.p = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  hipError_t alloc(size_t n) {
    release();
    if (n == 0) return hipSuccess;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  // grow-only allocation (contents are not preserved)
  hipError_t reserve(size_t n) { return n <= bytes ? hipSuccess : alloc(n); }
  template <typename T>
  T* as() const {
    return (T*)p;
  }
};

// Copy a caller tensor into a freshly allocated device buffer as fp16 or fp32;
// `pad_numel` (>= numel) zero-pads the destination.
int upload(const smi_tensor& t, int64_t expect_numel, bool want_f16, DevBuf& dst, const char* name,
           int64_t pad_numel = 0);

// w (row-major [rows][K] fp16 on the device) -> tile-major (common.hpp), in a fresh buffer
inline int to_tile_major(DevBuf& w, int rows, int K) {
  DevBuf t;
  HIP_TRY(t.alloc(w.bytes));
  HIP_TRY(launch_pack_tile_major(w.as<f16>(), t.as<f16>(), rows, K, 0, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  w = std::move(t);
  return SMI_OK;
}

// Text encoder on the generic-dimension kernels (flex_encoder.hip): configurations the MFMA engines do not tile for
struct FlexEncoder;
bool flex_encoder_wanted(const smi_text_encoder_config& c);
int flex_encoder_create(const smi_text_encoder_config* cfg, const smi_text_encoder_weights* w, FlexEncoder** out);
void flex_encoder_destroy(FlexEncoder* e);
int64_t flex_encoder_bytes(const FlexEncoder* e);
int flex_encoder_forward(FlexEncoder* e, const int64_t* ids, const int32_t* seq_lens, int n, int s, void* out_emb,
                         void* out_encoded, int out_dtype, int32_t* bad_ids_dev, hipStream_t stream);

}  // namespace smi_host
