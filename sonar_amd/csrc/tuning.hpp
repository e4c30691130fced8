// Tuning registry: every A/B switch of the library is ONE integer in this table, set through the C-ABI (smi_tuning_set,
// include/sonar_mi355.h).  The library never reads the process environment: a production process that does not call
// smi_tuning_set runs the shipped defaults, and a switch is an atomic that any host thread may read while another sets it
// (round 4 read `getenv` per launch from up to four decode-chain threads: a race with any `setenv` of the host program).
// The Python loader (sonar_amd/_lib.py) forwards `SMI_<NAME>=<int>` environment variables ONCE, at load, so the measurement
// tooling of tools/ keeps its `env SMI_X=.. probe` form; tests pin switches with `_lib.tuning(NAME=value)`.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>

namespace smi {

// name, meaning (default in brackets)
#define SMI_TUNE_LIST(X)                                                                                                   \
  X(ENC_LNFOLD)      /* text encoder: 0 LayerNorm launches, 1 fold with the exact mean term, [2] fold with centred weights (at create) */ \
  X(ENC_X_TM)        /* text encoder: [1] tile-major fp16 residual stream, 0 row-major */                               \
  X(ENC_SB)          /* text encoder small batches: [1] decoder-shaped layer schedule, 0 round-2 schedule */            \
  X(ENC_SLAB_F16)    /* text encoder small batches: [1] fp16 split-K partial sums on an fp16 stream, 0 fp32 */          \
  X(ATT_ORDER)       /* encoder attention: [1] sentence-major dispatch, 0 head-major */                                 \
  X(PREFETCH)        /* [1] weight prefetch by the surplus workgroups of the sum + LayerNorm launches, 0 none */        \
  X(DEC_KS_OUT)      /* decode step: split-K parts of the attention-output projection ([0] automatic, 1 no split) */    \
  X(DEC_KS_FFN)      /* decode step: split-K parts of the FFN-output projection ([0] automatic) */                      \
  X(DEC_FFN1_ENGINE) /* decode step FFN-inner: [-1] automatic, 0 the GEMM's choice, 1 128x128 family, 2 256x256 */      \
  X(DEC_LOGITS_GRID) /* persistent workgroups of a chained call's logits GEMM ([0] all CUs) */                          \
  X(DEC_SLAB_F16)    /* decoder split-K partial sums: [-1] the engine setting (smi_text_decoder_set_slab_dtype), 0 fp32, 1 fp16 */ \
  X(DEC_LOGITS_F16)  /* beam-search logits storage: [-1] the engine setting, 0 fp32, 1 fp16 */                          \
  X(DEC_CHAINS)      /* independent decode chains ([0] the engine's policy) */                                          \
  X(G2_RASTER)       /* 256x256 engine: [2] XCD-owned m-groups, 0 id-order raster */                                    \
  X(G2V2)            /* [1] 4-wave engine (gemm_v2.hpp) for the tile-major fp16 outputs it covers, 2 but for the logits, 0 off */ \
  X(G2V2_MIN)        /* 256x256 tiles from which a launch takes the 4-wave engine ([128]) */                            \
  X(DEC_M160)        /* [1] 128 / 160 / 192-row lone units for FFN projections whose 256-row tiles leave CUs idle (gemm_v2_lone.hip), 0 off, 2 also short K loops */ \
  X(GT_RING)         /* stages of the lone-tile ring ([4]; anything else: never use it) */                              \
  X(LONE)            /* [1] 64x64 lone-tile units where they fit, 0 round 3's 128x128 ring */                           \
  X(LONE16)          /* [1] k-sliced 64x64 unit on tile-major operands, 0 the LDS-ring unit */                          \
  X(LONE_KS)         /* split-K part count in the lone-tile regime ([0] the cost model) */                              \
  X(G2_AUTO_MIN)     /* 256x256 tiles from which the automatic choice takes the 256x256 engine ([128]) */               \
  X(G2_SPLITK_MIN)   /* 256x256 units from which a split-K launch takes the 256x256 engine ([96]) */                    \
  X(SPEECH_MID_TM)   /* speech encoder: [1] tile-major outputs of the per-clip kernels (at create) */                   \
  X(SPEECH_RP_LDS)   /* speech attention: [1] position rows staged once per workgroup through an LDS ring + fp16 score pad, 0 per-wave global loads + fp32 pad */ \
  X(SPEECH_GLU_TM)   /* speech encoder: [1] tile-major GLU output (pointwise_conv1 on the 4-wave engine; needs SPEECH_X_TM), 0 row-major, 8-wave engine */ \
  X(SPEECH_QKV_TM)   /* speech encoder: [1] tile-major q | k | v between the fused QKV GEMM (4-wave engine) and the attention (needs SPEECH_RP_LDS, SPEECH_X_TM), 0 row-major */ \
  X(SPEECH_X_TM)     /* speech encoder: [1] tile-major residual stream + LayerNorm fold (at create) */                  \
  X(XSIM_TM)         /* xsim: [1] tile-major normalised operands, 0 row-major */                                        \
  X(XSIM_LL)         /* xsim k >= 2: [1] per-row lists in LDS, 0 per-lane register lists */

enum Tune : int {
#define SMI_TUNE_ENUM(n) TUNE_##n,
  SMI_TUNE_LIST(SMI_TUNE_ENUM)
#undef SMI_TUNE_ENUM
  TUNE_COUNT
};

// 0 = unset (static zero-initialisation, no constructor order to get wrong); otherwise bit 40 | the value's 32 bits
extern std::atomic<int64_t> g_tune[TUNE_COUNT];

inline int tune(Tune t, int dflt) {
  const int64_t v = g_tune[t].load(std::memory_order_relaxed);
  return v ? (int)(int32_t)(uint32_t)(v & 0xffffffffll) : dflt;
}

inline const char* tune_name(int i) {
  static const char* const names[TUNE_COUNT] = {
#define SMI_TUNE_NAME(n) #n,
      SMI_TUNE_LIST(SMI_TUNE_NAME)
#undef SMI_TUNE_NAME
  };
  return i >= 0 && i < TUNE_COUNT ? names[i] : nullptr;
}

inline int tune_index(const char* name) {
  if (!name) return -1;
  if (!strncmp(name, "SMI_", 4)) name += 4;
  for (int i = 0; i < TUNE_COUNT; ++i)
    if (!strcmp(name, tune_name(i))) return i;
  return -1;
}

}  // namespace smi
