"""ctypes binding of the C ABI declared in include/sonar_mi355.h.

The shared library is the product path.  There is no CPU fallback: if the
library has not been built, `load()` raises, and every compute entry point of
the library itself fails with SMI_ERR_NO_DEVICE when no MI355X is visible.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from pathlib import Path
from typing import Optional

# SMI_LIB: development override -- load a variant build of the library (tools/README.md) instead of the in-tree one
LIB_PATH = Path(os.environ["SMI_LIB"]).resolve() if os.environ.get("SMI_LIB") else \
    Path(__file__).resolve().parent / "lib" / "libsonar_mi355.so"

SMI_OK = 0
SMI_F32, SMI_F16, SMI_BF16 = 0, 1, 2
SMI_POOL = {"mean": 0, "max": 1, "last": 2, "attention": 3}
SMI_MARGIN = {"ratio": 0, "distance": 1, "cosine": 2}
SMI_GEMM_IN_TM, SMI_GEMM_OUT_TM = 1 << 12, 1 << 13
SMI_ENC_FP16_RESIDUAL = 1
SMI_ENC_NORMALIZE_BEFORE = 2
SMI_ENC_LAYERNORM_EMBEDDING = 4
SMI_ENC_NO_POSITIONS = 8
PROF_SLOTS = ["embed", "layernorm", "gemm_qkv", "attention", "gemm_out", "gemm_ffn1", "gemm_ffn2", "ln_pool"]
STATUS_NAMES = {
    0: "SMI_OK",
    -1: "SMI_ERR_INVALID_ARG",
    -2: "SMI_ERR_UNSUPPORTED",
    -3: "SMI_ERR_NO_DEVICE",
    -4: "SMI_ERR_OOM",
    -5: "SMI_ERR_HIP",
}


class SmiError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class smi_tensor(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("dtype", C.c_int32),
        ("on_device", C.c_int32),
        ("numel", C.c_int64),
    ]


class smi_text_encoder_config(C.Structure):
    _fields_ = [
        ("model_dim", C.c_int32),
        ("num_layers", C.c_int32),
        ("num_heads", C.c_int32),
        ("ffn_inner_dim", C.c_int32),
        ("vocab_size", C.c_int64),
        ("max_seq_len", C.c_int32),
        ("pos_offset", C.c_int32),
        ("embed_scale", C.c_float),
        ("ln_eps", C.c_float),
        ("pooling", C.c_int32),
        ("flags", C.c_int32),
        ("embedding_dim", C.c_int32),
        ("pooler_layers", C.c_int32),
        ("pooler_heads", C.c_int32),
        ("pooler_ffn_dim", C.c_int32),
    ]


_LAYER_FIELDS = [
    "self_attn_layer_norm_w", "self_attn_layer_norm_b",
    "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "out_w", "out_b",
    "ffn_layer_norm_w", "ffn_layer_norm_b",
    "ffn_inner_w", "ffn_inner_b", "ffn_out_w", "ffn_out_b",
]


class smi_text_encoder_layer(C.Structure):
    _fields_ = [(n, smi_tensor) for n in _LAYER_FIELDS]


_TEXT_POOLER_LAYER_FIELDS = [
    "self_attn_layer_norm_w", "self_attn_layer_norm_b", "self_v_w", "self_v_b", "self_out_w", "self_out_b",
    "cross_layer_norm_w", "cross_layer_norm_b", "cross_q_w", "cross_q_b", "cross_k_w", "cross_k_b", "cross_v_w", "cross_v_b",
    "cross_out_w", "cross_out_b", "ffn_layer_norm_w", "ffn_layer_norm_b", "ffn_inner_w", "ffn_inner_b", "ffn_out_w", "ffn_out_b",
]


class smi_text_pooler_layer(C.Structure):
    _fields_ = [(n, smi_tensor) for n in _TEXT_POOLER_LAYER_FIELDS]


class smi_text_encoder_weights(C.Structure):
    _fields_ = [
        ("embed", smi_tensor),
        ("pos_table", smi_tensor),
        ("final_layer_norm_w", smi_tensor),
        ("final_layer_norm_b", smi_tensor),
        ("layers", C.POINTER(smi_text_encoder_layer)),
        ("encoder_layer_norm_w", smi_tensor),
        ("encoder_layer_norm_b", smi_tensor),
        ("embed_layer_norm_w", smi_tensor),
        ("embed_layer_norm_b", smi_tensor),
        ("pooler_query", smi_tensor),
        ("pooler", C.POINTER(smi_text_pooler_layer)),
        ("pooler_layer_norm_w", smi_tensor),
        ("pooler_layer_norm_b", smi_tensor),
        ("pooler_proj_w", smi_tensor),
        ("pooler_proj_b", smi_tensor),
    ]


class smi_text_decoder_config(C.Structure):
    _fields_ = [
        ("model_dim", C.c_int32),
        ("num_layers", C.c_int32),
        ("num_heads", C.c_int32),
        ("ffn_inner_dim", C.c_int32),
        ("vocab_size", C.c_int64),
        ("max_seq_len", C.c_int32),
        ("pos_offset", C.c_int32),
        ("input_dim", C.c_int32),
        ("embed_scale", C.c_float),
        ("ln_eps", C.c_float),
        ("pad_idx", C.c_int32),
        ("unk_idx", C.c_int32),
        ("bos_idx", C.c_int32),
        ("eos_idx", C.c_int32),
    ]


_DEC_LAYER_FIELDS = [
    "self_attn_layer_norm_w", "self_attn_layer_norm_b",
    "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "out_w", "out_b",
    "cross_v_w", "cross_v_b", "cross_out_w", "cross_out_b",
    "ffn_layer_norm_w", "ffn_layer_norm_b",
    "ffn_inner_w", "ffn_inner_b", "ffn_out_w", "ffn_out_b",
]


class smi_text_decoder_layer(C.Structure):
    _fields_ = [(n, smi_tensor) for n in _DEC_LAYER_FIELDS]


class smi_text_decoder_weights(C.Structure):
    _fields_ = [
        ("embed", smi_tensor),
        ("pos_table", smi_tensor),
        ("final_layer_norm_w", smi_tensor),
        ("final_layer_norm_b", smi_tensor),
        ("layers", C.POINTER(smi_text_decoder_layer)),
    ]


class smi_beam_search_params(C.Structure):
    _fields_ = [
        ("beam_size", C.c_int32),
        ("max_seq_len", C.c_int32),
        ("min_seq_len", C.c_int32),
        ("normalize_scores", C.c_int32),
        ("len_penalty", C.c_float),
        ("unk_penalty", C.c_float),
        ("temperature", C.c_float),
        ("reserved", C.c_int32),
    ]


SMI_SAMPLER_TOP_K, SMI_SAMPLER_TOP_P = 0, 1


class smi_sampling_params(C.Structure):
    _fields_ = [
        ("sampler", C.c_int32),
        ("top_k", C.c_int32),
        ("top_p", C.c_float),
        ("temperature", C.c_float),
        ("max_seq_len", C.c_int32),
        ("min_seq_len", C.c_int32),
        ("normalize_scores", C.c_int32),
        ("len_penalty", C.c_float),
        ("seed", C.c_uint64),
        ("unk_penalty", C.c_float),
    ]


class smi_speech_encoder_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "model_dim", "num_layers", "num_heads", "ffn_inner_dim", "conv_kernel", "num_mel_bins",
        "pooler_layers", "pooler_heads", "pooler_ffn_dim", "pooler_vocab", "bos_idx", "max_frames")] + [
        ("ln_eps", C.c_float), ("bn_eps", C.c_float), ("flags", C.c_int32), ("reserved", C.c_int32)]


_CONF_LAYER_FIELDS = [
    "ffn1_layer_norm_w", "ffn1_layer_norm_b", "ffn1_inner_w", "ffn1_inner_b", "ffn1_out_w", "ffn1_out_b",
    "self_attn_layer_norm_w", "self_attn_layer_norm_b",
    "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "out_w", "out_b",
    "r_proj_w", "u_bias", "v_bias",
    "conv_layer_norm_w", "conv_layer_norm_b", "pointwise_conv1_w", "depthwise_conv_w",
    "batch_norm_w", "batch_norm_b", "batch_norm_mean", "batch_norm_var", "pointwise_conv2_w",
    "ffn2_layer_norm_w", "ffn2_layer_norm_b", "ffn2_inner_w", "ffn2_inner_b", "ffn2_out_w", "ffn2_out_b",
    "layer_norm_w", "layer_norm_b",
]
_POOL_LAYER_FIELDS = [
    "self_v_w", "self_v_b", "self_out_w", "self_out_b", "self_attn_layer_norm_w", "self_attn_layer_norm_b",
    "cross_q_w", "cross_q_b", "cross_k_w", "cross_k_b", "cross_v_w", "cross_v_b", "cross_out_w", "cross_out_b",
    "cross_layer_norm_w", "cross_layer_norm_b",
    "ffn_inner_w", "ffn_inner_b", "ffn_out_w", "ffn_out_b", "ffn_layer_norm_w", "ffn_layer_norm_b",
]


class smi_conformer_layer(C.Structure):
    _fields_ = [(n, smi_tensor) for n in _CONF_LAYER_FIELDS]


class smi_pooler_layer(C.Structure):
    _fields_ = [(n, smi_tensor) for n in _POOL_LAYER_FIELDS]


class smi_speech_encoder_weights(C.Structure):
    _fields_ = [(n, smi_tensor) for n in (
        "post_extract_layer_norm_w", "post_extract_layer_norm_b", "model_dim_proj_w", "model_dim_proj_b",
        "layer_norm_w", "layer_norm_b", "pooler_embed", "pooler_projection_out_w")] + [
        ("layers", C.POINTER(smi_conformer_layer)), ("pooler", C.POINTER(smi_pooler_layer))]


class smi_mlp_head_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_dim", "n_layers", "hidden_act", "out_act")]


class smi_mlp_head_layer(C.Structure):
    _fields_ = [("w", smi_tensor), ("b", smi_tensor), ("out_dim", C.c_int32), ("reserved", C.c_int32)]


ABI_VERSION = 6  # SMI_ABI_VERSION of include/sonar_mi355.h

# every symbol include/sonar_mi355.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    "smi_version": (C.c_char_p, []),
    "smi_abi_version": (C.c_int, []),
    "smi_last_error": (C.c_char_p, []),
    "smi_tuning_set": (C.c_int, [C.c_char_p, _i32]),
    "smi_tuning_unset": (C.c_int, [C.c_char_p]),
    "smi_tuning_get": (C.c_int, [C.c_char_p, C.POINTER(_i32), C.POINTER(_i32)]),
    "smi_tuning_name": (C.c_char_p, [_i32]),
    "smi_init": (C.c_int, [C.c_int]),
    "smi_device_count": (C.c_int, []),
    "smi_text_encoder_create": (C.c_int, [C.POINTER(smi_text_encoder_config),
                                          C.POINTER(smi_text_encoder_weights), _i64,
                                          C.POINTER(_vp)]),
    "smi_text_encoder_destroy": (None, [_vp]),
    "smi_text_encoder_forward": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "smi_text_encoder_status": (C.c_int, [_vp, _vp]),
    "smi_text_encoder_device_bytes": (_i64, [_vp]),
    "smi_text_encoder_set_profiling": (C.c_int, [_vp, _i32]),
    "smi_text_encoder_read_profile": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "smi_text_decoder_create": (C.c_int, [C.POINTER(smi_text_decoder_config),
                                          C.POINTER(smi_text_decoder_weights), C.POINTER(_vp)]),
    "smi_text_decoder_destroy": (None, [_vp]),
    "smi_text_decoder_logits": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "smi_text_decoder_generate": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(_i64), _i32,
                                            C.POINTER(smi_beam_search_params), _vp, _vp, _vp, _vp]),
    "smi_text_decoder_last_margins": (C.c_int, [_vp, _vp, _i32, _vp]),
    "smi_text_decoder_set_chains": (C.c_int, [_vp, _i32]),
    "smi_text_decoder_set_beam_logits_dtype": (C.c_int, [_vp, _i32]),
    "smi_text_decoder_set_slab_dtype": (C.c_int, [_vp, _i32]),
    "smi_text_decoder_sample": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(_i64), _i32,
                                          C.POINTER(smi_sampling_params), _vp, _vp, _vp, _vp]),
    "smi_sample_rows": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _i32, _i32, _i32, _f32, _vp,
                                  _vp, _vp, _vp, _vp, _vp]),
    "smi_speech_encoder_create": (C.c_int, [C.POINTER(smi_speech_encoder_config),
                                            C.POINTER(smi_speech_encoder_weights), C.POINTER(_vp)]),
    "smi_speech_encoder_destroy": (None, [_vp]),
    "smi_speech_encoder_forward": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "smi_fbank_num_frames": (_i64, [_i64]),
    "smi_fbank": (C.c_int, [_vp, _i64, _f32, _i32, _vp, _vp]),
    "smi_fbank_batch": (C.c_int, [_vp, C.POINTER(_i64), _i32, _f32, _i32, _vp, _i64, _vp]),
    "smi_xsim_padded_rows": (_i64, [_i64]),
    "smi_xsim_normalize": (C.c_int, [_vp, _i32, _i64, _i32, _vp, _vp]),
    "smi_xsim_workspace_bytes": (_i64, [_i64, _i64, _i32, _i32]),
    "smi_xsim_topk": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _i64, _vp]),
    "smi_xsim_merge_topk": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "smi_xsim_margin_select": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "smi_gemm_tn": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "smi_gemm_tn_tile_stats": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, C.c_float, _i32, _vp, _vp, _vp]),
    "smi_gemm_tn_splitk": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "smi_mlp_head_create": (C.c_int, [C.POINTER(smi_mlp_head_config), C.POINTER(smi_mlp_head_layer), C.POINTER(_vp)]),
    "smi_mlp_head_destroy": (None, [_vp]),
    "smi_head_featurize": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "smi_mlp_head_forward": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "smi_host_token_lengths": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, C.POINTER(_i64)]),
    "smi_host_dynamic_bucket": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "smi_host_collate_nllb": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _i32, _i32, _i64, _vp, _i32, _i32]),
    "smi_host_wav_info": (C.c_int, [_vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64)]),
    "smi_host_wav_decode": (C.c_int, [_vp, _i64, _vp, _i64, _i32]),
    "smi_host_audio_info": (C.c_int, [_vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64)]),
    "smi_host_audio_decode": (C.c_int, [_vp, _i64, _vp, _i64, _i32]),
    "smi_pack_tile_major": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "smi_cast": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _vp]),
    "smi_layernorm": (C.c_int, [_vp, _vp, _vp, _f32, _vp, _i32, _i32, _i32, _vp]),
    "smi_attention": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "smi_relpos_attention": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libsonar_mi355.so (torch is imported first so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m sonar_amd.build` "
            "(needs hipcc, gfx950). There is no CPU fallback for the SONAR hot path."
        )
    import torch  # noqa: F401  (loads libamdhip64 the way torch wants it)

    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.smi_abi_version() != ABI_VERSION:  # the structs carry no size field: refuse a library of another revision
        raise RuntimeError(f"{LIB_PATH} has ABI revision {lib.smi_abi_version()}, this binding speaks {ABI_VERSION}: "
                           "rebuild with `python -m sonar_amd.build --force`")
    _forward_env_switches(lib)   # BEFORE the handle is cached: a malformed SMI_<NAME> raises on every load(), not only the first
    _lib = lib
    return lib


def tuning_names() -> list:
    """Names of the library's tuning switches (sonar_amd/csrc/tuning.hpp), without the SMI_ prefix."""
    lib, out, i = load(), [], 0
    while True:
        nm = lib.smi_tuning_name(i)
        if nm is None:
            return out
        out.append(nm.decode())
        i += 1


def _forward_env_switches(lib: C.CDLL) -> None:
    """The library itself never reads the environment (include/sonar_mi355.h, tuning registry).  For the measurement tooling
    (`env SMI_LONE=0 python tools/...`) this binding forwards `SMI_<NAME>=<int>` variables ONCE, when the library is loaded;
    later changes of os.environ have no effect -- use `tuning()` / `set_tuning()`."""
    i = 0
    while True:
        nm = lib.smi_tuning_name(i)
        if nm is None:
            break
        i += 1
        v = os.environ.get("SMI_" + nm.decode())
        if v is None or v == "":
            continue
        try:
            iv = int(v)
        except ValueError:
            raise RuntimeError(f"SMI_{nm.decode()}={v!r}: tuning switches are integers") from None
        _check_i32(nm.decode(), iv)
        if lib.smi_tuning_set(nm, iv) != SMI_OK:
            raise RuntimeError(lib.smi_last_error().decode("utf-8", "replace"))


def _check_i32(name: str, value: int) -> None:
    """smi_tuning_set takes an int32: refuse what ctypes would silently truncate."""
    if not -2 ** 31 <= value < 2 ** 31:
        raise ValueError(f"tuning switch {name}={value} does not fit an int32")


def set_tuning(**switches) -> None:
    """set_tuning(LONE=0, DEC_KS_OUT=2): process-wide tuning switches (None unsets one)."""
    lib = load()
    for name, value in switches.items():
        if value is not None:
            _check_i32(name, int(value))
        rc = lib.smi_tuning_unset(name.encode()) if value is None else lib.smi_tuning_set(name.encode(), int(value))
        check(rc)


@contextlib.contextmanager
def tuning(**switches):
    """with _lib.tuning(DEC_KS_OUT=2, G2_AUTO_MIN=1000000): ...  -- sets the switches, restores their previous state on exit."""
    lib = load()
    saved = {}
    for name in switches:
        v, st = _i32(0), _i32(0)
        check(lib.smi_tuning_get(name.encode(), C.byref(v), C.byref(st)))
        saved[name] = v.value if st.value else None
    set_tuning(**switches)
    try:
        yield
    finally:
        set_tuning(**saved)


def check(status: int) -> None:
    if status != SMI_OK:
        raise SmiError(status, load().smi_last_error().decode("utf-8", "replace"))


_TORCH_DTYPES = None


def smi_dtype_of(dtype) -> int:
    """torch dtype -> smi_dtype of the boundary (fp32 / fp16 / bf16)."""
    global _TORCH_DTYPES
    if _TORCH_DTYPES is None:
        import torch

        _TORCH_DTYPES = {torch.float32: SMI_F32, torch.float16: SMI_F16, torch.bfloat16: SMI_BF16}
    try:
        return _TORCH_DTYPES[dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {dtype}: the engine's boundary speaks float32, float16 and bfloat16") from None


def cast(t, dtype):
    """Device tensor -> a new contiguous device tensor of `dtype`, converted by the engine (smi_cast) on the current
    stream: the bf16 side of the pipelines' `dtype=` argument.  fp32 / fp16 / bf16 only."""
    import torch

    if t.dtype == dtype:
        return t
    if not t.is_cuda:
        raise ValueError("smi_cast converts device tensors")
    src = t.contiguous()
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    with torch.cuda.device(src.device):
        check(load().smi_cast(src.data_ptr(), smi_dtype_of(src.dtype), out.data_ptr(), smi_dtype_of(dtype), src.numel(),
                              current_stream_ptr()))
    return out


def current_stream_ptr() -> int:
    import torch

    return int(torch.cuda.current_stream().cuda_stream)
