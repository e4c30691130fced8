"""Host-side mirror of the reference's SONAR speech encoder objects, backed by the HIP engine.

Reference interfaces mirrored (paths relative to facebookresearch/SONAR):
  * SonarSpeechEncoderConfig + archs `english` / `non_english`  sonar/models/sonar_speech/config.py:16-95
  * checkpoint key conversion                                   sonar/models/sonar_speech/handler.py:46-110
  * SonarSpeechEncoderModel.forward                             sonar/models/sonar_speech/model.py:59-77
  * WaveformToFbankConverter options                            sonar/inference_pipelines/speech.py:283-290
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass
from typing import Dict, List, Mapping, Optional, Sequence, Union

import torch

from . import _lib
from .text_encoder import PaddingMask, SequenceBatch, SonarEncoderOutput, _tensor_view


@dataclass
class SonarSpeechEncoderConfig:
    """Forward-relevant fields of SonarSpeechEncoderConfig with the w2v-BERT "600m" encoder
    sub-config flattened (config.py:61-77)."""

    model_dim: int = 1024
    num_encoder_layers: int = 24
    num_encoder_attn_heads: int = 16
    ffn_inner_dim: int = 4096
    depthwise_conv_kernel_size: int = 31
    num_fbank_channels: int = 80
    fbank_stride: int = 2
    max_seq_len: int = 1024          # pooler positional table (only position 0 is used)
    pad_idx: Optional[int] = 1
    bos_idx: int = 2
    num_decoder_layers: int = 3
    num_decoder_attn_heads: int = 16
    decoder_ffn_inner_dim: int = 4096
    max_frames: int = 4096           # w2v2 encoder max_seq_len (stacked frames)


def _english() -> SonarSpeechEncoderConfig:
    return SonarSpeechEncoderConfig()


def _non_english() -> SonarSpeechEncoderConfig:
    return SonarSpeechEncoderConfig(num_decoder_layers=6)


SPEECH_ENCODER_ARCHS = {"english": _english, "non_english": _non_english}


def get_speech_encoder_config(arch: str) -> SonarSpeechEncoderConfig:
    try:
        return SPEECH_ENCODER_ARCHS[arch]()
    except KeyError:
        raise ValueError(f"unknown sonar speech encoder arch {arch!r}; known: {sorted(SPEECH_ENCODER_ARCHS)}")


_FAIRSEQ1_SPEECH_KEY_MAP = [
    (r"^encoder\.w2v_model\.layer_norm\.", r"encoder_frontend.post_extract_layer_norm."),
    (r"^encoder\.w2v_model\.post_extract_proj\.", r"encoder_frontend.model_dim_proj."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.conv_module\.batch_norm\.", r"encoder.layers.\1.conv.batch_norm."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.conv_module\.depthwise_conv\.", r"encoder.layers.\1.conv.depthwise_conv."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.conv_module\.layer_norm\.", r"encoder.layers.\1.conv_layer_norm."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.conv_module\.pointwise_conv1\.", r"encoder.layers.\1.conv.pointwise_conv1."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.conv_module\.pointwise_conv2\.", r"encoder.layers.\1.conv.pointwise_conv2."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.ffn(1|2)\.layer_norm\.", r"encoder.layers.\1.ffn\2_layer_norm."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.ffn(1|2)\.w_1\.", r"encoder.layers.\1.ffn\2.inner_proj."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.ffn(1|2)\.w_2\.", r"encoder.layers.\1.ffn\2.output_proj."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.self_attn_layer_norm\.", r"encoder.layers.\1.self_attn_layer_norm."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.self_attn\.linear_q\.", r"encoder.layers.\1.self_attn.q_proj."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.self_attn\.linear_k\.", r"encoder.layers.\1.self_attn.k_proj."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.self_attn\.linear_v\.", r"encoder.layers.\1.self_attn.v_proj."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.self_attn\.linear_out\.", r"encoder.layers.\1.self_attn.output_proj."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.self_attn\.linear_pos\.", r"encoder.layers.\1.self_attn.sdpa.r_proj."),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.self_attn\.pos_bias_u", r"encoder.layers.\1.self_attn.sdpa.u_bias"),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.self_attn\.pos_bias_v", r"encoder.layers.\1.self_attn.sdpa.v_bias"),
    (r"^encoder\.w2v_model\.encoder\.layers\.([0-9]+)\.final_layer_norm\.", r"encoder.layers.\1.layer_norm."),
    # the redundant post-encoder LayerNorm moves to the model level (handler.py:102-108)
    (r"^encoder\.w2v_model\.encoder\.layer_norm\.", r"layer_norm."),
    (r"^decoder\.embed_tokens\.", r"encoder_pooler.decoder_frontend.embed."),
    (r"^decoder\.layers\.([0-9]+)\.self_attn_layer_norm\.", r"encoder_pooler.decoder.layers.\1.self_attn_layer_norm."),
    (r"^decoder\.layers\.([0-9]+)\.self_attn\.out_proj\.", r"encoder_pooler.decoder.layers.\1.self_attn.output_proj."),
    (r"^decoder\.layers\.([0-9]+)\.self_attn\.", r"encoder_pooler.decoder.layers.\1.self_attn."),
    (r"^decoder\.layers\.([0-9]+)\.encoder_attn_layer_norm\.", r"encoder_pooler.decoder.layers.\1.encoder_decoder_attn_layer_norm."),
    (r"^decoder\.layers\.([0-9]+)\.encoder_attn\.out_proj\.", r"encoder_pooler.decoder.layers.\1.encoder_decoder_attn.output_proj."),
    (r"^decoder\.layers\.([0-9]+)\.encoder_attn\.", r"encoder_pooler.decoder.layers.\1.encoder_decoder_attn."),
    (r"^decoder\.layers\.([0-9]+)\.fc1\.", r"encoder_pooler.decoder.layers.\1.ffn.inner_proj."),
    (r"^decoder\.layers\.([0-9]+)\.fc2\.", r"encoder_pooler.decoder.layers.\1.ffn.output_proj."),
    (r"^decoder\.layers\.([0-9]+)\.final_layer_norm\.", r"encoder_pooler.decoder.layers.\1.ffn_layer_norm."),
    (r"^decoder\.embed_out", r"encoder_pooler.projection_out.weight"),
]


def convert_sonar_speech_checkpoint(checkpoint: Mapping) -> Dict[str, torch.Tensor]:
    """Flat fairseq2-style state dict from a fairseq2 (`encoder_frontend.model_dim_proj.*` present)
    or fairseq1 speech-encoder checkpoint (handler.py:46-110)."""
    sd = checkpoint["model"] if "model" in checkpoint else checkpoint
    if any(k.startswith("encoder_frontend.model_dim_proj") for k in sd):
        return dict(sd)
    out: Dict[str, torch.Tensor] = {}
    for key, val in sd.items():
        if key == "encoder.w2v_model.mask_emb" or key.startswith("encoder.w2v_model.encoder.pos_conv."):
            continue
        new = key
        for pat, rep in _FAIRSEQ1_SPEECH_KEY_MAP:
            new, n = re.subn(pat, rep, new)
            if n:
                break
        out[new] = val
    return out


class SpeechEncoderEngine:
    """Owns one `smi_speech_encoder` handle."""

    def __init__(self, cfg: SonarSpeechEncoderConfig, state_dict: Mapping[str, torch.Tensor],
                 device: Union[str, torch.device] = "cuda:0", fp16_residual: bool = True):
        """fp16_residual: the conformer's residual stream in fp16, as the reference's `.half()` model on a GPU
        (speech.py:426-429); False keeps it in fp32."""
        if cfg.fbank_stride != 2:
            raise NotImplementedError("only 2-frame stacking is covered by the MI355X engine")
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the SONAR MI355X engine runs on a HIP device only (no CPU path)")
        self.lib = _lib.load()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.check(self.lib.smi_init(idx))
        d = cfg.model_dim
        ccfg = _lib.smi_speech_encoder_config(
            model_dim=d, num_layers=cfg.num_encoder_layers, num_heads=cfg.num_encoder_attn_heads,
            ffn_inner_dim=cfg.ffn_inner_dim, conv_kernel=cfg.depthwise_conv_kernel_size,
            num_mel_bins=cfg.num_fbank_channels, pooler_layers=cfg.num_decoder_layers,
            pooler_heads=cfg.num_decoder_attn_heads, pooler_ffn_dim=cfg.decoder_ffn_inner_dim,
            pooler_vocab=int(state_dict["encoder_pooler.decoder_frontend.embed.weight"].shape[0]),
            bos_idx=cfg.bos_idx, max_frames=cfg.max_frames, ln_eps=1e-5, bn_eps=1e-5,
            flags=_lib.SMI_ENC_FP16_RESIDUAL if fp16_residual else 0, reserved=0)
        keep: List[torch.Tensor] = []

        def tv(name: str, flat: bool = False) -> _lib.smi_tensor:
            if name not in state_dict:
                raise KeyError(f"checkpoint is missing {name}")
            t = state_dict[name]
            return _tensor_view(t.reshape(t.shape[0], -1) if flat and t.dim() > 2 else t, keep)

        layers = (_lib.smi_conformer_layer * cfg.num_encoder_layers)()
        for i in range(cfg.num_encoder_layers):
            p = f"encoder.layers.{i}."
            L = layers[i]
            for ffn in ("ffn1", "ffn2"):
                setattr(L, f"{ffn}_layer_norm_w", tv(p + f"{ffn}_layer_norm.weight"))
                setattr(L, f"{ffn}_layer_norm_b", tv(p + f"{ffn}_layer_norm.bias"))
                setattr(L, f"{ffn}_inner_w", tv(p + f"{ffn}.inner_proj.weight"))
                setattr(L, f"{ffn}_inner_b", tv(p + f"{ffn}.inner_proj.bias"))
                setattr(L, f"{ffn}_out_w", tv(p + f"{ffn}.output_proj.weight"))
                setattr(L, f"{ffn}_out_b", tv(p + f"{ffn}.output_proj.bias"))
            L.self_attn_layer_norm_w = tv(p + "self_attn_layer_norm.weight")
            L.self_attn_layer_norm_b = tv(p + "self_attn_layer_norm.bias")
            for a, b in (("q", "q_proj"), ("k", "k_proj"), ("v", "v_proj"), ("out", "output_proj")):
                setattr(L, f"{a}_w", tv(p + f"self_attn.{b}.weight"))
                setattr(L, f"{a}_b", tv(p + f"self_attn.{b}.bias"))
            L.r_proj_w = tv(p + "self_attn.sdpa.r_proj.weight")
            L.u_bias = tv(p + "self_attn.sdpa.u_bias")
            L.v_bias = tv(p + "self_attn.sdpa.v_bias")
            L.conv_layer_norm_w = tv(p + "conv_layer_norm.weight")
            L.conv_layer_norm_b = tv(p + "conv_layer_norm.bias")
            L.pointwise_conv1_w = tv(p + "conv.pointwise_conv1.weight", flat=True)
            L.depthwise_conv_w = tv(p + "conv.depthwise_conv.weight", flat=True)
            L.batch_norm_w = tv(p + "conv.batch_norm.weight")
            L.batch_norm_b = tv(p + "conv.batch_norm.bias")
            L.batch_norm_mean = tv(p + "conv.batch_norm.running_mean")
            L.batch_norm_var = tv(p + "conv.batch_norm.running_var")
            L.pointwise_conv2_w = tv(p + "conv.pointwise_conv2.weight", flat=True)
            L.layer_norm_w = tv(p + "layer_norm.weight")
            L.layer_norm_b = tv(p + "layer_norm.bias")
        pool = (_lib.smi_pooler_layer * cfg.num_decoder_layers)()
        for i in range(cfg.num_decoder_layers):
            p = f"encoder_pooler.decoder.layers.{i}."
            L = pool[i]
            L.self_v_w, L.self_v_b = tv(p + "self_attn.v_proj.weight"), tv(p + "self_attn.v_proj.bias")
            L.self_out_w, L.self_out_b = tv(p + "self_attn.output_proj.weight"), tv(p + "self_attn.output_proj.bias")
            L.self_attn_layer_norm_w = tv(p + "self_attn_layer_norm.weight")
            L.self_attn_layer_norm_b = tv(p + "self_attn_layer_norm.bias")
            for a, b in (("q", "q_proj"), ("k", "k_proj"), ("v", "v_proj"), ("out", "output_proj")):
                setattr(L, f"cross_{a}_w", tv(p + f"encoder_decoder_attn.{b}.weight"))
                setattr(L, f"cross_{a}_b", tv(p + f"encoder_decoder_attn.{b}.bias"))
            L.cross_layer_norm_w = tv(p + "encoder_decoder_attn_layer_norm.weight")
            L.cross_layer_norm_b = tv(p + "encoder_decoder_attn_layer_norm.bias")
            L.ffn_inner_w, L.ffn_inner_b = tv(p + "ffn.inner_proj.weight"), tv(p + "ffn.inner_proj.bias")
            L.ffn_out_w, L.ffn_out_b = tv(p + "ffn.output_proj.weight"), tv(p + "ffn.output_proj.bias")
            L.ffn_layer_norm_w = tv(p + "ffn_layer_norm.weight")
            L.ffn_layer_norm_b = tv(p + "ffn_layer_norm.bias")
        w = _lib.smi_speech_encoder_weights()
        w.post_extract_layer_norm_w = tv("encoder_frontend.post_extract_layer_norm.weight")
        w.post_extract_layer_norm_b = tv("encoder_frontend.post_extract_layer_norm.bias")
        w.model_dim_proj_w = tv("encoder_frontend.model_dim_proj.weight")
        w.model_dim_proj_b = tv("encoder_frontend.model_dim_proj.bias")
        w.layer_norm_w, w.layer_norm_b = tv("layer_norm.weight"), tv("layer_norm.bias")
        w.pooler_embed = tv("encoder_pooler.decoder_frontend.embed.weight")
        w.pooler_projection_out_w = tv("encoder_pooler.projection_out.weight")
        w.layers = C.cast(layers, C.POINTER(_lib.smi_conformer_layer))
        w.pooler = C.cast(pool, C.POINTER(_lib.smi_pooler_layer))
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_speech_encoder_create(C.byref(ccfg), C.byref(w), C.byref(handle)))
        self._handle = handle
        del keep

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                self.lib.smi_speech_encoder_destroy(h)
            except Exception:
                pass
            self._handle = None

    def forward(self, fbank: torch.Tensor, fbank_lens: Optional[Union[torch.Tensor, Sequence[int]]],
                out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
        """fbank: fp32 [N, T, 80] zero-padded with even T; fbank_lens: host ints or None."""
        if fbank.dim() != 3 or fbank.shape[2] != self.cfg.num_fbank_channels:
            raise ValueError(f"fbank must be [N, T, {self.cfg.num_fbank_channels}]")
        fb = fbank.to(self.device, torch.float32).contiguous()
        n, t, _ = fb.shape
        if t % 2:
            raise ValueError("the number of frames must be even (Collater pad_to_multiple=2, speech.py:444)")
        lens_arr = None
        if fbank_lens is not None:
            if isinstance(fbank_lens, torch.Tensor):
                fbank_lens = fbank_lens.detach().to("cpu", torch.int32).tolist()
            lens_arr = (C.c_int32 * n)(*[int(v) for v in fbank_lens])
        if out_dtype not in (torch.float16, torch.float32, torch.bfloat16):
            raise ValueError("out_dtype must be float16, bfloat16 or float32")
        eng_dtype = torch.float32 if out_dtype == torch.bfloat16 else out_dtype   # bf16: rounded once, on the way out
        out = torch.empty((n, self.cfg.model_dim), dtype=eng_dtype, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_speech_encoder_forward(
                self._handle, fb.data_ptr(), C.cast(lens_arr, C.c_void_p) if lens_arr is not None else None, n, t,
                out.data_ptr(), _lib.SMI_F32 if eng_dtype == torch.float32 else _lib.SMI_F16,
                _lib.current_stream_ptr()))
        return out if eng_dtype == out_dtype else _lib.cast(out, out_dtype)


def waveform_to_fbank(waveform: torch.Tensor, waveform_scale: float = 2.0 ** 15, standardize: bool = True) -> torch.Tensor:
    """WaveformToFbankConverter(num_mel_bins=80, waveform_scale=2**15, standardize=True)
    (speech.py:283-290) on the GPU.  waveform: 1-D fp32 on a HIP device, 16 kHz, range [-1, 1]."""
    if not waveform.is_cuda:
        raise RuntimeError("waveform_to_fbank runs on a HIP device only (no CPU path)")
    w = waveform.reshape(-1).to(torch.float32).contiguous()
    lib = _lib.load()
    frames = int(lib.smi_fbank_num_frames(w.numel()))
    out = torch.empty((frames, 80), dtype=torch.float32, device=w.device)
    if frames:
        with torch.cuda.device(w.device):
            _lib.check(lib.smi_fbank(w.data_ptr(), w.numel(), float(waveform_scale), 1 if standardize else 0,
                                     out.data_ptr(), _lib.current_stream_ptr()))
    return out


def fbank_batch_flat(cat: torch.Tensor, offsets: Sequence[int], waveform_scale: float = 2.0 ** 15,
                     standardize: bool = True, pad_to_multiple: int = 2):
    """`smi_fbank_batch` on clips that are already concatenated on the device: `cat` fp32 1-D, clip i =
    cat[offsets[i]:offsets[i+1]].  Returns (fbank fp32 [n, T, 80] zero-padded, frames per clip)."""
    if cat.device.type != "cuda":
        raise RuntimeError("the filterbank runs on a HIP device only (no CPU path)")
    n = len(offsets) - 1
    if n <= 0:
        raise ValueError("empty batch")
    lib = _lib.load()
    lens = [int(lib.smi_fbank_num_frames(int(offsets[i + 1] - offsets[i]))) for i in range(n)]
    t = max(max(lens), 1)
    t = (t + pad_to_multiple - 1) // pad_to_multiple * pad_to_multiple
    out = torch.empty((n, t, 80), dtype=torch.float32, device=cat.device)
    arr = (C.c_int64 * (n + 1))(*[int(o) for o in offsets])
    with torch.cuda.device(cat.device):
        _lib.check(lib.smi_fbank_batch(cat.data_ptr(), arr, n, float(waveform_scale), 1 if standardize else 0,
                                       out.data_ptr(), t, _lib.current_stream_ptr()))
    return out, lens


def waveforms_to_fbank_batch(waveforms: Sequence[torch.Tensor], waveform_scale: float = 2.0 ** 15,
                             standardize: bool = True, pad_to_multiple: int = 2):
    """The filterbank of a whole batch in one launch, collated as the reference's
    `Collater(pad_value=0, pad_to_multiple=2)` does (speech.py:444).
    waveforms: 1-D fp32 tensors on one HIP device.  Returns (fbank fp32 [n, T, 80] zero-padded, frames per clip)."""
    if not waveforms:
        raise ValueError("empty batch")
    dev = waveforms[0].device
    if dev.type != "cuda":
        raise RuntimeError("waveforms_to_fbank_batch runs on a HIP device only (no CPU path)")
    flat = [w.reshape(-1).to(dev, torch.float32) for w in waveforms]
    offs = [0]
    for w in flat:
        offs.append(offs[-1] + w.numel())
    cat = torch.cat(flat) if len(flat) > 1 else flat[0].contiguous()
    return fbank_batch_flat(cat, offs, waveform_scale, standardize, pad_to_multiple)


class SonarSpeechEncoderModel:
    """Drop-in for the object SpeechToEmbeddingModelPipeline calls as `model(batch)`
    (speech.py:452): SequenceBatch of fbank features -> SonarEncoderOutput."""

    def __init__(self, cfg: SonarSpeechEncoderConfig, state_dict: Mapping[str, torch.Tensor],
                 device: Union[str, torch.device] = "cuda:0", dtype: torch.dtype = torch.float16,
                 fp16_residual: Optional[bool] = None):
        """dtype: dtype of the returned embeddings and (as in the reference) of the residual stream, unless
        `fp16_residual` says otherwise.  A bf16 model keeps the fp32 residual stream: its activations have fp32 range,
        an fp16 stream does not (round 4, as the text encoder)."""
        self.config = cfg
        self.model_dim = cfg.model_dim
        self.dtype = dtype
        self.engine = SpeechEncoderEngine(cfg, state_dict, device,
                                          dtype == torch.float16 if fp16_residual is None else fp16_residual)
        self.device = self.engine.device

    def eval(self):
        return self

    def __call__(self, batch: SequenceBatch) -> SonarEncoderOutput:
        return self.forward(batch)

    @torch.inference_mode()
    def forward(self, batch: SequenceBatch) -> SonarEncoderOutput:
        lens = batch.padding_mask.seq_lens if batch.padding_mask is not None else None
        emb = self.engine.forward(batch.seqs, lens, self.dtype)
        return SonarEncoderOutput(encoded_seqs=None, sentence_embeddings=emb, padding_mask=batch.padding_mask)


def load_sonar_speech_encoder(checkpoint: Union[str, Mapping], arch: str = "english",
                              device: Union[str, torch.device] = "cuda:0", dtype: torch.dtype = torch.float16,
                              config: Optional[SonarSpeechEncoderConfig] = None,
                              load_stats: Optional[dict] = None) -> SonarSpeechEncoderModel:
    """Card name / checkpoint file (through the packed cache) / in-memory dict -> speech encoder model."""
    cfg_arch = arch
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        from .cards import resolve_checkpoint
        from .packed_cache import load_converted

        path, cfg_arch = resolve_checkpoint(checkpoint, arch)
        sd = load_converted(path, convert_sonar_speech_checkpoint, "speech_encoder", load_stats)
    else:
        sd = convert_sonar_speech_checkpoint(checkpoint)
    cfg = config or get_speech_encoder_config(cfg_arch)
    return SonarSpeechEncoderModel(cfg, sd, device, dtype)
