"""Host-side mirror of the reference's SONAR text encoder model objects, backed
by the HIP engine (libsonar_mi355.so) instead of fairseq2 modules.

Reference interfaces mirrored (paths relative to facebookresearch/SONAR):
  * SonarTextEncoderConfig + archs `basic` / `small`  sonar/models/sonar_text/config.py:14-127
  * SonarEncoderOutput / SonarEncoderModel           sonar/models/encoder_model.py:17-67
  * SonarTextTransformerEncoderModel.forward         sonar/models/sonar_text/model.py:130-143
  * checkpoint key conversion                        sonar/models/sonar_text/handler.py:52-94

PyTorch is used only for device memory, streams and tensors at the boundary;
all arithmetic runs in the hand-written gfx950 kernels.
"""
from __future__ import annotations

import contextlib

import ctypes as C
import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional, Sequence, Union

import torch

from . import _lib


# --------------------------------------------------------------------- config
@dataclass
class VocabularyInfo:
    size: int
    unk_idx: Optional[int] = 1
    bos_idx: Optional[int] = 2
    eos_idx: Optional[int] = 3
    pad_idx: Optional[int] = 1


@dataclass
class SonarTextEncoderConfig:
    """Same fields as the reference dataclass (config.py:14-85)."""

    model_dim: int = 1024
    max_seq_len: int = 512
    vocab_info: VocabularyInfo = field(default_factory=lambda: VocabularyInfo(size=256206))
    num_encoder_layers: int = 24
    num_decoder_layers: int = 24
    num_encoder_attn_heads: int = 16
    num_decoder_attn_heads: int = 16
    ffn_inner_dim: int = 1024 * 8
    pooling: str = "mean"
    embedding_dim: Optional[int] = None
    decoder_ffn_inner_dim: Optional[int] = None
    activation_fn: str = "ReLU"
    layernorm_embedding: bool = False
    no_scale_embedding: bool = False
    no_token_positional_embeddings: bool = False
    learned_pos: bool = False
    emb_dropout_p: float = 0.1
    attention_dropout_p: float = 0.1
    activation_dropout_p: float = 0.1
    normalize_before: bool = False
    _from_fairseq: bool = False

    @property
    def pos_offset(self) -> int:
        # SinusoidalPositionEncoder(_legacy_pad_idx=pad_idx), factory.py:88-92
        return (self.vocab_info.pad_idx or 0) + 1

    @property
    def model_max_seq_len(self) -> int:
        # factory.py:56-59: max_seq_len += pad_idx + 1 for fairseq-trained models
        return self.max_seq_len + (self.pos_offset if self._from_fairseq else 0)


def _basic() -> SonarTextEncoderConfig:
    return SonarTextEncoderConfig(_from_fairseq=True)


def _small(vocab_size: int = 32005, depth: int = 6, hidden_dim: int = 1024 * 4) -> SonarTextEncoderConfig:
    c = _basic()
    c.vocab_info = VocabularyInfo(size=vocab_size)
    c.num_encoder_layers = depth
    c.num_decoder_layers = depth
    c.ffn_inner_dim = hidden_dim
    return c


TEXT_ENCODER_ARCHS = {"basic": _basic, "small": _small}


def get_text_encoder_config(arch: str) -> SonarTextEncoderConfig:
    try:
        return TEXT_ENCODER_ARCHS[arch]()
    except KeyError:
        raise ValueError(f"unknown sonar text encoder arch {arch!r}; known: {sorted(TEXT_ENCODER_ARCHS)}")


def check_supported(cfg: SonarTextEncoderConfig) -> None:
    """What the library runs: the released models' shapes on the MFMA engines, every other shape / option of the
    reference's factory (any model_dim divisible by the head count with head_dim <= 256, attention pooling with
    embedding_dim != model_dim, normalize_before, layernorm_embedding, learned / no positions) on its
    generic-dimension kernels (csrc/flex.hip) -- chosen inside smi_text_encoder_create.  Not covered: activations
    other than ReLU."""
    bad = []
    if cfg.activation_fn != "ReLU":
        bad.append(f"activation_fn={cfg.activation_fn}")
    if cfg.pooling not in ("mean", "max", "last", "attention"):
        bad.append(f"pooling={cfg.pooling}")
    # (embedding_dim is read by the attention pooler only; with mean / max / last pooling the reference's factory never
    #  looks at it -- factory.py:108-112 -- and neither does the engine)
    if cfg.model_dim % max(cfg.num_encoder_attn_heads, 1) or cfg.model_dim // max(cfg.num_encoder_attn_heads, 1) > 256:
        bad.append("model_dim must be a multiple of the head count with head_dim <= 256")
    if bad:
        raise NotImplementedError("not covered by the MI355X engine: " + ", ".join(bad))


# ------------------------------------------------------------------ positions
def sinusoidal_table(num_positions: int, dim: int) -> torch.Tensor:
    """fp32 [num_positions, dim]; row p encodes absolute position p in the fairseq
    half-split layout [sin | cos] (SinusoidalPositionEncoder)."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000.0) / (half - 1)))
    ang = torch.arange(num_positions, dtype=torch.float32).unsqueeze(1) * freq.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(num_positions, 1)], dim=1)
    return tab.contiguous()


# ----------------------------------------------------------------- checkpoints
_FAIRSEQ1_KEY_MAP = [
    (r"^layers\.([0-9]+)\.self_attn\.q_proj\.", r"encoder.layers.\1.self_attn.q_proj."),
    (r"^layers\.([0-9]+)\.self_attn\.v_proj\.", r"encoder.layers.\1.self_attn.v_proj."),
    (r"^layers\.([0-9]+)\.self_attn\.k_proj\.", r"encoder.layers.\1.self_attn.k_proj."),
    (r"^layers\.([0-9]+)\.self_attn\.out_proj\.", r"encoder.layers.\1.self_attn.output_proj."),
    (r"^layers\.([0-9]+)\.self_attn_layer_norm\.", r"encoder.layers.\1.self_attn_layer_norm."),
    (r"^layers\.([0-9]+)\.fc1\.", r"encoder.layers.\1.ffn.inner_proj."),
    (r"^layers\.([0-9]+)\.fc2\.", r"encoder.layers.\1.ffn.output_proj."),
    (r"^layers\.([0-9]+)\.final_layer_norm\.", r"encoder.layers.\1.ffn_layer_norm."),
    (r"^embed_tokens\.", r"encoder_frontend.embed."),
]


def convert_sonar_text_encoder_checkpoint(checkpoint: Mapping) -> Dict[str, torch.Tensor]:
    """Return a flat fairseq2-style state dict from either checkpoint layout the
    reference accepts (handler.py:52-94): a fairseq2 `{"model": {...}}` dict is
    passed through; a fairseq1 `{"state_dict": {...}}` dict gets its keys renamed
    and the four control-token embedding rows permuted
    (BOS, PAD, EOS, UNK) -> (PAD, UNK, BOS, EOS)  (handler.py:86-92)."""
    if "model" in checkpoint and "encoder_frontend.embed.weight" in checkpoint["model"]:
        return dict(checkpoint["model"])
    if "state_dict" not in checkpoint:
        # already a flat fairseq2-style dict
        if "encoder_frontend.embed.weight" in checkpoint:
            return dict(checkpoint)
        raise ValueError("unrecognised SONAR text encoder checkpoint layout")
    out: Dict[str, torch.Tensor] = {}
    for key, val in checkpoint["state_dict"].items():
        if key in ("version", "embed_positions._float_tensor"):
            continue
        new = key
        for pat, rep in _FAIRSEQ1_KEY_MAP:
            new, n = re.subn(pat, rep, new)
            if n:
                break
        out[new] = val
    emb = out["encoder_frontend.embed.weight"].clone()
    emb[[0, 1, 2, 3]] = emb[[1, 3, 0, 2]]
    out["encoder_frontend.embed.weight"] = emb
    return out


# --------------------------------------------------------------------- engine
def _tensor_view(t: torch.Tensor, keep: List[torch.Tensor]) -> _lib.smi_tensor:
    if t.dtype not in (torch.float32, torch.float16):
        t = t.float()
    t = t.detach().contiguous()
    keep.append(t)
    return _lib.smi_tensor(
        data=t.data_ptr(),
        dtype=_lib.SMI_F32 if t.dtype == torch.float32 else _lib.SMI_F16,
        on_device=1 if t.is_cuda else 0,
        numel=t.numel(),
    )


class TextEncoderEngine:
    """Owns one `smi_text_encoder` handle (packed fp16 weights + workspace in HBM)."""

    def __init__(self, cfg: SonarTextEncoderConfig, state_dict: Mapping[str, torch.Tensor],
                 device: Union[str, torch.device] = "cuda:0", max_tokens_hint: int = 0,
                 fp16_residual: bool = False):
        """fp16_residual: keep the residual stream in fp16 as the reference's fp16 model does
        (SMI_ENC_FP16_RESIDUAL); default is an fp32 stream (more accurate, 2x the residual traffic)."""
        check_supported(cfg)
        self.cfg = cfg
        self.fp16_residual = bool(fp16_residual)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the SONAR MI355X engine runs on a HIP device only (no CPU path)")
        self.lib = _lib.load()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.check(self.lib.smi_init(idx))
        d = cfg.model_dim
        attn_pool = cfg.pooling == "attention"
        self.embedding_dim = (cfg.embedding_dim or d) if attn_pool else d
        flags = _lib.SMI_ENC_FP16_RESIDUAL if fp16_residual else 0
        if cfg.normalize_before:
            flags |= _lib.SMI_ENC_NORMALIZE_BEFORE
        if cfg.layernorm_embedding:
            flags |= _lib.SMI_ENC_LAYERNORM_EMBEDDING
        if cfg.no_token_positional_embeddings:
            flags |= _lib.SMI_ENC_NO_POSITIONS
        # LearnedPositionEncoder indexes its table from 0; the sinusoidal encoder keeps fairseq's pad_idx + 1 offset
        pos_offset = 0 if cfg.learned_pos else cfg.pos_offset
        ccfg = _lib.smi_text_encoder_config(
            model_dim=d, num_layers=cfg.num_encoder_layers, num_heads=cfg.num_encoder_attn_heads,
            ffn_inner_dim=cfg.ffn_inner_dim, vocab_size=cfg.vocab_info.size,
            max_seq_len=cfg.model_max_seq_len, pos_offset=pos_offset,
            embed_scale=1.0 if cfg.no_scale_embedding else math.sqrt(d), ln_eps=1e-5,
            pooling=_lib.SMI_POOL[cfg.pooling], flags=flags,
            embedding_dim=self.embedding_dim if attn_pool else 0,
            pooler_layers=cfg.num_decoder_layers if attn_pool else 0,
            pooler_heads=cfg.num_decoder_attn_heads if attn_pool else 0,
            pooler_ffn_dim=(cfg.decoder_ffn_inner_dim or cfg.ffn_inner_dim) if attn_pool else 0)
        keep: List[torch.Tensor] = []
        sd = state_dict

        def tv(name: str) -> _lib.smi_tensor:
            if name not in sd:
                raise KeyError(f"checkpoint is missing {name}")
            return _tensor_view(sd[name], keep)

        layers = (_lib.smi_text_encoder_layer * max(cfg.num_encoder_layers, 1))()
        for i in range(cfg.num_encoder_layers):
            p = f"encoder.layers.{i}."
            L = layers[i]
            L.self_attn_layer_norm_w = tv(p + "self_attn_layer_norm.weight")
            L.self_attn_layer_norm_b = tv(p + "self_attn_layer_norm.bias")
            L.q_w, L.q_b = tv(p + "self_attn.q_proj.weight"), tv(p + "self_attn.q_proj.bias")
            L.k_w, L.k_b = tv(p + "self_attn.k_proj.weight"), tv(p + "self_attn.k_proj.bias")
            L.v_w, L.v_b = tv(p + "self_attn.v_proj.weight"), tv(p + "self_attn.v_proj.bias")
            L.out_w, L.out_b = tv(p + "self_attn.output_proj.weight"), tv(p + "self_attn.output_proj.bias")
            L.ffn_layer_norm_w = tv(p + "ffn_layer_norm.weight")
            L.ffn_layer_norm_b = tv(p + "ffn_layer_norm.bias")
            L.ffn_inner_w, L.ffn_inner_b = tv(p + "ffn.inner_proj.weight"), tv(p + "ffn.inner_proj.bias")
            L.ffn_out_w, L.ffn_out_b = tv(p + "ffn.output_proj.weight"), tv(p + "ffn.output_proj.bias")
        w = _lib.smi_text_encoder_weights()
        w.embed = tv("encoder_frontend.embed.weight")
        if cfg.no_token_positional_embeddings:
            pass                                         # pos_table.data stays NULL (SMI_ENC_NO_POSITIONS)
        elif cfg.learned_pos:
            w.pos_table = tv("encoder_frontend.pos_encoder.weight")
        else:
            w.pos_table = _tensor_view(sinusoidal_table(cfg.model_max_seq_len + cfg.pos_offset, d), keep)
        w.final_layer_norm_w = tv("layer_norm.weight")
        w.final_layer_norm_b = tv("layer_norm.bias")
        w.layers = C.cast(layers, C.POINTER(_lib.smi_text_encoder_layer))
        if cfg.normalize_before:
            w.encoder_layer_norm_w, w.encoder_layer_norm_b = tv("encoder.layer_norm.weight"), tv("encoder.layer_norm.bias")
        if cfg.layernorm_embedding:
            w.embed_layer_norm_w = tv("encoder_frontend.layer_norm.weight")
            w.embed_layer_norm_b = tv("encoder_frontend.layer_norm.bias")
        if attn_pool:
            # AttentionEncoderOutputPooler (sonar/nn/encoder_pooler.py:49-95, factory.py:155-226): the pooler's one
            # input token after its frontend -- embed.weight[bos = 0] * sqrt(E) + the sinusoidal encoding of position 0
            e_dim = self.embedding_dim
            q0 = sd["pooler.decoder_frontend.embed.weight"][0].detach().float().cpu() * math.sqrt(e_dim) \
                + sinusoidal_table(1, e_dim)[0]
            w.pooler_query = _tensor_view(q0, keep)
            w.pooler_proj_w, w.pooler_proj_b = tv("pooler.projection_out.weight"), tv("pooler.projection_out.bias")
            if cfg.normalize_before:
                w.pooler_layer_norm_w = tv("pooler.decoder.layer_norm.weight")
                w.pooler_layer_norm_b = tv("pooler.decoder.layer_norm.bias")
            pl = (_lib.smi_text_pooler_layer * max(cfg.num_decoder_layers, 1))()
            for i in range(cfg.num_decoder_layers):
                p = f"pooler.decoder.layers.{i}."
                P = pl[i]
                P.self_attn_layer_norm_w, P.self_attn_layer_norm_b = tv(p + "self_attn_layer_norm.weight"), tv(p + "self_attn_layer_norm.bias")
                P.self_v_w, P.self_v_b = tv(p + "self_attn.v_proj.weight"), tv(p + "self_attn.v_proj.bias")
                P.self_out_w, P.self_out_b = tv(p + "self_attn.output_proj.weight"), tv(p + "self_attn.output_proj.bias")
                P.cross_layer_norm_w = tv(p + "encoder_decoder_attn_layer_norm.weight")
                P.cross_layer_norm_b = tv(p + "encoder_decoder_attn_layer_norm.bias")
                for nm in ("q", "k", "v"):
                    setattr(P, f"cross_{nm}_w", tv(p + f"encoder_decoder_attn.{nm}_proj.weight"))
                    setattr(P, f"cross_{nm}_b", tv(p + f"encoder_decoder_attn.{nm}_proj.bias"))
                P.cross_out_w = tv(p + "encoder_decoder_attn.output_proj.weight")
                P.cross_out_b = tv(p + "encoder_decoder_attn.output_proj.bias")
                P.ffn_layer_norm_w, P.ffn_layer_norm_b = tv(p + "ffn_layer_norm.weight"), tv(p + "ffn_layer_norm.bias")
                P.ffn_inner_w, P.ffn_inner_b = tv(p + "ffn.inner_proj.weight"), tv(p + "ffn.inner_proj.bias")
                P.ffn_out_w, P.ffn_out_b = tv(p + "ffn.output_proj.weight"), tv(p + "ffn.output_proj.bias")
            w.pooler = C.cast(pl, C.POINTER(_lib.smi_text_pooler_layer))
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_text_encoder_create(C.byref(ccfg), C.byref(w), max_tokens_hint,
                                                        C.byref(handle)))
        self._handle = handle
        del keep

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                self.lib.smi_text_encoder_destroy(h)
            except Exception:
                pass
            self._handle = None

    @property
    def device_bytes(self) -> int:
        return int(self.lib.smi_text_encoder_device_bytes(self._handle))

    def check(self) -> None:
        """Synchronise the current stream and raise IndexError if a batch held out-of-vocabulary token
        ids (the reference's embedding lookup raises there; smi_text_encoder_status)."""
        with torch.cuda.device(self.device):
            rc = self.lib.smi_text_encoder_status(self._handle, _lib.current_stream_ptr())
        if rc != _lib.SMI_OK:
            raise IndexError(self.lib.smi_last_error().decode("utf-8", "replace"))

    def set_profiling(self, enable: bool) -> None:
        _lib.check(self.lib.smi_text_encoder_set_profiling(self._handle, 1 if enable else 0))

    def read_profile(self) -> Dict[str, Dict[str, float]]:
        """Per-kernel {ms, launches} accumulated since the last read (synchronises the events)."""
        n = len(_lib.PROF_SLOTS)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        _lib.check(self.lib.smi_text_encoder_read_profile(self._handle, ms, cnt))
        return {name: {"ms": ms[i], "launches": int(cnt[i])} for i, name in enumerate(_lib.PROF_SLOTS)}

    def forward(self, ids: torch.Tensor, seq_lens: Optional[Union[torch.Tensor, Sequence[int]]],
                out_dtype: torch.dtype = torch.float16, return_encoded: bool = False):
        """ids: int64 [N,S] on the engine's device; seq_lens: host ints [N] or None."""
        if ids.dim() != 2:
            raise ValueError("ids must be [N, S]")
        if ids.device != self.device:
            ids = ids.to(self.device)
        ids = ids.to(torch.int64).contiguous()
        n, s = ids.shape
        lens_arr = None
        if seq_lens is not None:
            if isinstance(seq_lens, torch.Tensor):
                seq_lens = seq_lens.detach().to("cpu", torch.int32).tolist()
            if len(seq_lens) != n:
                raise ValueError("seq_lens must have one entry per sequence")
            lens_arr = (C.c_int32 * n)(*[int(v) for v in seq_lens])
        if out_dtype not in (torch.float16, torch.float32, torch.bfloat16):
            raise ValueError("out_dtype must be float16, bfloat16 or float32")
        want = out_dtype
        if out_dtype == torch.bfloat16:  # bf16 exists at the boundary only: fp32 out of the engine, rounded once
            out_dtype = torch.float32
        emb = torch.empty((n, self.embedding_dim), dtype=out_dtype, device=self.device)
        enc = torch.empty((n, s, self.cfg.model_dim), dtype=out_dtype, device=self.device) if return_encoded else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_text_encoder_forward(
                self._handle, ids.data_ptr(), C.cast(lens_arr, C.c_void_p) if lens_arr is not None else None,
                n, s, emb.data_ptr(), enc.data_ptr() if enc is not None else None,
                _lib.SMI_F32 if out_dtype == torch.float32 else _lib.SMI_F16,
                _lib.current_stream_ptr()))
        if want != out_dtype:
            emb = _lib.cast(emb, want)
            enc = _lib.cast(enc, want) if enc is not None else None
        return emb, enc


# ------------------------------------------------ reference-shaped model objects
@dataclass
class SonarEncoderOutput:
    """sonar/models/encoder_model.py:17-38."""

    encoded_seqs: Optional[torch.Tensor]
    sentence_embeddings: torch.Tensor
    padding_mask: Optional["PaddingMask"]


@dataclass
class PaddingMask:
    """Stand-in for fairseq2.nn.padding.PaddingMask: lengths + padded length."""

    seq_lens: torch.Tensor  # int [N] (host or device)
    batch_seq_len: int


@dataclass
class SequenceBatch:
    """Stand-in for fairseq2.models.sequence.SequenceBatch (seqs + optional mask)."""

    seqs: torch.Tensor
    padding_mask: Optional[PaddingMask]
    # recorded after the H2D copy that produced `seqs` when that copy ran on another thread / stream;
    # the model makes its stream wait for it before reading `seqs`
    ready: Optional["torch.cuda.Event"] = None


class _PosEncoderInfo:
    def __init__(self, max_seq_len: int):
        self.max_seq_len = max_seq_len


class _FrontendInfo:
    def __init__(self, max_seq_len: int):
        self.pos_encoder = _PosEncoderInfo(max_seq_len)


class SonarTextTransformerEncoderModel:
    """Drop-in for the object `TextToEmbeddingModelPipeline` calls as `self.model(batch)`
    (sonar/inference_pipelines/text.py:244): callable on a SequenceBatch, exposes
    `.eval()`, `.dtype`, `.encoder_frontend.pos_encoder.max_seq_len` (text.py:202)."""

    def __init__(self, cfg: SonarTextEncoderConfig, state_dict: Mapping[str, torch.Tensor],
                 device: Union[str, torch.device] = "cuda:0", dtype: torch.dtype = torch.float16,
                 return_encoded_seqs: bool = False, max_tokens_hint: int = 0,
                 fp16_residual: Optional[bool] = None):
        """dtype: dtype of the returned embeddings and, as in the reference (`model.to(device, dtype)`,
        text.py:161-162), of the residual stream: fp16 model -> fp16 residual adds (one rounding each),
        fp32 model -> fp32 residual stream.  `fp16_residual` overrides that choice."""
        if dtype not in (torch.float16, torch.bfloat16, torch.float32):
            raise ValueError(f"unsupported model dtype {dtype} (float16, bfloat16 or float32)")
        if fp16_residual is None:
            # fp16 model -> fp16 residual stream, as the reference's `.half()` model.  A bf16 model's activations have
            # fp32 RANGE (sonar/inference_pipelines/text.py:36-54 accepts any dtype), which an fp16 stream does not: it
            # gets the fp32 residual stream (round 4; the GEMM operands -- LayerNorm outputs, q/k/v, the FFN hidden
            # activation -- are O(1)-scaled quantities and stay fp16 operands of the fp16 MFMA, fp32 accumulation).
            # bf16 WEIGHTS are exact fp16 operands as long as they lie in fp16's exponent range (|w| in
            # [6.1e-5, 65504]; smaller ones lose mantissa bits as fp16 subnormals); embeddings are rounded to bf16
            # once, on the way out (include/sonar_mi355.h, SMI_BF16).
            fp16_residual = dtype == torch.float16
        self.config = cfg
        self.dtype = dtype
        self.model_dim = cfg.model_dim
        self.pooling = cfg.pooling
        self.return_encoded_seqs = return_encoded_seqs
        self.encoder_frontend = _FrontendInfo(cfg.model_max_seq_len)
        self.engine = TextEncoderEngine(cfg, state_dict, device, max_tokens_hint, fp16_residual)
        self.device = self.engine.device
        # Out-of-vocabulary token ids: the reference's embedding lookup raises IndexError inside forward().  The engine
        # raises a sticky device flag instead and reports it from `engine.check()` (one stream synchronisation).
        # forward() runs that check before it returns, so a direct caller never gets embeddings computed from a
        # wrong table row silently; a caller that queues many batches (predict(), sharded_encode, bench.py) sets
        # `deferred_check` for its loop and calls `engine.check()` once at the end.
        self.deferred_check = False

    def eval(self):
        return self

    @contextlib.contextmanager
    def deferring_check(self):
        """Queue forward() calls without a per-call synchronisation; the out-of-vocabulary check runs once on exit."""
        prev, self.deferred_check = self.deferred_check, True
        try:
            yield self
        except BaseException:
            # the block failed for a reason of its own: drain and CLEAR the sticky device flag all the same (or the next,
            # unrelated forward() would report this block's bad id), but let the original exception travel
            self.deferred_check = prev
            try:
                self.engine.check()
            except Exception:
                pass
            raise
        self.deferred_check = prev
        self.engine.check()

    def __call__(self, batch: SequenceBatch) -> SonarEncoderOutput:
        return self.forward(batch)

    @torch.inference_mode()
    def forward(self, batch: SequenceBatch) -> SonarEncoderOutput:
        lens = batch.padding_mask.seq_lens if batch.padding_mask is not None else None
        if getattr(batch, "ready", None) is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(batch.ready)
            if batch.seqs.is_cuda:  # allocated on the producer's stream: this stream reads it too (allocator reuse)
                batch.seqs.record_stream(cur)
        emb, enc = self.engine.forward(batch.seqs, lens, self.dtype, self.return_encoded_seqs)
        if not self.deferred_check:
            self.engine.check()  # IndexError for out-of-vocabulary ids, as the reference's embedding lookup
        return SonarEncoderOutput(encoded_seqs=enc, sentence_embeddings=emb, padding_mask=batch.padding_mask)


def load_sonar_text_encoder(checkpoint: Union[str, Mapping], arch: str = "basic",
                            device: Union[str, torch.device] = "cuda:0",
                            dtype: torch.dtype = torch.float16,
                            config: Optional[SonarTextEncoderConfig] = None,
                            load_stats: Optional[dict] = None) -> SonarTextTransformerEncoderModel:
    """hub.load() equivalent for a card name (resolved under $SONAR_CHECKPOINT_DIR, sonar_amd/cards.py), a
    local checkpoint file, or an in-memory dict (reference: sonar/inference_pipelines/text.py:161-162).
    Files go through the packed cache (sonar_amd/packed_cache.py): the second load of a file skips the
    unpickling, the key conversion and the fp32 -> fp16 conversion."""
    cfg_arch = arch
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        from .cards import resolve_checkpoint
        from .packed_cache import load_converted

        path, cfg_arch = resolve_checkpoint(checkpoint, arch)
        sd = load_converted(path, convert_sonar_text_encoder_checkpoint, "text_encoder", load_stats)
    else:
        sd = convert_sonar_text_encoder_checkpoint(checkpoint)
    cfg = config or get_text_encoder_config(cfg_arch)
    return SonarTextTransformerEncoderModel(cfg, sd, device=device, dtype=dtype)
