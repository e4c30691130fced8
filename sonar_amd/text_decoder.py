"""Host-side mirror of the reference's SONAR text decoder objects, backed by the HIP
engine (libsonar_mi355.so).

Reference interfaces mirrored (paths relative to facebookresearch/SONAR):
  * SonarTextDecoderConfig + archs `basic` / `small` / `toy`   sonar/models/sonar_text/config.py:130-255
  * checkpoint key conversion                                   sonar/models/sonar_text/handler.py:122-172
  * ConditionalTransformerDecoderModel / SonarEncoderDecoderModel
        sonar/nn/conditional_decoder_model.py:26-94, sonar/models/sonar_translation/model.py:48-78
  * BeamSearchSeq2SeqGenerator defaults (fairseq2 ~=0.4) -- run ON DEVICE here:
    decoder step, fp32 log-softmax, top-2*beam selection, EOS bookkeeping and the beam
    re-indexing of the KV cache (an ancestry table, no cache copy) are all engine kernels.
"""
from __future__ import annotations

import ctypes as C
import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from .text_encoder import VocabularyInfo, _tensor_view, sinusoidal_table


@dataclass
class SonarTextDecoderConfig:
    """Same fields as the reference dataclass (config.py:130-190)."""

    model_dim: int = 1024
    max_seq_len: int = 512
    vocab_info: VocabularyInfo = field(default_factory=lambda: VocabularyInfo(size=256206))
    num_encoder_layers: int = 24
    num_decoder_layers: int = 24
    num_encoder_attn_heads: int = 16
    num_decoder_attn_heads: int = 16
    ffn_inner_dim: int = 1024 * 8
    activation_fn: str = "ReLU"
    layernorm_embedding: bool = False
    no_scale_embedding: bool = False
    no_token_positional_embeddings: bool = False
    learned_pos: bool = False
    emb_dropout_p: float = 0.1
    attention_dropout_p: float = 0.1
    activation_dropout_p: float = 0.1
    normalize_before: bool = True
    input_dim: Optional[int] = None

    @property
    def pos_offset(self) -> int:
        return (self.vocab_info.pad_idx or 0) + 1


def _basic() -> SonarTextDecoderConfig:
    return SonarTextDecoderConfig()


def _small(vocab_size: int = 32005, depth: int = 6, hidden_dim: int = 1024 * 4) -> SonarTextDecoderConfig:
    c = _basic()
    c.vocab_info = VocabularyInfo(size=vocab_size)
    c.num_encoder_layers = depth
    c.num_decoder_layers = depth
    c.ffn_inner_dim = hidden_dim
    return c


def _toy() -> SonarTextDecoderConfig:
    # the reference's `toy` arch (config.py:232-255: model_dim 32, 4 heads of 8).  The MFMA engines are shaped for
    # head_dim 64; this arch -- and any other shape they do not tile for -- runs on the library's generic-dimension
    # kernels (csrc/flex.hip, fp32), selected inside smi_text_decoder_create.
    return SonarTextDecoderConfig(model_dim=32, vocab_info=VocabularyInfo(size=1024), num_encoder_layers=2,
                                  num_decoder_layers=2, num_encoder_attn_heads=4, num_decoder_attn_heads=4,
                                  ffn_inner_dim=128)


TEXT_DECODER_ARCHS = {"basic": _basic, "small": _small, "toy": _toy}


def get_text_decoder_config(arch: str) -> SonarTextDecoderConfig:
    try:
        return TEXT_DECODER_ARCHS[arch]()
    except KeyError:
        raise ValueError(f"unknown sonar text decoder arch {arch!r}; known: {sorted(TEXT_DECODER_ARCHS)}")


_FAIRSEQ1_DECODER_KEY_MAP = [
    (r"^layers\.([0-9]+)\.self_attn\.out_proj\.", r"decoder.layers.\1.self_attn.output_proj."),
    (r"^layers\.([0-9]+)\.self_attn\.(q|k|v)_proj\.", r"decoder.layers.\1.self_attn.\2_proj."),
    (r"^layers\.([0-9]+)\.self_attn_layer_norm\.", r"decoder.layers.\1.self_attn_layer_norm."),
    (r"^layers\.([0-9]+)\.encoder_attn\.out_proj\.", r"decoder.layers.\1.encoder_decoder_attn.output_proj."),
    (r"^layers\.([0-9]+)\.encoder_attn\.(q|k|v)_proj\.", r"decoder.layers.\1.encoder_decoder_attn.\2_proj."),
    (r"^layers\.([0-9]+)\.encoder_attn_layer_norm\.", r"decoder.layers.\1.encoder_decoder_attn_layer_norm."),
    (r"^layers\.([0-9]+)\.ffn\.(inner|output)_proj\.", r"decoder.layers.\1.ffn.\2_proj."),
    (r"^layers\.([0-9]+)\.ffn_layer_norm\.", r"decoder.layers.\1.ffn_layer_norm."),
    (r"^layers\.([0-9]+)\.fc1\.", r"decoder.layers.\1.ffn.inner_proj."),
    (r"^layers\.([0-9]+)\.fc2\.", r"decoder.layers.\1.ffn.output_proj."),
    (r"^layers\.([0-9]+)\.final_layer_norm\.", r"decoder.layers.\1.ffn_layer_norm."),
    (r"^output_projection\.", r"final_proj."),
    (r"^embed_tokens\.", r"decoder_frontend.embed."),
    (r"^layer_norm\.", r"decoder.layer_norm."),
]


def convert_sonar_text_decoder_checkpoint(checkpoint: Mapping) -> Dict[str, torch.Tensor]:
    """Flat fairseq2-style state dict from either layout the reference accepts
    (handler.py:122-172).  fairseq1 keys are renamed and the control-token rows of the
    embedding permuted (BOS, PAD, EOS, UNK) -> (PAD, UNK, BOS, EOS).  `final_proj.weight`
    is dropped: the output projection is tied to the (permuted) embedding
    (TiedProjection, factory.py:306-307)."""
    if "model" in checkpoint and "decoder_frontend.embed.weight" in checkpoint["model"]:
        out = dict(checkpoint["model"])
        out.pop("final_proj.weight", None)
        return out
    if "state_dict" not in checkpoint:
        if "decoder_frontend.embed.weight" in checkpoint:
            out = dict(checkpoint)
            out.pop("final_proj.weight", None)
            return out
        raise ValueError("unrecognised SONAR text decoder checkpoint layout")
    out: Dict[str, torch.Tensor] = {}
    for key, val in checkpoint["state_dict"].items():
        if key in ("version", "embed_positions._float_tensor"):
            continue
        new = key
        for pat, rep in _FAIRSEQ1_DECODER_KEY_MAP:
            new, n = re.subn(pat, rep, new)
            if n:
                break
        out[new] = val
    emb = out["decoder_frontend.embed.weight"].clone()
    emb[[0, 1, 2, 3]] = emb[[1, 3, 0, 2]]
    out["decoder_frontend.embed.weight"] = emb
    out.pop("final_proj.weight", None)
    return out


# --------------------------------------------------------------------- engine
class TextDecoderEngine:
    """Owns one `smi_text_decoder` handle (packed fp16 weights in HBM + generation workspace)."""

    def __init__(self, cfg: SonarTextDecoderConfig, state_dict: Mapping[str, torch.Tensor],
                 device: Union[str, torch.device] = "cuda:0",
                 tokenizer_special: Tuple[int, int, int, int] = (0, 1, 2, 3), dtype: torch.dtype = torch.float16):
        """dtype: the model's nominal dtype (the reference's `model.to(device, dtype)`).  The engine multiplies fp16
        operands into fp32 accumulators whatever it is; what follows the dtype is the storage type of the beam search's
        logits: a float16 model's logits are float16, as its fp16 final_proj produces them in the reference."""
        if cfg.activation_fn != "ReLU" or cfg.layernorm_embedding or cfg.learned_pos or cfg.no_token_positional_embeddings:
            raise NotImplementedError("decoder variant not covered by the MI355X engine")
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the SONAR MI355X engine runs on a HIP device only (no CPU path)")
        self.lib = _lib.load()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.check(self.lib.smi_init(idx))
        d = cfg.model_dim
        pad, unk, bos, eos = tokenizer_special
        ccfg = _lib.smi_text_decoder_config(
            model_dim=d, num_layers=cfg.num_decoder_layers, num_heads=cfg.num_decoder_attn_heads,
            ffn_inner_dim=cfg.ffn_inner_dim, vocab_size=cfg.vocab_info.size, max_seq_len=cfg.max_seq_len,
            pos_offset=cfg.pos_offset, input_dim=cfg.input_dim or d,
            embed_scale=1.0 if cfg.no_scale_embedding else math.sqrt(d), ln_eps=1e-5,
            pad_idx=pad, unk_idx=unk, bos_idx=bos, eos_idx=eos)
        keep: List[torch.Tensor] = []

        def tv(name: str) -> _lib.smi_tensor:
            if name not in state_dict:
                raise KeyError(f"checkpoint is missing {name}")
            return _tensor_view(state_dict[name], keep)

        layers = (_lib.smi_text_decoder_layer * max(cfg.num_decoder_layers, 1))()
        for i in range(cfg.num_decoder_layers):
            p = f"decoder.layers.{i}."
            L = layers[i]
            L.self_attn_layer_norm_w = tv(p + "self_attn_layer_norm.weight")
            L.self_attn_layer_norm_b = tv(p + "self_attn_layer_norm.bias")
            L.q_w, L.q_b = tv(p + "self_attn.q_proj.weight"), tv(p + "self_attn.q_proj.bias")
            L.k_w, L.k_b = tv(p + "self_attn.k_proj.weight"), tv(p + "self_attn.k_proj.bias")
            L.v_w, L.v_b = tv(p + "self_attn.v_proj.weight"), tv(p + "self_attn.v_proj.bias")
            L.out_w, L.out_b = tv(p + "self_attn.output_proj.weight"), tv(p + "self_attn.output_proj.bias")
            L.cross_v_w = tv(p + "encoder_decoder_attn.v_proj.weight")
            L.cross_v_b = tv(p + "encoder_decoder_attn.v_proj.bias")
            L.cross_out_w = tv(p + "encoder_decoder_attn.output_proj.weight")
            L.cross_out_b = tv(p + "encoder_decoder_attn.output_proj.bias")
            L.ffn_layer_norm_w = tv(p + "ffn_layer_norm.weight")
            L.ffn_layer_norm_b = tv(p + "ffn_layer_norm.bias")
            L.ffn_inner_w, L.ffn_inner_b = tv(p + "ffn.inner_proj.weight"), tv(p + "ffn.inner_proj.bias")
            L.ffn_out_w, L.ffn_out_b = tv(p + "ffn.output_proj.weight"), tv(p + "ffn.output_proj.bias")
        w = _lib.smi_text_decoder_weights()
        w.embed = tv("decoder_frontend.embed.weight")
        w.pos_table = _tensor_view(sinusoidal_table(cfg.max_seq_len + cfg.pos_offset, d), keep)
        w.final_layer_norm_w = tv("decoder.layer_norm.weight")
        w.final_layer_norm_b = tv("decoder.layer_norm.bias")
        w.layers = C.cast(layers, C.POINTER(_lib.smi_text_decoder_layer))
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_text_decoder_create(C.byref(ccfg), C.byref(w), C.byref(handle)))
        self._handle = handle
        del keep
        self.set_beam_logits_dtype(torch.float16 if dtype == torch.float16 else torch.float32)
        self.set_slab_dtype(torch.float16 if dtype == torch.float16 else torch.float32)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                self.lib.smi_text_decoder_destroy(h)
            except Exception:
                pass
            self._handle = None

    def _emb(self, embeddings: torch.Tensor) -> torch.Tensor:
        cond = self.cfg.input_dim or self.cfg.model_dim     # width of the conditioning vector (factory.py:264)
        if embeddings.dim() != 2 or embeddings.shape[1] != cond:
            raise ValueError(f"embeddings must be [n, {cond}]")
        e = embeddings.to(self.device)
        if e.dtype == torch.bfloat16:
            e = _lib.cast(e, torch.float32)      # a bf16 sentence vector is exact in fp32
        elif e.dtype not in (torch.float16, torch.float32):
            e = e.float()
        return e.contiguous()

    def logits(self, embeddings: torch.Tensor, prev_tokens: torch.Tensor) -> torch.Tensor:
        """Teacher-forced logits fp32 [n, t, vocab] (cf. test_text_sonar.py:61-105)."""
        e = self._emb(embeddings)
        prev = prev_tokens.to(self.device, torch.int64).contiguous()
        n, t = prev.shape
        if n != e.shape[0]:
            raise ValueError("one embedding per token row expected")
        out = torch.empty((n, t, self.cfg.vocab_info.size), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_text_decoder_logits(
                self._handle, e.data_ptr(), _lib.SMI_F32 if e.dtype == torch.float32 else _lib.SMI_F16, n,
                prev.data_ptr(), t, out.data_ptr(), _lib.current_stream_ptr()))
        return out

    def _length_limits(self, plen: int, min_gen_len: int, max_gen_len: Tuple[int, int],
                       max_seq_len: Optional[int], source_len: Optional[int]) -> Tuple[int, int]:
        """fairseq2's Seq2SeqGenerator length rule: max_gen_len = a * max_source_len + b, where
        max_source_len is `source_seqs.size(1)` (no padding mask) or the longest source in the batch.
        EmbeddingToTextModelPipeline hands the generator `torch.stack(embeddings)` [n, model_dim] as
        `source_seqs` (text.py:329-333), so there the "source length" is model_dim and the cap is in
        practice the decoder's max_seq_len; TextToText / SpeechToText pass their token / frame count."""
        model_max = max_seq_len if max_seq_len is not None else self.cfg.max_seq_len
        if model_max > self.cfg.max_seq_len:
            raise ValueError(f"max_seq_len cannot be larger than the decoder's {self.cfg.max_seq_len}")
        if source_len is None:
            source_len = self.cfg.input_dim or self.cfg.model_dim   # width of the stacked sentence vectors
        gen_cap = int(max_gen_len[0] * int(source_len) + max_gen_len[1])
        if gen_cap < 1:
            raise ValueError("`max_gen_len` must be greater than or equal to 1 for the given source length")
        if min_gen_len > gen_cap:
            raise ValueError(f"`min_gen_len` must be less than or equal to `max_gen_len` ({gen_cap}), "
                             f"but is {min_gen_len} instead")
        max_len = min(plen + gen_cap, model_max)
        if max_len <= plen:
            raise ValueError("`max_seq_len` leaves no room for generation after the prompt")
        return max_len, min(plen + min_gen_len, max_len)

    def set_beam_logits_dtype(self, dtype: torch.dtype) -> None:
        """Type of the logits the beam search of generate() compares (smi_text_decoder_set_beam_logits_dtype): float16 is what
        the reference's fp16 model produces (its tied final_proj is an fp16 Linear), float32 keeps the accumulators."""
        if dtype not in (torch.float16, torch.float32):
            raise ValueError("float16 or float32")
        _lib.check(self.lib.smi_text_decoder_set_beam_logits_dtype(
            self._handle, _lib.SMI_F16 if dtype == torch.float16 else _lib.SMI_F32))

    def set_slab_dtype(self, dtype: torch.dtype) -> None:
        """Storage type of the split-K partial sums of the two N = model_dim projections inside generate()
        (smi_text_decoder_set_slab_dtype; its own setting since round 5): float16 (an fp16 model's default here: the
        reference's fp16 model rounds every sublayer output to fp16; partials saturate, they never become inf) or float32."""
        if dtype not in (torch.float16, torch.float32):
            raise ValueError("float16 or float32")
        _lib.check(self.lib.smi_text_decoder_set_slab_dtype(
            self._handle, _lib.SMI_F16 if dtype == torch.float16 else _lib.SMI_F32))

    def set_chains(self, chains: int) -> None:
        """Independent decode chains of generate() (smi_text_decoder_set_chains): 0 = the engine's choice."""
        _lib.check(self.lib.smi_text_decoder_set_chains(self._handle, int(chains)))

    def last_margins(self, n: int) -> torch.Tensor:
        """Decision margins fp32 [n, 2] of the last generate() call (smi_text_decoder_last_margins)."""
        out = torch.empty((n, 2), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_text_decoder_last_margins(self._handle, out.data_ptr(), n,
                                                              _lib.current_stream_ptr()))
        return out

    def generate(self, embeddings: torch.Tensor, prompt: Sequence[int], beam_size: int = 5, min_gen_len: int = 1,
                 max_gen_len: Tuple[int, int] = (1, 128), max_seq_len: Optional[int] = None,
                 normalize_scores: bool = True, len_penalty: float = 1.0, unk_penalty: float = 0.0,
                 temperature: float = 1.0, source_len: Optional[int] = None):
        """Beam search with fairseq2's BeamSearchSeq2SeqGenerator defaults.
        Returns (tokens int32 [n, beam, L] (-1 padded), lens int32 [n, beam], scores fp32 [n, beam]),
        hypotheses best first; tokens are the generated part (after the prompt) incl. the final EOS.
        `source_len`: see `_length_limits` (None = the sentence-vector case)."""
        e = self._emb(embeddings)
        n = e.shape[0]
        plen = len(prompt)
        max_len, min_len = self._length_limits(plen, min_gen_len, max_gen_len, max_seq_len, source_len)
        bp = _lib.smi_beam_search_params(beam_size=beam_size, max_seq_len=max_len, min_seq_len=min_len,
                                         normalize_scores=1 if normalize_scores else 0, len_penalty=len_penalty,
                                         unk_penalty=unk_penalty, temperature=temperature, reserved=0)
        toks = torch.empty((n, beam_size, max_len), dtype=torch.int32, device=self.device)
        lens = torch.empty((n, beam_size), dtype=torch.int32, device=self.device)
        scores = torch.empty((n, beam_size), dtype=torch.float32, device=self.device)
        prompt_arr = (C.c_int64 * plen)(*[int(t) for t in prompt])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_text_decoder_generate(
                self._handle, e.data_ptr(), _lib.SMI_F32 if e.dtype == torch.float32 else _lib.SMI_F16, n,
                prompt_arr, plen, C.byref(bp), toks.data_ptr(), lens.data_ptr(), scores.data_ptr(),
                _lib.current_stream_ptr()))
        return toks, lens, scores

    def sample(self, embeddings: torch.Tensor, prompt: Sequence[int], sampler, min_gen_len: int = 1,
               max_gen_len: Tuple[int, int] = (1, 128), max_seq_len: Optional[int] = None,
               normalize_scores: bool = True, len_penalty: float = 1.0, unk_penalty: float = 0.0,
               temperature: float = 1.0, seed: Optional[int] = None, sentence_offset: int = 0,
               source_len: Optional[int] = None):
        """fairseq2's SamplingSeq2SeqGenerator (one hypothesis per sentence) with a TopKSampler /
        TopPSampler (sonar_amd.generation).  Returns (tokens int32 [n, L] (-1 padded), lens int32 [n],
        scores fp32 [n]).  `unk_penalty` is subtracted from the UNK token's probability before the filter, as the
        reference's generator does.  `seed` None draws one from torch's global CPU generator, so
        `torch.manual_seed` makes a run repeatable as it does for the reference; the random stream of a
        sentence depends on (seed, sentence_offset + its index, step) only, not on the batch."""
        from .generation import resolve_sampler

        kind, k, p = resolve_sampler(sampler)
        e = self._emb(embeddings)
        n = e.shape[0]
        plen = len(prompt)
        max_len, min_len = self._length_limits(plen, min_gen_len, max_gen_len, max_seq_len, source_len)
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())
        seed = (int(seed) + 0x9E3779B97F4A7C15 * 65536 * int(sentence_offset)) & 0xFFFFFFFFFFFFFFFF
        sp = _lib.smi_sampling_params(sampler=kind, top_k=k, top_p=p, temperature=temperature, max_seq_len=max_len,
                                      min_seq_len=min_len, normalize_scores=1 if normalize_scores else 0,
                                      len_penalty=len_penalty, seed=seed, unk_penalty=unk_penalty)
        toks = torch.empty((n, max_len), dtype=torch.int32, device=self.device)
        lens = torch.empty((n,), dtype=torch.int32, device=self.device)
        scores = torch.empty((n,), dtype=torch.float32, device=self.device)
        prompt_arr = (C.c_int64 * plen)(*[int(t) for t in prompt])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_text_decoder_sample(
                self._handle, e.data_ptr(), _lib.SMI_F32 if e.dtype == torch.float32 else _lib.SMI_F16, n,
                prompt_arr, plen, C.byref(sp), toks.data_ptr(), lens.data_ptr(), scores.data_ptr(),
                _lib.current_stream_ptr()))
        return toks, lens, scores


class ConditionalTransformerDecoderModel:
    """Drop-in for the object `EmbeddingToTextModelPipeline` receives as `decoder`
    (sonar/nn/conditional_decoder_model.py:26-94): exposes model_dim, max_target_seq_len and the
    engine's on-device generation."""

    def __init__(self, cfg: SonarTextDecoderConfig, state_dict: Mapping[str, torch.Tensor],
                 device: Union[str, torch.device] = "cuda:0", dtype: torch.dtype = torch.float16):
        self.config = cfg
        self.model_dim = cfg.model_dim
        self.max_target_seq_len = cfg.max_seq_len
        self.dtype = dtype
        self.engine = TextDecoderEngine(cfg, state_dict, device, dtype=dtype)
        self.device = self.engine.device

    def eval(self):
        return self


class SonarEncoderDecoderModel:
    """sonar/models/sonar_translation/model.py:24-78: the (encoder, decoder) pair fairseq2's generators drive through
    encode / decode / project.  Here generation runs inside the decoder engine, so the object only holds the two
    engine-backed models the speech / text translation pipelines hand around (`model.encoder(batch)
    .sentence_embeddings` is what `encode` returns, unsqueezed, in the reference)."""

    def __init__(self, encoder, decoder: ConditionalTransformerDecoderModel) -> None:
        enc_dim, dec_dim = getattr(encoder, "model_dim", None), getattr(decoder, "model_dim", None)
        if enc_dim is not None and dec_dim is not None and enc_dim != dec_dim:
            raise ValueError(f"`model_dim` of `encoder` and `model_dim` of `decoder` must be equal, but are {enc_dim} "
                             f"and {dec_dim} instead.")
        self.encoder = encoder
        self.decoder = decoder
        self.model_dim = dec_dim
        self.max_target_seq_len = getattr(decoder, "max_target_seq_len", None)

    @property
    def dtype(self):
        return self.encoder.dtype

    @property
    def device(self):
        return self.decoder.device

    def eval(self):
        return self

    def to(self, device=None, dtype=None):
        """The engines live on the HIP device they were created on; `.to` accepts that device and nothing else."""
        if device is not None and torch.device(device).type != "cpu":
            want, have = torch.device(device), torch.device(self.device)
            # an index-less "cuda" means the CURRENT device on both sides, as torch resolves it (resolving one side
            # to the current device and the other to cuda:0 made `.to("cuda")` fail on a rank whose current device is not 0)
            cur = torch.cuda.current_device() if torch.cuda.is_available() else 0
            if want.type == "cuda" and want.index is None:
                want = torch.device("cuda", cur)
            if have.type == "cuda" and have.index is None:
                have = torch.device("cuda", cur)
            if want != have:
                raise RuntimeError(f"the engine-backed model lives on {self.device}")
        return self

    def encode(self, seqs: torch.Tensor, padding_mask=None):
        from .text_encoder import SequenceBatch

        return self.encoder(SequenceBatch(seqs, padding_mask)).sentence_embeddings.unsqueeze(1), None


def load_sonar_text_decoder(checkpoint: Union[str, Mapping], arch: str = "basic",
                            device: Union[str, torch.device] = "cuda:0", dtype: torch.dtype = torch.float16,
                            config: Optional[SonarTextDecoderConfig] = None,
                            load_stats: Optional[dict] = None) -> ConditionalTransformerDecoderModel:
    """Card name / checkpoint file (through the packed cache) / in-memory dict -> decoder model."""
    cfg_arch = arch
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        from .cards import resolve_checkpoint
        from .packed_cache import load_converted

        path, cfg_arch = resolve_checkpoint(checkpoint, arch)
        sd = load_converted(path, convert_sonar_text_decoder_checkpoint, "text_decoder", load_stats)
    else:
        sd = convert_sonar_text_decoder_checkpoint(checkpoint)
    cfg = config or get_text_decoder_config(cfg_arch)
    return ConditionalTransformerDecoderModel(cfg, sd, device, dtype)
