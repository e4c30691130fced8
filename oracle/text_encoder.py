"""ORACLE -- test infrastructure only.  Never imported by the product package.

CPU / plain-PyTorch fp32 restatement of the reference's TextToEmbedding hot
path (facebookresearch/SONAR v0.4.0, paths relative to the reference repo).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module, and only as the checker / reported baseline.

Pinning status (see DESIGN.md "Oracle"):
  * pooling is pinned EXACTLY by the reference's own unit vectors
    (tests/unit_tests/test_sonar_pooling.py:16-68) -> tests/test_oracle_cpu.py;
  * the transformer stack is pinned against an independent implementation of
    the same architecture, HuggingFace `M2M100Encoder` (the reference itself
    loads SONAR weights into it: examples/finetune_sonar_as_toxicity_classifier.ipynb
    cells 50-57), through committed golden vectors in tests/golden/
    (generator: tests/golden/make_golden.py);
  * the reference's real-checkpoint goldens
    (tests/integration_tests/test_text_sonar.py:46-53) need the 3 GB checkpoint
    and fairseq2, neither present offline: for those values PARITY IS UNPINNED.
    The block semantics of the un-vendored dependency fairseq2 (~=0.4.0,
    pyproject.toml:27) are restated from its published behaviour and from the
    reference's call sites cited below.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class OracleTextEncoderConfig:
    """Fields of SonarTextEncoderConfig that reach the forward pass
    (sonar/models/sonar_text/config.py:14-85; arch `basic` at :92-116)."""

    model_dim: int = 1024
    num_layers: int = 24
    num_heads: int = 16
    ffn_inner_dim: int = 8192
    vocab_size: int = 256206
    max_seq_len: int = 512          # config value; +pad_idx+1 when _from_fairseq (factory.py:56-59)
    pad_idx: int = 1                # model vocab_info.pad_idx (config.py:96-98) -> position offset
    from_fairseq: bool = True
    no_scale_embedding: bool = False
    pooling: str = "mean"
    ln_eps: float = 1e-5
    # the less common builder options (config.py:54-85) and the attention pooler (factory.py:155-226)
    normalize_before: bool = False      # encoder stack norm order PRE -> its own final LayerNorm; pooler layers pre-norm
    layernorm_embedding: bool = False
    learned_pos: bool = False
    no_token_positional_embeddings: bool = False
    embedding_dim: Optional[int] = None # sentence-vector width with pooling == "attention"
    pooler_layers: int = 0              # config.num_decoder_layers
    pooler_heads: int = 0               # config.num_decoder_attn_heads
    pooler_ffn_dim: int = 0             # config.decoder_ffn_inner_dim or ffn_inner_dim

    @property
    def pos_offset(self) -> int:
        # SinusoidalPositionEncoder(_legacy_pad_idx=pad_idx): first token sits at
        # position pad_idx + 1 (factory.py:88-92).
        return self.pad_idx + 1

    @property
    def model_max_seq_len(self) -> int:
        return self.max_seq_len + (self.pad_idx + 1 if self.from_fairseq else 0)

    @property
    def edim(self) -> int:
        return self.embedding_dim or self.model_dim


def sinusoidal_table(num_positions: int, dim: int) -> torch.Tensor:
    """fp32 table, row p = encoding of absolute position p.

    fairseq-style half-split layout [sin(p*f_0..f_{h-1}) | cos(p*f_0..f_{h-1})],
    f_i = exp(-i * ln(1e4) / (h - 1)), h = dim // 2 (identical to
    HF M2M100SinusoidalPositionalEmbedding.get_embedding, checked in
    tests/test_oracle_cpu.py).  The reference selects rows t + pad_idx + 1.
    """
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000.0) / (half - 1)))
    ang = torch.arange(num_positions, dtype=torch.float32).unsqueeze(1) * freq.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(num_positions, 1)], dim=1)
    return tab


def param_names(cfg: OracleTextEncoderConfig):
    """fairseq2-style names produced by the reference's checkpoint conversion
    (sonar/models/sonar_text/handler.py:71-82; `layer_norm.*` stays unmapped)."""
    names = ["encoder_frontend.embed.weight", "layer_norm.weight", "layer_norm.bias"]
    for i in range(cfg.num_layers):
        p = f"encoder.layers.{i}."
        for lin in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj",
                    "self_attn.output_proj", "ffn.inner_proj", "ffn.output_proj"):
            names += [p + lin + ".weight", p + lin + ".bias"]
        for ln in ("self_attn_layer_norm", "ffn_layer_norm"):
            names += [p + ln + ".weight", p + ln + ".bias"]
    if cfg.normalize_before:
        names += ["encoder.layer_norm.weight", "encoder.layer_norm.bias"]
    if cfg.layernorm_embedding:
        names += ["encoder_frontend.layer_norm.weight", "encoder_frontend.layer_norm.bias"]
    if cfg.learned_pos and not cfg.no_token_positional_embeddings:
        names += ["encoder_frontend.pos_encoder.weight"]
    if cfg.pooling == "attention":
        names += ["pooler.decoder_frontend.embed.weight", "pooler.projection_out.weight", "pooler.projection_out.bias"]
        if cfg.normalize_before:
            names += ["pooler.decoder.layer_norm.weight", "pooler.decoder.layer_norm.bias"]
        for i in range(cfg.pooler_layers):
            p = f"pooler.decoder.layers.{i}."
            for att in ("self_attn", "encoder_decoder_attn"):
                for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
                    names += [p + f"{att}.{lin}.weight", p + f"{att}.{lin}.bias"]
            for lin in ("ffn.inner_proj", "ffn.output_proj"):
                names += [p + lin + ".weight", p + lin + ".bias"]
            for ln in ("self_attn_layer_norm", "encoder_decoder_attn_layer_norm", "ffn_layer_norm"):
                names += [p + ln + ".weight", p + ln + ".bias"]
    return names


def param_shape(cfg: OracleTextEncoderConfig, name: str) -> Tuple[int, ...]:
    d, f = cfg.model_dim, cfg.ffn_inner_dim
    if name == "encoder_frontend.embed.weight":
        return (cfg.vocab_size, d)
    if name == "encoder_frontend.pos_encoder.weight":
        return (cfg.model_max_seq_len, d)
    if name.startswith("pooler."):
        e, pf = cfg.edim, cfg.pooler_ffn_dim
        w = name.endswith("weight")
        if name == "pooler.decoder_frontend.embed.weight":
            return (1, e)
        if "layer_norm" in name:
            return (e,)
        if "ffn.inner_proj" in name:
            return (pf, e) if w else (pf,)
        if "ffn.output_proj" in name:
            return (e, pf) if w else (e,)
        if "encoder_decoder_attn.k_proj" in name or "encoder_decoder_attn.v_proj" in name:
            return (e, d) if w else (e,)     # kv_dim = model_dim (factory.py:207-211)
        return (e, e) if w else (e,)
    if name.endswith("layer_norm.weight") or name.endswith("layer_norm.bias"):
        return (d,)
    if "ffn.inner_proj" in name:
        return (f, d) if name.endswith("weight") else (f,)
    if "ffn.output_proj" in name:
        return (d, f) if name.endswith("weight") else (d,)
    return (d, d) if name.endswith("weight") else (d,)


def make_synthetic_params(cfg: OracleTextEncoderConfig, seed: int = 1234,
                          std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: Linear/Embedding ~ N(0, std^2),
    biases ~ N(0, std^2), LN weight 1 + N(0, std^2), LN bias N(0, std^2)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in param_names(cfg):
        t = torch.randn(param_shape(cfg, name), generator=g, dtype=torch.float32) * std
        if name.endswith("layer_norm.weight"):
            t = t + 1.0
        out[name] = t
    return out


def _layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _mha_generic(params, prefix, q_in, kv_in, heads, key_mask):
    """StandardMultiheadAttention(model_dim = q_in width, kv_dim = kv_in width): q_in [N, Tq, E], kv_in [N, Tk, Dk],
    key_mask [N, Tk] True = padding."""
    n, tq, e = q_in.shape
    tk = kv_in.shape[1]
    dh = e // heads
    q = F.linear(q_in, params[prefix + "q_proj.weight"], params[prefix + "q_proj.bias"]).view(n, tq, heads, dh).transpose(1, 2)
    k = F.linear(kv_in, params[prefix + "k_proj.weight"], params[prefix + "k_proj.bias"]).view(n, tk, heads, dh).transpose(1, 2)
    v = F.linear(kv_in, params[prefix + "v_proj.weight"], params[prefix + "v_proj.bias"]).view(n, tk, heads, dh).transpose(1, 2)
    att = torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5
    if key_mask is not None:
        att = att.masked_fill(key_mask[:, None, None, :], -torch.inf)
    y = torch.matmul(torch.softmax(att, dim=-1), v).transpose(1, 2).reshape(n, tq, e)
    return F.linear(y, params[prefix + "output_proj.weight"], params[prefix + "output_proj.bias"])


def attention_pooling(params, cfg: OracleTextEncoderConfig, enc: torch.Tensor, seq_lens: Optional[torch.Tensor]) -> torch.Tensor:
    """AttentionEncoderOutputPooler.__call__ (sonar/nn/encoder_pooler.py:72-95) as SonarTextEncoderFactory builds it
    (factory.py:155-226): one BOS token (bos_idx 0 of a 1-row embedding) through TransformerEmbeddingFrontend
    (x sqrt(embedding_dim), sinusoidal position 0, no legacy offset), `pooler_layers` StandardTransformerDecoderLayers
    (self-attention on that one token, cross-attention to the encoder output with kv_dim = model_dim, FFN) in the model's
    norm order, the decoder's final LayerNorm when pre-norm, projection_out.  PARITY UNPINNED against the reference
    (fairseq2 is not installable here); restated from the cited source."""
    n, s, _ = enc.shape
    e = cfg.edim
    x = params["pooler.decoder_frontend.embed.weight"][0].float() * math.sqrt(e) + sinusoidal_table(1, e)[0]
    x = x.expand(n, 1, e).clone()
    key_mask = None if seq_lens is None else torch.arange(s).unsqueeze(0) >= seq_lens.unsqueeze(1)
    pre = cfg.normalize_before
    for i in range(cfg.pooler_layers):
        p = f"pooler.decoder.layers.{i}."
        ln = lambda name, t: _layer_norm(t, params[p + name + ".weight"], params[p + name + ".bias"], cfg.ln_eps)
        if pre:
            h = ln("self_attn_layer_norm", x)
            x = x + _mha_generic(params, p + "self_attn.", h, h, cfg.pooler_heads, None)
            h = ln("encoder_decoder_attn_layer_norm", x)
            x = x + _mha_generic(params, p + "encoder_decoder_attn.", h, enc, cfg.pooler_heads, key_mask)
            h = ln("ffn_layer_norm", x)
            x = x + F.linear(F.relu(F.linear(h, params[p + "ffn.inner_proj.weight"], params[p + "ffn.inner_proj.bias"])),
                             params[p + "ffn.output_proj.weight"], params[p + "ffn.output_proj.bias"])
        else:
            x = ln("self_attn_layer_norm", x + _mha_generic(params, p + "self_attn.", x, x, cfg.pooler_heads, None))
            x = ln("encoder_decoder_attn_layer_norm",
                   x + _mha_generic(params, p + "encoder_decoder_attn.", x, enc, cfg.pooler_heads, key_mask))
            x = ln("ffn_layer_norm",
                   x + F.linear(F.relu(F.linear(x, params[p + "ffn.inner_proj.weight"], params[p + "ffn.inner_proj.bias"])),
                                params[p + "ffn.output_proj.weight"], params[p + "ffn.output_proj.bias"]))
    if pre:
        x = _layer_norm(x, params["pooler.decoder.layer_norm.weight"], params["pooler.decoder.layer_norm.bias"], cfg.ln_eps)
    return F.linear(x, params["pooler.projection_out.weight"], params["pooler.projection_out.bias"]).squeeze(1)


def static_pooling(seqs: torch.Tensor, seq_lens: Optional[torch.Tensor], pooling: str) -> torch.Tensor:
    """SonarTextTransformerEncoderModel.static_pooling
    (sonar/models/sonar_text/model.py:86-128). seq_lens None <=> no padding mask."""
    n, s = seqs.shape[:2]
    if pooling == "last":
        if seq_lens is None:
            return seqs[:, -1]
        return seqs[torch.arange(n), (seq_lens - 1).clip(0)]
    if seq_lens is not None:
        keep = torch.arange(s).unsqueeze(0) < seq_lens.unsqueeze(1)
        keep = keep.reshape(n, s, *([1] * (seqs.dim() - 2)))
    if pooling == "max":
        if seq_lens is not None:
            seqs = torch.where(keep, seqs, torch.full_like(seqs, -torch.inf))
        return seqs.max(dim=1).values
    if pooling == "mean":
        if seq_lens is not None:
            seqs = torch.where(keep, seqs, torch.zeros_like(seqs))
        summed = seqs.sum(dim=1)
        if seq_lens is None:
            return summed * (1.0 / (s + 1e-7))
        weights = 1.0 / (seq_lens.to(summed.dtype) + 1e-7)
        return torch.einsum("i...,i->i...", summed, weights)
    raise NotImplementedError(pooling)


@torch.inference_mode()
def text_encoder_forward(params: Dict[str, torch.Tensor], cfg: OracleTextEncoderConfig,
                         ids: torch.Tensor, seq_lens: Optional[torch.Tensor]):
    """SonarTextTransformerEncoderModel.forward (sonar/models/sonar_text/model.py:130-143).

    ids: int64 [N, S] right-padded (pad value is irrelevant to valid outputs);
    seq_lens: int [N] or None (batch not ragged, utils.py:18-21).
    Returns (encoded_seqs [N,S,d], sentence_embeddings [N,d]) in fp32.
    """
    d, h = cfg.model_dim, cfg.num_heads
    n, s = ids.shape
    if s > cfg.model_max_seq_len:
        raise ValueError("sequence longer than the positional table")
    dh = d // h
    # frontend: TransformerEmbeddingFrontend (factory.py:73-100): E[tok]*sqrt(d) + PE, no LN
    scale = 1.0 if cfg.no_scale_embedding else math.sqrt(d)
    x = params["encoder_frontend.embed.weight"][ids].float() * scale
    if not cfg.no_token_positional_embeddings:
        if cfg.learned_pos:   # LearnedPositionEncoder: rows 0.. of its table, no legacy offset (factory.py:81-85)
            pe = params["encoder_frontend.pos_encoder.weight"][:s].float()
        else:
            pe = sinusoidal_table(cfg.pos_offset + s, d)[cfg.pos_offset:]
        x = x + pe.unsqueeze(0)
    if cfg.layernorm_embedding:  # TransformerEmbeddingFrontend(layer_norm=True), factory.py:94-100
        x = _layer_norm(x, params["encoder_frontend.layer_norm.weight"], params["encoder_frontend.layer_norm.bias"], cfg.ln_eps)
    key_mask = None
    if seq_lens is not None:
        key_mask = torch.arange(s).unsqueeze(0) >= seq_lens.unsqueeze(1)  # True = pad
    for i in range(cfg.num_layers):
        p = f"encoder.layers.{i}."
        # StandardTransformerEncoderLayer, norm_order=PRE always (factory.py:122-128)
        r = x
        y = _layer_norm(x, params[p + "self_attn_layer_norm.weight"],
                        params[p + "self_attn_layer_norm.bias"], cfg.ln_eps)
        q = F.linear(y, params[p + "self_attn.q_proj.weight"], params[p + "self_attn.q_proj.bias"])
        k = F.linear(y, params[p + "self_attn.k_proj.weight"], params[p + "self_attn.k_proj.bias"])
        v = F.linear(y, params[p + "self_attn.v_proj.weight"], params[p + "self_attn.v_proj.bias"])
        q = q.view(n, s, h, dh).transpose(1, 2)
        k = k.view(n, s, h, dh).transpose(1, 2)
        v = v.view(n, s, h, dh).transpose(1, 2)
        att = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)  # factory.py:130-141 (SDPA)
        if key_mask is not None:
            att = att.masked_fill(key_mask[:, None, None, :], -torch.inf)
        att = torch.softmax(att, dim=-1)
        y = torch.matmul(att, v).transpose(1, 2).reshape(n, s, d)
        y = F.linear(y, params[p + "self_attn.output_proj.weight"], params[p + "self_attn.output_proj.bias"])
        x = r + y
        r = x
        y = _layer_norm(x, params[p + "ffn_layer_norm.weight"], params[p + "ffn_layer_norm.bias"], cfg.ln_eps)
        y = F.linear(y, params[p + "ffn.inner_proj.weight"], params[p + "ffn.inner_proj.bias"])
        y = F.relu(y)  # factory.py:143-153
        y = F.linear(y, params[p + "ffn.output_proj.weight"], params[p + "ffn.output_proj.bias"])
        x = r + y
    if cfg.normalize_before:  # StandardTransformerEncoder(norm_order=PRE) ends in its own LayerNorm (factory.py:107-109)
        x = _layer_norm(x, params["encoder.layer_norm.weight"], params["encoder.layer_norm.bias"], cfg.ln_eps)
    # model-level LayerNorm (factory.py:117, model.py:136-137)
    x = _layer_norm(x, params["layer_norm.weight"], params["layer_norm.bias"], cfg.ln_eps)
    if cfg.pooling == "attention":
        emb = attention_pooling(params, cfg, x, seq_lens)
    else:
        emb = static_pooling(x, seq_lens, cfg.pooling)
    return x, emb


def synthetic_batch(n: int, min_len: int, max_len: int, vocab_size: int, seed: int = 0,
                    lang_id: Optional[int] = None, pad_to: Optional[int] = None):
    """Synthetic token batch of SURVEY 8(d): ids = [lang] + randint(4, V') + [3], pad 0."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(min_len, max_len + 1, (n,), generator=g)
    s = int(pad_to or lens.max())
    hi = min(256001, vocab_size)
    lang = lang_id if lang_id is not None else min(256047, vocab_size - 1)
    ids = torch.zeros(n, s, dtype=torch.int64)
    for i, L in enumerate(lens.tolist()):
        body = torch.randint(4, hi, (max(L - 2, 0),), generator=g)
        seq = torch.cat([torch.tensor([lang]), body, torch.tensor([3])])[:L]
        ids[i, :L] = seq
    return ids, lens.to(torch.int32)
