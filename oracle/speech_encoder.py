"""ORACLE -- test infrastructure only.  Never imported by the product package.

CPU / plain-PyTorch fp32 restatement of the reference's SpeechToEmbedding hot path
(facebookresearch/SONAR v0.4.0): Kaldi-style log-mel filterbank front-end, the
w2v-BERT conformer encoder and SONAR's attention pooler
(sonar/inference_pipelines/speech.py:283-308,402-474; sonar/models/sonar_speech/
{config.py:61-77, factory.py:53-152, model.py:59-77}; sonar/nn/encoder_pooler.py:47-89).

Pinning status:
  * the conformer block (macaron FFNs, Transformer-XL/ESPnet relative-position attention,
    GLU + depthwise-conv + BatchNorm module) is pinned against HuggingFace
    `Wav2Vec2ConformerEncoderLayer` (position_embeddings_type="relative"), a port of the
    same fairseq module whose checkpoint keys sonar/models/sonar_speech/handler.py:63-95
    consumes -> tests/golden/conformer_twin.pt;
  * the pooler's POST-norm decoder layer is pinned against HuggingFace `BartDecoderLayer`;
  * the filterbank restates kaldi-native-fbank / Kaldi `compute-fbank-feats` defaults as used
    by fairseq2's WaveformToFbankConverter (un-vendored).  fairseq2n / kaldi are absent,
    so it is pinned against an independent implementation of the same front end for the same
    w2v-BERT encoder family, HuggingFace `SeamlessM4TFeatureExtractor`: log-mel features, the
    per-utterance standardisation and the 2-frame stacking order agree to 2e-3 absolute on values
    in [5, 30] (mean 2e-5) -> tests/golden/fbank_seamless_twin.pt (make_golden_fbank.py);
  * the whole path waveform -> embedding (stacking order, frontend, block stack, moved LayerNorm,
    pooler query and layers, projection) agrees to 1e-6 (1 - cos) with a composition of those
    independent implementations on a GPU-sized twin -> tests/golden/speech_e2e_twin.pt;
  * the reference's real-checkpoint goldens (tests/integration_tests/data/speech_embedding.pt,
    test_sonar_speech_pipeline_models.py:28-40) need the checkpoint: PARITY UNPINNED.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .text_encoder import sinusoidal_table


# ----------------------------------------------------------------------------- filterbank
def povey_window(n: int) -> torch.Tensor:
    i = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2 * math.pi * i / (n - 1))).pow(0.85).float()


def mel_banks(num_bins: int = 80, n_fft: int = 512, sample_rate: float = 16000.0, low_freq: float = 20.0,
              high_freq: float = 0.0) -> torch.Tensor:
    """Kaldi MelBanks: [num_bins, n_fft/2] triangular weights in the mel domain
    (mel(f) = 1127 ln(1 + f/700)); the Nyquist bin is not used."""
    nyquist = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    nbins = n_fft // 2
    fft_bin_width = sample_rate / n_fft
    out = torch.zeros(num_bins, nbins, dtype=torch.float64)
    for b in range(num_bins):
        left = mel_low + b * delta
        center = left + delta
        right = center + delta
        for k in range(nbins):
            m = mel(fft_bin_width * k)
            if left < m < right:
                out[b, k] = (m - left) / (center - left) if m <= center else (right - m) / (right - center)
    return out.float()


def kaldi_fbank(waveform: torch.Tensor, sample_rate: float = 16000.0, num_mel_bins: int = 80,
                waveform_scale: float = 2.0 ** 15, standardize: bool = True) -> torch.Tensor:
    """[T] float waveform -> [frames, num_mel_bins] log-mel features.
    Kaldi fbank defaults as configured by the reference (speech.py:283-290): 25 ms / 10 ms frames,
    dither 0, remove DC, pre-emphasis 0.97, povey window, FFT 512, 80 mel bins from 20 Hz,
    power spectrum, natural log floored at FLT_EPSILON, snip_edges; then per-utterance
    standardisation over time with the unbiased std."""
    x = waveform.float() * waveform_scale
    win, shift, n_fft = int(sample_rate * 0.025), int(sample_rate * 0.010), 512
    if x.numel() < win:
        return torch.zeros(0, num_mel_bins)
    frames = x.unfold(0, win, shift).clone()                    # snip_edges: 1 + (T - 400) // 160
    frames = frames - frames.mean(dim=1, keepdim=True)           # remove_dc_offset
    prev = torch.cat([frames[:, :1], frames[:, :-1]], dim=1)     # x[i] -= 0.97 x[i-1]; x[0] -= 0.97 x[0]
    frames = frames - 0.97 * prev
    frames = frames * povey_window(win)
    spec = torch.fft.rfft(F.pad(frames, (0, n_fft - win)), n=n_fft)
    power = spec.real ** 2 + spec.imag ** 2                      # [frames, 257]
    mel = power[:, : n_fft // 2] @ mel_banks(num_mel_bins, n_fft, sample_rate).T
    fb = torch.log(torch.clamp(mel, min=torch.finfo(torch.float32).eps))
    if standardize and fb.shape[0] > 1:
        std, mean = torch.std_mean(fb, dim=0)                    # unbiased, over time
        fb = (fb - mean) / std
    return fb


# ----------------------------------------------------------------------------- model config
@dataclass
class OracleSpeechEncoderConfig:
    """w2v-BERT "600m" encoder + SONAR pooler (sonar_speech/config.py:61-77)."""

    model_dim: int = 1024
    feature_dim: int = 160            # 2 stacked 80-bin frames
    num_layers: int = 24
    num_heads: int = 16
    ffn_inner_dim: int = 4096
    conv_kernel: int = 31
    pooler_layers: int = 3            # "english" arch; "non_english" has 6
    pooler_heads: int = 16
    pooler_ffn_dim: int = 4096
    pooler_vocab: int = 1024          # Embedding(num_embeddings=model_dim), factory.py:94-100
    bos_idx: int = 2
    ln_eps: float = 1e-5
    bn_eps: float = 1e-5


def param_shapes(cfg: OracleSpeechEncoderConfig) -> Dict[str, Tuple[int, ...]]:
    """fairseq2-style names after sonar/models/sonar_speech/handler.py:63-110."""
    d, f, h = cfg.model_dim, cfg.ffn_inner_dim, cfg.num_heads
    s: Dict[str, Tuple[int, ...]] = {
        "encoder_frontend.post_extract_layer_norm.weight": (cfg.feature_dim,),
        "encoder_frontend.post_extract_layer_norm.bias": (cfg.feature_dim,),
        "encoder_frontend.model_dim_proj.weight": (d, cfg.feature_dim),
        "encoder_frontend.model_dim_proj.bias": (d,),
        "layer_norm.weight": (d,), "layer_norm.bias": (d,),
        "encoder_pooler.decoder_frontend.embed.weight": (cfg.pooler_vocab, d),
        "encoder_pooler.projection_out.weight": (d, d),
    }
    for i in range(cfg.num_layers):
        p = f"encoder.layers.{i}."
        for ln in ("ffn1_layer_norm", "self_attn_layer_norm", "conv_layer_norm", "ffn2_layer_norm", "layer_norm"):
            s[p + ln + ".weight"] = (d,)
            s[p + ln + ".bias"] = (d,)
        for ffn in ("ffn1", "ffn2"):
            s[p + ffn + ".inner_proj.weight"] = (f, d)
            s[p + ffn + ".inner_proj.bias"] = (f,)
            s[p + ffn + ".output_proj.weight"] = (d, f)
            s[p + ffn + ".output_proj.bias"] = (d,)
        for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
            s[p + f"self_attn.{lin}.weight"] = (d, d)
            s[p + f"self_attn.{lin}.bias"] = (d,)
        s[p + "self_attn.sdpa.r_proj.weight"] = (d, d)
        s[p + "self_attn.sdpa.u_bias"] = (h, d // h)
        s[p + "self_attn.sdpa.v_bias"] = (h, d // h)
        s[p + "conv.pointwise_conv1.weight"] = (2 * d, d, 1)
        s[p + "conv.depthwise_conv.weight"] = (d, 1, cfg.conv_kernel)
        for bn in ("weight", "bias", "running_mean", "running_var"):
            s[p + "conv.batch_norm." + bn] = (d,)
        s[p + "conv.pointwise_conv2.weight"] = (d, d, 1)
    for i in range(cfg.pooler_layers):
        p = f"encoder_pooler.decoder.layers.{i}."
        for att in ("self_attn", "encoder_decoder_attn"):
            for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
                s[p + f"{att}.{lin}.weight"] = (d, d)
                s[p + f"{att}.{lin}.bias"] = (d,)
            s[p + att + "_layer_norm.weight"] = (d,)
            s[p + att + "_layer_norm.bias"] = (d,)
        s[p + "ffn.inner_proj.weight"] = (cfg.pooler_ffn_dim, d)
        s[p + "ffn.inner_proj.bias"] = (cfg.pooler_ffn_dim,)
        s[p + "ffn.output_proj.weight"] = (d, cfg.pooler_ffn_dim)
        s[p + "ffn.output_proj.bias"] = (d,)
        s[p + "ffn_layer_norm.weight"] = (d,)
        s[p + "ffn_layer_norm.bias"] = (d,)
    return s


def make_synthetic_params(cfg: OracleSpeechEncoderConfig, seed: int = 99, std: float = 0.05) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        t = torch.randn(shape, generator=g, dtype=torch.float32) * std
        if name.endswith("layer_norm.weight") or name.endswith("batch_norm.weight"):
            t = t + 1.0
        if name.endswith("running_var"):
            t = t.abs() + 0.5
        if name.endswith("depthwise_conv.weight"):
            t = t * 4
        out[name] = t
    return out


# ----------------------------------------------------------------------------- conformer
def rel_pos_encoding(seq_len: int, dim: int) -> torch.Tensor:
    """ESPnet / fairseq RelPositionalEncoding: [2S-1, dim], row idx <-> relative position
    (S-1 - idx), sin on even and cos on odd feature indices."""
    pos = torch.arange(seq_len - 1, -seq_len, -1, dtype=torch.float32).unsqueeze(1)   # S-1 ... -(S-1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe = torch.zeros(2 * seq_len - 1, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _ln(x, p, name, eps):
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], eps)


def _ffn(x, p, name, act):
    h = F.linear(x, p[name + ".inner_proj.weight"], p[name + ".inner_proj.bias"])
    return F.linear(act(h), p[name + ".output_proj.weight"], p[name + ".output_proj.bias"])


def rel_pos_attention(x, p, prefix, heads, key_pad: Optional[torch.Tensor]):
    """StandardMultiheadAttention + RelativePositionSDPA (Transformer-XL form):
    scores[i,j] = ((q_i + u) . k_j + (q_i + v) . r[S-1 + j - i]) / sqrt(dh)."""
    n, s, d = x.shape
    dh = d // heads
    q = F.linear(x, p[prefix + "q_proj.weight"], p[prefix + "q_proj.bias"]).view(n, s, heads, dh)
    k = F.linear(x, p[prefix + "k_proj.weight"], p[prefix + "k_proj.bias"]).view(n, s, heads, dh)
    v = F.linear(x, p[prefix + "v_proj.weight"], p[prefix + "v_proj.bias"]).view(n, s, heads, dh)
    r = F.linear(rel_pos_encoding(s, d), p[prefix + "sdpa.r_proj.weight"]).view(2 * s - 1, heads, dh)
    qu = (q + p[prefix + "sdpa.u_bias"]).transpose(1, 2)            # [n,h,s,dh]
    qv = (q + p[prefix + "sdpa.v_bias"]).transpose(1, 2)
    ac = qu @ k.permute(0, 2, 3, 1)                                   # [n,h,s,s]
    bd_full = qv @ r.permute(1, 2, 0)                                 # [n,h,s,2s-1]
    idx = (s - 1) + torch.arange(s).unsqueeze(0) - torch.arange(s).unsqueeze(1)   # [i,j] -> S-1 + j - i
    bd = bd_full.gather(3, idx.expand(n, heads, s, s))
    att = (ac + bd) * dh ** -0.5
    if key_pad is not None:
        att = att.masked_fill(key_pad[:, None, None, :], -torch.inf)
    att = torch.softmax(att, dim=-1)
    y = (att @ v.transpose(1, 2)).transpose(1, 2).reshape(n, s, d)
    return F.linear(y, p[prefix + "output_proj.weight"], p[prefix + "output_proj.bias"])


def conv_module(x, p, prefix, cfg, key_pad: Optional[torch.Tensor]):
    """ConformerConvolution: zero pads, pointwise(d->2d) -> GLU -> depthwise k=31 'same' ->
    BatchNorm (eval) -> SiLU -> pointwise(d->d); no conv biases."""
    if key_pad is not None:
        x = x.masked_fill(key_pad.unsqueeze(-1), 0.0)
    y = x.transpose(1, 2)
    y = F.conv1d(y, p[prefix + "pointwise_conv1.weight"])
    y = F.glu(y, dim=1)
    k = cfg.conv_kernel
    y = F.conv1d(y, p[prefix + "depthwise_conv.weight"], padding=(k - 1) // 2, groups=y.shape[1])
    y = F.batch_norm(y, p[prefix + "batch_norm.running_mean"], p[prefix + "batch_norm.running_var"],
                     p[prefix + "batch_norm.weight"], p[prefix + "batch_norm.bias"], training=False, eps=cfg.bn_eps)
    y = F.silu(y)
    y = F.conv1d(y, p[prefix + "pointwise_conv2.weight"])
    return y.transpose(1, 2)


def conformer_block(x, p, i, cfg, key_pad):
    pre = f"encoder.layers.{i}."
    x = x + 0.5 * _ffn(_ln(x, p, pre + "ffn1_layer_norm", cfg.ln_eps), p, pre + "ffn1", F.silu)
    x = x + rel_pos_attention(_ln(x, p, pre + "self_attn_layer_norm", cfg.ln_eps), p, pre + "self_attn.", cfg.num_heads, key_pad)
    x = x + conv_module(_ln(x, p, pre + "conv_layer_norm", cfg.ln_eps), p, pre + "conv.", cfg, key_pad)
    x = x + 0.5 * _ffn(_ln(x, p, pre + "ffn2_layer_norm", cfg.ln_eps), p, pre + "ffn2", F.silu)
    return _ln(x, p, pre + "layer_norm", cfg.ln_eps)


# ----------------------------------------------------------------------------- pooler
def _plain_mha(p, prefix, q_in, kv_in, heads, key_pad):
    n, tq, d = q_in.shape
    tk = kv_in.shape[1]
    dh = d // heads
    q = F.linear(q_in, p[prefix + "q_proj.weight"], p[prefix + "q_proj.bias"]).view(n, tq, heads, dh).transpose(1, 2)
    k = F.linear(kv_in, p[prefix + "k_proj.weight"], p[prefix + "k_proj.bias"]).view(n, tk, heads, dh).transpose(1, 2)
    v = F.linear(kv_in, p[prefix + "v_proj.weight"], p[prefix + "v_proj.bias"]).view(n, tk, heads, dh).transpose(1, 2)
    att = (q @ k.transpose(-1, -2)) * dh ** -0.5
    if key_pad is not None:
        att = att.masked_fill(key_pad[:, None, None, :], -torch.inf)
    y = (torch.softmax(att, dim=-1) @ v).transpose(1, 2).reshape(n, tq, d)
    return F.linear(y, p[prefix + "output_proj.weight"], p[prefix + "output_proj.bias"])


def pooler_layer(x, p, i, cfg, enc, key_pad):
    """StandardTransformerDecoderLayer, norm_order POST (config.py:72, factory.py:110-121)."""
    pre = f"encoder_pooler.decoder.layers.{i}."
    x = _ln(x + _plain_mha(p, pre + "self_attn.", x, x, cfg.pooler_heads, None), p, pre + "self_attn_layer_norm", cfg.ln_eps)
    x = _ln(x + _plain_mha(p, pre + "encoder_decoder_attn.", x, enc, cfg.pooler_heads, key_pad), p,
            pre + "encoder_decoder_attn_layer_norm", cfg.ln_eps)
    return _ln(x + _ffn(x, p, pre + "ffn", F.relu), p, pre + "ffn_layer_norm", cfg.ln_eps)


def attention_pooler(p, cfg, enc, key_pad):
    """AttentionEncoderOutputPooler.__call__ (sonar/nn/encoder_pooler.py:70-89)."""
    n, _, d = enc.shape
    x = p["encoder_pooler.decoder_frontend.embed.weight"][cfg.bos_idx].float() * math.sqrt(d)
    x = (x + sinusoidal_table(1, d)[0]).expand(n, 1, d)   # position 0, no legacy offset (factory.py:88-92)
    for i in range(cfg.pooler_layers):
        x = pooler_layer(x, p, i, cfg, enc, key_pad)
    return F.linear(x, p["encoder_pooler.projection_out.weight"]).squeeze(1)


# ----------------------------------------------------------------------------- full model
@torch.inference_mode()
def speech_encoder_forward(p: Dict[str, torch.Tensor], cfg: OracleSpeechEncoderConfig, fbank: torch.Tensor,
                           fbank_lens: Optional[torch.Tensor]):
    """SonarSpeechEncoderModel.forward (sonar/models/sonar_speech/model.py:59-77).
    fbank: [N, T, 80] zero-padded, T even (Collater pad_to_multiple=2, speech.py:444);
    fbank_lens: frames per clip or None.  Returns (encoder_output [N,T/2,d], embeddings [N,d])."""
    n, t, nb = fbank.shape
    assert t % 2 == 0 and 2 * nb == cfg.feature_dim
    x = fbank.float().reshape(n, t // 2, 2 * nb)               # stack 2 frames
    key_pad = None
    if fbank_lens is not None:
        lens = fbank_lens // 2
        key_pad = torch.arange(t // 2).unsqueeze(0) >= lens.unsqueeze(1)
        if not key_pad.any():
            key_pad = None
    x = _ln(x, p, "encoder_frontend.post_extract_layer_norm", cfg.ln_eps)
    x = F.linear(x, p["encoder_frontend.model_dim_proj.weight"], p["encoder_frontend.model_dim_proj.bias"])
    for i in range(cfg.num_layers):
        x = conformer_block(x, p, i, cfg, key_pad)
    x = _ln(x, p, "layer_norm", cfg.ln_eps)                     # SONAR's moved LayerNorm (handler.py:102-108)
    emb = attention_pooler(p, cfg, x, key_pad)
    return x, emb
