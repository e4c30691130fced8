"""Generate tests/golden/speech_e2e_twin.pt -- the speech path END TO END (waveform -> sentence
embedding) computed from independent implementations only, at the smallest size the HIP engine accepts
(d = 256, 4 heads of 64):

  waveform --HF SeamlessM4TFeatureExtractor--> standardised 2-frame-stacked log-mel [T', 160]
           --torch.nn.LayerNorm(160) + torch.nn.Linear(160, d)--> frontend (sonar_speech/factory.py:53-100)
           --L x HF Wav2Vec2ConformerEncoderLayer (relative positions) + torch.nn.LayerNorm--> encoder
           --query E[bos]*sqrt(d) + PE[0]; P x HF BartDecoderLayer (POST norm); bias-free Linear--> pooler
             (sonar/nn/encoder_pooler.py:70-89, sonar_speech/factory.py:88-152)

The weights are `oracle.speech_encoder.make_synthetic_params` (seeded, fairseq2 names after
sonar_speech/handler.py:63-110), loaded into the HF modules by the inverse of that key map; the GPU test
rebuilds them from the same seed, so the fixture holds only the waveforms and the outputs.  One clip
per forward pass: no padding masks are involved.

Run in the build container:  python tests/golden/make_golden_speech_e2e.py
"""
import math
import os
import sys

import numpy as np
import torch
from transformers import BartConfig, SeamlessM4TFeatureExtractor, Wav2Vec2ConformerConfig
from transformers.models.bart.modeling_bart import BartDecoderLayer
from transformers.models.wav2vec2_conformer import modeling_wav2vec2_conformer as W

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import speech_encoder as OS  # noqa: E402  (only make_synthetic_params / config: the weights)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "speech_e2e_twin.pt")
CFG = dict(model_dim=256, num_layers=2, num_heads=4, ffn_inner_dim=512, conv_kernel=31, pooler_layers=2,
           pooler_heads=4, pooler_ffn_dim=512, pooler_vocab=256)
SEED, STD = 1234, 0.06

CONF = {"ffn1_layer_norm": "ffn1_layer_norm", "ffn2_layer_norm": "ffn2_layer_norm",
        "ffn1.inner_proj": "ffn1.intermediate_dense", "ffn1.output_proj": "ffn1.output_dense",
        "ffn2.inner_proj": "ffn2.intermediate_dense", "ffn2.output_proj": "ffn2.output_dense",
        "self_attn_layer_norm": "self_attn_layer_norm", "self_attn.q_proj": "self_attn.linear_q",
        "self_attn.k_proj": "self_attn.linear_k", "self_attn.v_proj": "self_attn.linear_v",
        "self_attn.output_proj": "self_attn.linear_out", "self_attn.sdpa.r_proj": "self_attn.linear_pos",
        "self_attn.sdpa.u_bias": "self_attn.pos_bias_u", "self_attn.sdpa.v_bias": "self_attn.pos_bias_v",
        "conv_layer_norm": "conv_module.layer_norm", "conv.pointwise_conv1": "conv_module.pointwise_conv1",
        "conv.depthwise_conv": "conv_module.depthwise_conv", "conv.batch_norm": "conv_module.batch_norm",
        "conv.pointwise_conv2": "conv_module.pointwise_conv2", "layer_norm": "final_layer_norm"}
POOL = {"self_attn.output_proj": "self_attn.out_proj", "self_attn.": "self_attn.",
        "self_attn_layer_norm": "self_attn_layer_norm", "encoder_decoder_attn.output_proj": "encoder_attn.out_proj",
        "encoder_decoder_attn.": "encoder_attn.", "encoder_decoder_attn_layer_norm": "encoder_attn_layer_norm",
        "ffn.inner_proj": "fc1", "ffn.output_proj": "fc2", "ffn_layer_norm": "final_layer_norm"}


def rename(params, prefix, table):
    out = {}
    for k, v in params.items():
        if not k.startswith(prefix):
            continue
        rest = k[len(prefix):]
        for old, new in sorted(table.items(), key=lambda kv: -len(kv[0])):
            if rest.startswith(old):
                out[new + rest[len(old):]] = v
                break
        else:
            raise KeyError(k)
    return out


def main():
    cfg = OS.OracleSpeechEncoderConfig(**CFG)
    p = OS.make_synthetic_params(cfg, seed=SEED, std=STD)
    d = cfg.model_dim
    wcfg = Wav2Vec2ConformerConfig(hidden_size=d, num_attention_heads=cfg.num_heads, intermediate_size=cfg.ffn_inner_dim,
                                   hidden_act="swish", conv_depthwise_kernel_size=cfg.conv_kernel,
                                   position_embeddings_type="relative", max_source_positions=512, hidden_dropout=0.0,
                                   attention_dropout=0.0, activation_dropout=0.0, conformer_conv_dropout=0.0,
                                   layer_norm_eps=cfg.ln_eps)
    layers = []
    for i in range(cfg.num_layers):
        lay = W.Wav2Vec2ConformerEncoderLayer(wcfg).eval().float()
        res = lay.load_state_dict(rename(p, f"encoder.layers.{i}.", CONF), strict=False)
        assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys), res
        layers.append(lay)
    relpos = W.Wav2Vec2ConformerRelPositionalEmbedding(wcfg)
    bcfg = BartConfig(d_model=d, decoder_attention_heads=cfg.pooler_heads, decoder_ffn_dim=cfg.pooler_ffn_dim,
                      activation_function="relu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    pool = []
    for i in range(cfg.pooler_layers):
        dl = BartDecoderLayer(bcfg).eval().float()
        dl.load_state_dict(rename(p, f"encoder_pooler.decoder.layers.{i}.", POOL), strict=True)
        pool.append(dl)
    ln_in = torch.nn.LayerNorm(cfg.feature_dim, eps=cfg.ln_eps)
    ln_in.load_state_dict({"weight": p["encoder_frontend.post_extract_layer_norm.weight"],
                           "bias": p["encoder_frontend.post_extract_layer_norm.bias"]})
    proj = torch.nn.Linear(cfg.feature_dim, d)
    proj.load_state_dict({"weight": p["encoder_frontend.model_dim_proj.weight"], "bias": p["encoder_frontend.model_dim_proj.bias"]})
    ln_out = torch.nn.LayerNorm(d, eps=cfg.ln_eps)
    ln_out.load_state_dict({"weight": p["layer_norm.weight"], "bias": p["layer_norm.bias"]})
    fe = SeamlessM4TFeatureExtractor()

    g = torch.Generator().manual_seed(8)
    waves, embs, encs = [], [], []
    for n in (16000 * 7 // 10 + 33, 16000 + 5, 16000 * 3 // 2 + 391):
        t = torch.arange(n) / 16000.0
        wav = 0.25 * (torch.rand(n, generator=g) * 2 - 1) + 0.3 * torch.sin(2 * math.pi * (300 + 40 * len(waves)) * t)
        feats = fe(wav.numpy().astype(np.float32), sampling_rate=16000, return_tensors="pt", padding=False,
                   do_normalize_per_mel_bins=True)["input_features"].float()          # [1, T', 160]
        with torch.no_grad():
            x = proj(ln_in(feats))
            rel = relpos(x)
            for lay in layers:
                x = lay(x, attention_mask=None, relative_position_embeddings=rel)[0]
            enc = ln_out(x)
            q = p["encoder_pooler.decoder_frontend.embed.weight"][cfg.bos_idx] * math.sqrt(d)
            q = q + torch.cat([torch.zeros(d // 2), torch.ones(d // 2)])               # sinusoid at position 0: sin 0 | cos 1
            q = q.view(1, 1, d)
            for dl in pool:
                out = dl(q, attention_mask=None, encoder_hidden_states=enc, encoder_attention_mask=None)
                q = out[0] if isinstance(out, tuple) else out
            emb = torch.nn.functional.linear(q, p["encoder_pooler.projection_out.weight"]).view(d)
        waves.append(wav)
        embs.append(emb)
        encs.append(enc[0])
    torch.save({"config": CFG, "seed": SEED, "std": STD, "waveforms": waves, "embeddings": torch.stack(embs),
                "encoder_out_first_clip": encs[0]}, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; frames", [e.shape[0] for e in encs])


if __name__ == "__main__":
    main()
