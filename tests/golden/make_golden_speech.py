"""Generate tests/golden/conformer_twin.pt -- pins the oracle's conformer block and the pooler's
POST-norm decoder layer against independent HuggingFace implementations:

  * `Wav2Vec2ConformerEncoderLayer` (position_embeddings_type="relative"): a port of the fairseq
    conformer whose checkpoint names sonar/models/sonar_speech/handler.py:63-95 consumes
    (linear_pos / pos_bias_u / pos_bias_v -> sdpa.r_proj / u_bias / v_bias);
  * `BartDecoderLayer`: a POST-norm transformer decoder layer (self-attn, cross-attn, ReLU FFN),
    the structure sonar/models/sonar_speech/factory.py:102-121 builds for the attention pooler.

Run in the build container:  python tests/golden/make_golden_speech.py
"""
import os

import torch
from transformers import BartConfig, Wav2Vec2ConformerConfig
from transformers.models.bart.modeling_bart import BartDecoderLayer
from transformers.models.wav2vec2_conformer import modeling_wav2vec2_conformer as W

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conformer_twin.pt")
D, H, F, K, T, N = 64, 4, 128, 7, 23, 3


def randomize(mod, scale=0.2):
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if ("layer_norm" in name or "batch_norm" in name) and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(scale * torch.randn_like(p))
        for name, b in mod.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.3 * torch.randn_like(b))
            if name.endswith("running_var"):
                b.copy_(0.5 + torch.rand_like(b))


def main():
    torch.manual_seed(20240926)
    cfg = Wav2Vec2ConformerConfig(hidden_size=D, num_attention_heads=H, intermediate_size=F, hidden_act="swish",
                                  conv_depthwise_kernel_size=K, position_embeddings_type="relative",
                                  max_source_positions=64, hidden_dropout=0.0, attention_dropout=0.0,
                                  activation_dropout=0.0, conformer_conv_dropout=0.0)
    layer = W.Wav2Vec2ConformerEncoderLayer(cfg).eval().float()
    relpos = W.Wav2Vec2ConformerRelPositionalEmbedding(cfg)
    randomize(layer)
    x = torch.randn(N, T, D)
    with torch.no_grad():
        rel = relpos(x)
        y, _ = layer(x, attention_mask=None, relative_position_embeddings=rel)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    # fairseq2-style names of ONE conformer block (handler.py:63-95 applied to HF/fairseq names)
    ren = {
        "ffn1_layer_norm": "ffn1_layer_norm", "ffn2_layer_norm": "ffn2_layer_norm",
        "ffn1.intermediate_dense": "ffn1.inner_proj", "ffn1.output_dense": "ffn1.output_proj",
        "ffn2.intermediate_dense": "ffn2.inner_proj", "ffn2.output_dense": "ffn2.output_proj",
        "self_attn_layer_norm": "self_attn_layer_norm",
        "self_attn.linear_q": "self_attn.q_proj", "self_attn.linear_k": "self_attn.k_proj",
        "self_attn.linear_v": "self_attn.v_proj", "self_attn.linear_out": "self_attn.output_proj",
        "self_attn.linear_pos": "self_attn.sdpa.r_proj", "self_attn.pos_bias_u": "self_attn.sdpa.u_bias",
        "self_attn.pos_bias_v": "self_attn.sdpa.v_bias",
        "conv_module.layer_norm": "conv_layer_norm", "conv_module.pointwise_conv1": "conv.pointwise_conv1",
        "conv_module.depthwise_conv": "conv.depthwise_conv", "conv_module.batch_norm": "conv.batch_norm",
        "conv_module.pointwise_conv2": "conv.pointwise_conv2", "final_layer_norm": "layer_norm",
    }
    block = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        for old, new in sorted(ren.items(), key=lambda kv: -len(kv[0])):
            if k.startswith(old):
                block["encoder.layers.0." + new + k[len(old):]] = v
                break
        else:
            raise KeyError(k)

    # --- POST-norm decoder layer twin (pooler) ---
    bcfg = BartConfig(d_model=D, decoder_attention_heads=H, decoder_ffn_dim=F, activation_function="relu",
                      dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    dl = BartDecoderLayer(bcfg).eval().float()
    randomize(dl)
    q = torch.randn(N, 1, D)
    enc = torch.randn(N, T, D)
    lens = torch.tensor([T, 9, 1])
    pad = torch.arange(T).unsqueeze(0) >= lens.unsqueeze(1)
    enc_mask = torch.zeros(N, 1, 1, T).masked_fill(pad[:, None, None, :], torch.finfo(torch.float32).min)
    with torch.no_grad():
        out = dl(q, attention_mask=None, encoder_hidden_states=enc, encoder_attention_mask=enc_mask)
    dout = out[0] if isinstance(out, tuple) else out
    dren = {"self_attn.out_proj": "self_attn.output_proj", "self_attn.": "self_attn.",
            "self_attn_layer_norm": "self_attn_layer_norm",
            "encoder_attn.out_proj": "encoder_decoder_attn.output_proj", "encoder_attn.": "encoder_decoder_attn.",
            "encoder_attn_layer_norm": "encoder_decoder_attn_layer_norm",
            "fc1": "ffn.inner_proj", "fc2": "ffn.output_proj", "final_layer_norm": "ffn_layer_norm"}
    pool = {}
    for k, v in dl.state_dict().items():
        for old, new in sorted(dren.items(), key=lambda kv: -len(kv[0])):
            if k.startswith(old):
                pool["encoder_pooler.decoder.layers.0." + new + k[len(old):]] = v.clone()
                break
        else:
            raise KeyError(k)

    torch.save({"dims": dict(model_dim=D, num_heads=H, ffn_inner_dim=F, conv_kernel=K),
                "block_params": block, "block_in": x, "block_out": y, "rel_pos": rel[0],
                "pooler_params": pool, "pooler_q": q, "pooler_enc": enc, "pooler_lens": lens, "pooler_out": dout}, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
