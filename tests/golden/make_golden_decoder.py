"""Generate tests/golden/m2m100_decoder_twin.pt -- pins the oracle's DECODER stack against
HuggingFace `M2M100Decoder` (same architecture and fairseq parameter names that
sonar/models/sonar_text/handler.py:139-159 consumes).  The conditioning sentence embedding
is fed as a length-1 encoder output, exactly as SonarEncoderDecoderModel.encode does
(sonar/models/sonar_translation/model.py:48-53).

Run in the build container:  python tests/golden/make_golden_decoder.py
"""
import os

import torch
from transformers import M2M100Config
from transformers.models.m2m_100.modeling_m2m_100 import M2M100Decoder

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "m2m100_decoder_twin.pt")
D, H, F, L, V, MAXPOS = 64, 4, 128, 2, 200, 64


def main():
    torch.manual_seed(20240925)
    cfg = M2M100Config(vocab_size=V, d_model=D, decoder_layers=L, decoder_attention_heads=H,
                       decoder_ffn_dim=F, encoder_layers=1, encoder_attention_heads=H, encoder_ffn_dim=F,
                       activation_function="relu", scale_embedding=True, max_position_embeddings=MAXPOS,
                       dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, decoder_layerdrop=0.0,
                       pad_token_id=1, bos_token_id=0, eos_token_id=2)
    dec = M2M100Decoder(cfg).eval().float()
    with torch.no_grad():
        for name, p in dec.named_parameters():
            if "layer_norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(0.15 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in dec.state_dict().items() if not k.startswith("embed_positions")}
    sd["output_projection.weight"] = sd["embed_tokens.weight"].clone()   # tied in fairseq checkpoints
    sd["version"] = torch.tensor([3.0])
    sd["embed_positions._float_tensor"] = torch.zeros(1)
    ckpt = {"state_dict": sd}

    g = torch.Generator().manual_seed(9)
    n, t = 4, 9
    emb = torch.randn(n, D, generator=g) * 0.5
    prev = torch.randint(4, V, (n, t), generator=g)          # SONAR ids
    prev[:, 0] = 3                                            # </s> first, as the decoder prompt
    prev[1, 3] = 1                                            # an <unk>
    prev[2, 4] = 2                                            # a <s>
    to_hf = torch.arange(V)
    to_hf[0], to_hf[1], to_hf[2], to_hf[3] = 1, 3, 0, 2
    with torch.no_grad():
        hid = dec(input_ids=to_hf[prev], encoder_hidden_states=emb.unsqueeze(1)).last_hidden_state
        logits_hf = torch.nn.functional.linear(hid, dec.embed_tokens.weight)   # tied projection, HF id order
    # columns back to SONAR id order: column j (SONAR id) = HF column to_hf[j]
    logits = logits_hf[..., to_hf]
    torch.save({"config": dict(model_dim=D, num_heads=H, ffn_inner_dim=F, num_layers=L, vocab_size=V,
                               max_seq_len=MAXPOS - 2),
                "checkpoint": ckpt, "embeddings": emb, "prev_tokens": prev, "logits": logits}, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
