"""Generate tests/golden/m2m100_greedy_twin.pt -- pins the oracle's GENERATION LOOP (prompt
forcing, EOS stop, min-length EOS suppression, incremental decoding) against HuggingFace
`M2M100ForConditionalGeneration.generate(num_beams=1, do_sample=False)`, an implementation
independent of both fairseq2 and this repo.  The sentence embedding is fed as a length-1 encoder
output (sonar/models/sonar_translation/model.py:48-53) and the decoder prompt is `[</s>, lang]`
(sonar/inference_pipelines/text.py:209-215 via the NLLB target-language prefix).

The embedding is small against the layer weights and its EOS row scaled up so that the tied
projection does not just echo the previous token and the 16 sentences stop at different lengths
(some never, which exercises the length cap).  Greedy only: HF's beam scorer normalises
and finalises hypotheses differently from fairseq2's BeamSearchSeq2SeqGenerator, so beam>1 stays
pinned by properties (tests/test_oracle_decoder_cpu.py) rather than by this twin.

Run in the build container:  python tests/golden/make_golden_generate.py
"""
import os

import torch
from transformers import M2M100Config, M2M100ForConditionalGeneration
from transformers.modeling_outputs import BaseModelOutput

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "m2m100_greedy_twin.pt")
D, H, F, L, V, MAXPOS = 64, 4, 128, 2, 200, 64
N, MAX_NEW = 16, 14


def main():
    torch.manual_seed(1)
    cfg = M2M100Config(vocab_size=V, d_model=D, decoder_layers=L, decoder_attention_heads=H,
                       decoder_ffn_dim=F, encoder_layers=1, encoder_attention_heads=H, encoder_ffn_dim=F,
                       activation_function="relu", scale_embedding=True, max_position_embeddings=MAXPOS,
                       dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, decoder_layerdrop=0.0,
                       pad_token_id=1, bos_token_id=0, eos_token_id=2, tie_word_embeddings=True)
    m = M2M100ForConditionalGeneration(cfg).eval().float()
    dec = m.model.decoder
    with torch.no_grad():
        for name, p in dec.named_parameters():
            if "layer_norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif "embed_tokens" in name:
                p.copy_(0.03 * torch.randn_like(p))
            else:
                p.copy_(0.3 * torch.randn_like(p))
        dec.embed_tokens.weight[2] *= 1.8                # HF id 2 = </s>
    m.lm_head.weight = dec.embed_tokens.weight            # tied, as TiedProjection (factory.py:306-307)

    sd = {k: v.clone() for k, v in dec.state_dict().items() if not k.startswith("embed_positions")}
    sd["output_projection.weight"] = sd["embed_tokens.weight"].clone()   # tied in fairseq checkpoints
    sd["version"] = torch.tensor([3.0])
    sd["embed_positions._float_tensor"] = torch.zeros(1)

    to_hf = torch.arange(V)
    to_hf[0], to_hf[1], to_hf[2], to_hf[3] = 1, 3, 0, 2       # SONAR pad/unk/bos/eos -> HF ids
    from_hf = torch.empty_like(to_hf)
    from_hf[to_hf] = torch.arange(V)

    emb = torch.randn(N, D, generator=torch.Generator().manual_seed(11)) * 0.5
    runs = []
    for lang, min_new in ((57, 1), (101, 1), (57, 4)):
        prompt = torch.tensor([3, lang])
        with torch.no_grad():
            out = m.generate(encoder_outputs=BaseModelOutput(last_hidden_state=emb.unsqueeze(1)),
                             decoder_input_ids=to_hf[prompt].unsqueeze(0).expand(N, -1).contiguous(),
                             num_beams=1, do_sample=False, max_new_tokens=MAX_NEW, min_new_tokens=min_new)
        toks = from_hf[out]                                    # SONAR ids, pad (0) after </s>
        gen = []
        for row in toks[:, 2:].tolist():
            gen.append(row[: row.index(3) + 1] if 3 in row else row)
        runs.append({"prompt": prompt.tolist(), "min_gen_len": min_new, "max_new": MAX_NEW, "generated": gen})
        print(f"lang {lang} min_new {min_new}: lengths", [len(g) for g in gen])
    torch.save({"config": dict(model_dim=D, num_heads=H, ffn_inner_dim=F, num_layers=L, vocab_size=V,
                               max_seq_len=MAXPOS - 2),
                "checkpoint": {"state_dict": sd}, "embeddings": emb, "runs": runs}, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
