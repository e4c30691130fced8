"""Generate tests/golden/gpu_twin_outputs.pt -- outputs of HuggingFace `M2M100Encoder` / `M2M100Decoder`
/ `generate(num_beams=1)` on the GPU-sized twins of tests/golden/twin_weights.py (d = 256, 4 heads of
64, the smallest model the HIP engines accept).  The GPU tests compare the HIP engines DIRECTLY with
these outputs of an independent implementation, not only with this repository's oracle.

Run in the build container:  python tests/golden/make_golden_gpu_twin.py
"""
import os
import sys

import torch
from transformers import M2M100Config, M2M100ForConditionalGeneration
from transformers.modeling_outputs import BaseModelOutput
from transformers.models.m2m_100.modeling_m2m_100 import M2M100Decoder, M2M100Encoder

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import twin_weights as TW  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_twin_outputs.pt")


def config(tie=True):
    return M2M100Config(vocab_size=TW.V, d_model=TW.D, encoder_layers=TW.L, decoder_layers=TW.L,
                        encoder_attention_heads=TW.H, decoder_attention_heads=TW.H, encoder_ffn_dim=TW.F,
                        decoder_ffn_dim=TW.F, activation_function="relu", scale_embedding=True,
                        max_position_embeddings=TW.MAXPOS, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                        encoder_layerdrop=0.0, decoder_layerdrop=0.0, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                        tie_word_embeddings=tie)


def main():
    out = {}
    # ---- encoder: ragged batch, mean pooling over the valid tokens ----
    enc = M2M100Encoder(config()).eval().float()
    missing = enc.load_state_dict(TW.hf_state_dict("encoder"), strict=False)
    assert not missing.unexpected_keys and all("embed_positions" in k for k in missing.missing_keys), missing
    g = torch.Generator().manual_seed(5)
    lens = torch.tensor([1, 2, 7, 19, 33, 40, 62, 62])
    S = int(lens.max())
    ids = torch.zeros(len(lens), S, dtype=torch.int64)
    for i, n in enumerate(lens.tolist()):
        row = torch.randint(4, TW.V, (n,), generator=g)
        row[-1] = 3
        ids[i, :n] = row
    mask = torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)
    hf_ids = torch.where(mask, TW.TO_HF[ids], torch.full_like(ids, 1))
    with torch.no_grad():
        hid = enc(input_ids=hf_ids, attention_mask=mask.long()).last_hidden_state * mask.unsqueeze(-1)
    out.update(enc_ids=ids, enc_lens=lens, enc_pooled=hid.sum(1) / lens.unsqueeze(1).float())

    # ---- decoder: teacher-forced logits, the sentence vector as a length-1 encoder output ----
    dec = M2M100Decoder(config()).eval().float()
    missing = dec.load_state_dict(TW.hf_state_dict("decoder"), strict=False)
    assert not missing.unexpected_keys and all("embed_positions" in k for k in missing.missing_keys), missing
    n, t = 5, 11
    emb = torch.randn(n, TW.D, generator=g) * 0.3
    prev = torch.randint(4, TW.V, (n, t), generator=g)
    prev[:, 0] = 3
    with torch.no_grad():
        h = dec(input_ids=TW.TO_HF[prev], encoder_hidden_states=emb.unsqueeze(1)).last_hidden_state
        logits = torch.nn.functional.linear(h, dec.embed_tokens.weight)[..., TW.TO_HF]   # SONAR id order
    out.update(dec_emb=emb, dec_prev=prev, dec_logits=logits)

    # ---- greedy generation with the tied model ----
    m = M2M100ForConditionalGeneration(config()).eval().float()
    m.model.decoder.load_state_dict(TW.hf_state_dict("decoder"), strict=False)
    m.lm_head.weight = m.model.decoder.embed_tokens.weight
    from_hf = torch.empty_like(TW.TO_HF)
    from_hf[TW.TO_HF] = torch.arange(TW.V)
    emb2 = torch.randn(12, TW.D, generator=g) * 0.3
    prompt = torch.tensor([3, 57])
    with torch.no_grad():
        gen = m.generate(encoder_outputs=BaseModelOutput(last_hidden_state=emb2.unsqueeze(1)),
                         decoder_input_ids=TW.TO_HF[prompt].unsqueeze(0).expand(12, -1).contiguous(), num_beams=1,
                         do_sample=False, max_new_tokens=10, min_new_tokens=1)
        # HF's margin between the best and the second-best token at every generated position
        full = dec(input_ids=gen[:, :-1], encoder_hidden_states=emb2.unsqueeze(1)).last_hidden_state
        lg = torch.nn.functional.linear(full, dec.embed_tokens.weight)
    top2 = lg.topk(2, dim=-1).values
    toks = from_hf[gen][:, 2:]
    out.update(gen_emb=emb2, gen_prompt=prompt, gen_tokens=toks, gen_margin=(top2[..., 0] - top2[..., 1])[:, 1:])
    torch.save(out, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    print(toks)


if __name__ == "__main__":
    main()
