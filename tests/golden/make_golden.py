"""Generate tests/golden/m2m100_twin.pt -- golden vectors that pin the oracle's
transformer stack against an INDEPENDENT implementation of the same
architecture: HuggingFace `M2M100Encoder`.

Why this twin: the reference itself loads SONAR text-encoder weights into
M2M100Encoder and mean-pools with the attention mask
(examples/finetune_sonar_as_toxicity_classifier.ipynb cells 50-57), and HF's
parameter names are the fairseq checkpoint names that
sonar/models/sonar_text/handler.py:71-82 consumes.  The reference's own runtime
(fairseq2) is not installable offline, so this is the strongest pin available.

Run in the build container:  python tests/golden/make_golden.py
The fixture stores the (fairseq1-layout) checkpoint, the SONAR-id token batch
and HF's outputs, so the tests need neither transformers nor this script.
"""
import os

import torch
from transformers import M2M100Config
from transformers.models.m2m_100.modeling_m2m_100 import M2M100Encoder

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "m2m100_twin.pt")

D, H, F, L, V, MAXPOS = 64, 4, 128, 2, 200, 64


def main():
    torch.manual_seed(20240924)
    cfg = M2M100Config(vocab_size=V, d_model=D, encoder_layers=L, encoder_attention_heads=H,
                       encoder_ffn_dim=F, activation_function="relu", scale_embedding=True,
                       max_position_embeddings=MAXPOS, dropout=0.0, attention_dropout=0.0,
                       activation_dropout=0.0, encoder_layerdrop=0.0, pad_token_id=1,
                       bos_token_id=0, eos_token_id=2)
    enc = M2M100Encoder(cfg).eval().float()
    with torch.no_grad():
        for name, p in enc.named_parameters():  # make every parameter non-trivial
            if "layer_norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(0.15 * torch.randn_like(p))
    # fairseq1-layout checkpoint, as the reference's handler expects it
    sd = {k: v.clone() for k, v in enc.state_dict().items() if not k.startswith("embed_positions")}
    sd["version"] = torch.tensor([3.0])
    sd["embed_positions._float_tensor"] = torch.zeros(1)
    ckpt = {"state_dict": sd}

    # SONAR-id batch (pad 0, unk 1, bos 2, eos 3), ragged, right-padded with 0
    g = torch.Generator().manual_seed(5)
    lens = torch.tensor([1, 2, 7, 19, 33, 40], dtype=torch.int64)
    S = int(lens.max())
    ids = torch.zeros(len(lens), S, dtype=torch.int64)
    for i, n in enumerate(lens.tolist()):
        row = torch.randint(4, V, (n,), generator=g)
        row[-1] = 3                      # </s>
        if n > 2:
            row[1] = 1                   # an <unk>
            row[2] = 2                   # a <s>
        ids[i, :n] = row
    # SONAR ids -> fairseq/HF ids for the four control tokens (handler.py:86-92 inverse)
    to_hf = torch.arange(V)
    to_hf[0], to_hf[1], to_hf[2], to_hf[3] = 1, 3, 0, 2
    mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1))
    hf_ids = torch.where(mask, to_hf[ids], torch.full_like(ids, 1))
    with torch.no_grad():
        hid = enc(input_ids=hf_ids, attention_mask=mask.long()).last_hidden_state
    hid = hid * mask.unsqueeze(-1)       # HF leaves garbage at padded positions
    pooled = hid.sum(1) / lens.unsqueeze(1).float()

    # a full (non-ragged) batch as well: the reference passes no mask then
    ids_full = torch.randint(4, V, (3, 16), generator=g)
    with torch.no_grad():
        hid_full = enc(input_ids=to_hf[ids_full], attention_mask=torch.ones(3, 16, dtype=torch.long)).last_hidden_state

    torch.save({
        "config": dict(model_dim=D, num_heads=H, ffn_inner_dim=F, num_layers=L, vocab_size=V,
                       max_seq_len=MAXPOS - 2),
        "checkpoint": ckpt, "ids": ids, "lens": lens, "hidden": hid, "pooled": pooled,
        "ids_full": ids_full, "hidden_full": hid_full, "pooled_full": hid_full.mean(1),
        "pos_rows_2_5": enc.embed_positions.weights[2:5].clone(),
    }, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
