"""Generate tests/golden/pooling_reference.pt by RUNNING THE REFERENCE'S OWN CODE in this container:

  * `SonarTextTransformerEncoderModel.static_pooling` and `.forward` (/root/reference/sonar/models/sonar_text/model.py:86-143)
  * `DummyEncoderModel.forward` and `SonarEncoderDecoderModel.encode` (/root/reference/sonar/models/sonar_translation/model.py:48-53,
    80-95): how EmbeddingToText hands a sentence vector to the decoder (length-1 source, no padding mask)

imported by path.  Stubbed because fairseq2 is not installable here: the TYPES those functions are annotated with (`SequenceBatch`,
`PaddingMask` = an object with `.seq_lens` and `.materialize()` [fs2-recall: bool [N, S], True on valid positions],
`TransformerFrontend`, `TransformerEncoder`, `LayerNorm`, `EncoderDecoderModel`, `override`), and
`fairseq2.nn.padding.apply_padding_mask(seqs, mask, pad_value)` [fs2-recall: `seqs.where(mask.materialize() broadcast over the
trailing dims, pad_value)`, identity for mask None].  The arithmetic that is pinned -- the LAST gather with `clip_(0)`, the -inf /
0 fills, the `1 / (len + 1e-7)` weights in the OUTPUT dtype, the einsum, the model-level LayerNorm before the pooling, the
`unsqueeze(1)` of the translation model -- is the reference's.  Run in the build container:
    python tests/golden/make_golden_pooling.py
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True   # importing the reference by path must not leave __pycache__ in /root/reference

import torch

REF = "/root/reference/sonar/models"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pooling_reference.pt")


class PaddingMask:
    def __init__(self, seq_lens, batch_seq_len):
        self.seq_lens, self.batch_seq_len = seq_lens, batch_seq_len

    def materialize(self):
        return torch.arange(self.batch_seq_len).unsqueeze(0) < self.seq_lens.unsqueeze(1)


def apply_padding_mask(seqs, padding_mask, pad_value=0):
    if padding_mask is None:
        return seqs
    m = padding_mask.materialize()
    for _ in range(seqs.ndim - m.ndim):
        m = m.unsqueeze(-1)
    return seqs.where(m, pad_value)


class SequenceBatch:
    def __init__(self, seqs, padding_mask):
        self.seqs, self.padding_mask = seqs, padding_mask


def _stub(name, **attrs):
    mod = sys.modules.get(name) or types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def load_reference():
    dummy = lambda n: type(n, (), {})

    class EncoderDecoderModel(torch.nn.Module):
        def __init__(self, model_dim, max_target_seq_len, target_vocab_info):
            super().__init__()
            self.model_dim = model_dim

    _stub("fairseq2")
    _stub("fairseq2.models")
    _stub("fairseq2.models.sequence", SequenceBatch=SequenceBatch, SequenceModelOutput=dummy("SequenceModelOutput"))
    _stub("fairseq2.models.transformer", TransformerFrontend=dummy("TransformerFrontend"))
    _stub("fairseq2.models.encoder_decoder", EncoderDecoderModel=EncoderDecoderModel)
    _stub("fairseq2.nn", LayerNorm=torch.nn.LayerNorm)
    _stub("fairseq2.nn.padding", PaddingMask=PaddingMask, apply_padding_mask=apply_padding_mask)
    _stub("fairseq2.nn.transformer", TransformerEncoder=dummy("TransformerEncoder"))
    _stub("fairseq2.typing", override=lambda f: f)
    for pkg in ("sonar", "sonar.models", "sonar.nn", "sonar.models.sonar_text", "sonar.models.sonar_translation"):
        _stub(pkg)
    _stub("sonar.nn.encoder_pooler", EncoderOutputPooler=dummy("EncoderOutputPooler"))
    _stub("sonar.nn.conditional_decoder_model", ConditionalTransformerDecoderModel=dummy("ConditionalTransformerDecoderModel"))

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    load("sonar.models.encoder_model", f"{REF}/encoder_model.py")          # the reference's own base class / output dataclass
    text = load("sonar.models.sonar_text.model", f"{REF}/sonar_text/model.py")
    trans = load("sonar.models.sonar_translation.model", f"{REF}/sonar_translation/model.py")
    return text, trans


def main():
    text, trans = load_reference()
    M, P = text.SonarTextTransformerEncoderModel, text.Pooling
    g = torch.Generator().manual_seed(20260930)
    cases = []
    for dtype in (torch.float32, torch.float16):
        for shape in ((5, 9, 16), (3, 4, 6, 2)):          # [N, S, M] and a trailing extra dim (the reference's unit test has one)
            seqs = torch.randn(*shape, generator=g).to(dtype)
            n, s = shape[:2]
            lens = torch.randint(1, s + 1, (n,), generator=g)
            lens[0] = s
            for pooling in ("LAST", "MAX", "MEAN"):
                for masked in (True, False):
                    pm = PaddingMask(lens.clone(), s) if masked else None
                    out = M.static_pooling(seqs.clone(), pm, P[pooling])
                    cases.append({"seqs": seqs, "seq_lens": lens if masked else None, "pooling": pooling.lower(), "out": out})
    # LAST with an empty row: (seq_lens - 1).clip_(0) -> position 0
    seqs = torch.randn(3, 5, 4, generator=g)
    lens = torch.tensor([0, 5, 2])
    cases.append({"seqs": seqs, "seq_lens": lens, "pooling": "last",
                  "out": M.static_pooling(seqs.clone(), PaddingMask(lens.clone(), 5), P.LAST)})

    # forward(): frontend -> encoder -> model-level LayerNorm -> pool.  Frontend / encoder are pass-through stand-ins (their
    # arithmetic is pinned elsewhere); what this pins is the ORDER: the LayerNorm is applied to every position before pooling.
    class Frontend(torch.nn.Module):
        model_dim = 16

        def forward(self, seqs, padding_mask):
            return seqs, padding_mask

    class Encoder(torch.nn.Module):
        model_dim = 16

        def forward(self, seqs, padding_mask):
            return seqs * 1.5 + 0.25, padding_mask

    ln = torch.nn.LayerNorm(16)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(16, generator=g) * 0.3 + 1)
        ln.bias.copy_(torch.randn(16, generator=g) * 0.2)
    fwd = []
    x = torch.randn(4, 7, 16, generator=g)
    lens = torch.tensor([7, 3, 1, 5])
    for pooling in ("LAST", "MAX", "MEAN"):
        model = M(Frontend(), Encoder(), layer_norm=ln, pooling=P[pooling])
        with torch.no_grad():
            o = model(SequenceBatch(x.clone(), PaddingMask(lens.clone(), 7)))
        fwd.append({"x": x, "seq_lens": lens, "pooling": pooling.lower(), "ln_weight": ln.weight.detach().clone(),
                    "ln_bias": ln.bias.detach().clone(), "encoded_seqs": o.encoded_seqs, "sentence_embeddings": o.sentence_embeddings})

    # translation glue: DummyEncoderModel is the identity; encode() returns (embeddings.unsqueeze(1), None)
    dummy = trans.DummyEncoderModel(8)
    emb = torch.randn(6, 8, generator=g)
    do = dummy(SequenceBatch(emb, None))
    assert do.sentence_embeddings is emb and do.encoded_seqs is emb and do.padding_mask is None

    class Dec(torch.nn.Module):
        model_dim, max_target_seq_len, target_vocab_info = 8, 64, None

    encdec = trans.SonarEncoderDecoderModel(dummy, Dec())
    enc_out, enc_mask = encdec.encode(emb, None)
    glue = {"embeddings": emb, "encoder_output": enc_out, "encoder_padding_mask_is_none": enc_mask is None}

    torch.save({"static_pooling": cases, "forward": fwd, "translation_glue": glue}, OUT)
    print(len(cases), "static_pooling cases,", len(fwd), "forward cases; wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
