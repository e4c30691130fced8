"""Generate tests/golden/wiring_reference.pt by RUNNING THE REFERENCE'S OWN MODEL CLASSES in this container:

  * `SonarSpeechEncoderModel.forward` (/root/reference/sonar/models/sonar_speech/model.py:59-77): frontend -> encoder -> the moved
    LayerNorm -> dropout -> pooler, and what the three output fields are;
  * `AttentionEncoderOutputPooler.__call__` (/root/reference/sonar/nn/encoder_pooler.py:70-89): one BOS token per clip through the
    decoder frontend WITHOUT a padding mask, the decoder attending over the encoder output WITH the encoder's mask,
    `projection_out(...).squeeze(1)`;
  * `ConditionalTransformerDecoderModel.decode / project` (/root/reference/sonar/nn/conditional_decoder_model.py:66-94) under
    `SonarEncoderDecoderModel.encode / decode / project` (/root/reference/sonar/models/sonar_translation/model.py:48-78) with the
    reference's `DummyEncoderModel` in front: how a sentence vector becomes teacher-forced logits.

imported by path.  fairseq2 is not installable here, so the LAYER STACKS those classes are handed (frontend, conformer encoder,
pooler decoder, text decoder, projections) are torch modules assembled from the oracle's block functions
(oracle/speech_encoder.py, oracle/text_decoder.py -- whose block arithmetic is pinned elsewhere: HF twins);
what this fixture pins is everything the reference's classes do AROUND them -- order, which mask goes where, the BOS token, the
squeeze / unsqueeze, the output fields.  The expected outputs are what the reference's classes returned; the test
(tests/test_oracle_cpu.py::test_wiring_matches_reference_classes) requires the oracle's top-level functions to equal them.
Run in the build container:   python tests/golden/make_golden_wiring.py
"""
import importlib.util
import math
import os
import sys
import types

sys.dont_write_bytecode = True   # importing the reference by path must not leave __pycache__ in /root/reference

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/sonar"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wiring_reference.pt")


class PaddingMask:
    def __init__(self, seq_lens, batch_seq_len):
        self.seq_lens, self.batch_seq_len = seq_lens, batch_seq_len

    def materialize(self):
        return torch.arange(self.batch_seq_len).unsqueeze(0) < self.seq_lens.unsqueeze(1)


class SequenceBatch:
    def __init__(self, seqs, padding_mask):
        self.seqs, self.padding_mask = seqs, padding_mask


class SequenceModelOutput:
    def __init__(self, logits, pad_idx=None):
        self.logits, self.pad_idx = logits, pad_idx


class VocabularyInfo:
    def __init__(self, size, unk_idx, bos_idx, eos_idx, pad_idx):
        self.size, self.unk_idx, self.bos_idx, self.eos_idx, self.pad_idx = size, unk_idx, bos_idx, eos_idx, pad_idx


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
            self.model_dim, self.max_target_seq_len, self.target_vocab_info = model_dim, max_target_seq_len, target_vocab_info

    _stub("fairseq2")
    _stub("fairseq2.data", VocabularyInfo=VocabularyInfo)
    _stub("fairseq2.models")
    _stub("fairseq2.models.sequence", SequenceBatch=SequenceBatch, SequenceModelOutput=SequenceModelOutput)
    _stub("fairseq2.models.transformer", TransformerFrontend=dummy("TransformerFrontend"))
    _stub("fairseq2.models.encoder_decoder", EncoderDecoderModel=EncoderDecoderModel)
    _stub("fairseq2.nn", LayerNorm=torch.nn.LayerNorm, Linear=torch.nn.Linear, IncrementalStateBag=dummy("IncrementalStateBag"),
          Projection=dummy("Projection"))
    _stub("fairseq2.nn.padding", PaddingMask=PaddingMask)
    _stub("fairseq2.nn.transformer", TransformerEncoder=dummy("TransformerEncoder"), TransformerDecoder=dummy("TransformerDecoder"))
    _stub("fairseq2.typing", override=lambda f: f, Device=torch.device)
    for pkg in ("sonar", "sonar.models", "sonar.nn", "sonar.models.sonar_speech", "sonar.models.sonar_translation"):
        _stub(pkg)

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    load("sonar.models.encoder_model", f"{REF}/models/encoder_model.py")
    pool = load("sonar.nn.encoder_pooler", f"{REF}/nn/encoder_pooler.py")
    cond = load("sonar.nn.conditional_decoder_model", f"{REF}/nn/conditional_decoder_model.py")
    speech = load("sonar.models.sonar_speech.model", f"{REF}/models/sonar_speech/model.py")
    trans = load("sonar.models.sonar_translation.model", f"{REF}/models/sonar_translation/model.py")
    return pool, cond, speech, trans


def _key_pad(pm):
    if pm is None:
        return None
    kp = ~pm.materialize()
    return kp if kp.any() else None


def speech_case(pool, speech):
    from oracle import speech_encoder as S
    from oracle.text_encoder import sinusoidal_table

    cfg = S.OracleSpeechEncoderConfig(model_dim=32, ffn_inner_dim=64, num_layers=2, num_heads=4, conv_kernel=5, pooler_layers=2,
                                      pooler_heads=4, pooler_ffn_dim=48, pooler_vocab=11)
    p = S.make_synthetic_params(cfg, seed=31)
    d = cfg.model_dim

    class Frontend(torch.nn.Module):   # w2v-BERT frontend: two stacked fbank frames -> LayerNorm -> Linear (oracle: speech_encoder_forward)
        def forward(self, seqs, padding_mask):
            n, t, nb = seqs.shape
            x = seqs.float().reshape(n, t // 2, 2 * nb)
            x = S._ln(x, p, "encoder_frontend.post_extract_layer_norm", cfg.ln_eps)
            x = F.linear(x, p["encoder_frontend.model_dim_proj.weight"], p["encoder_frontend.model_dim_proj.bias"])
            pm = None if padding_mask is None else PaddingMask(padding_mask.seq_lens // 2, t // 2)
            return x, pm

    class Encoder(torch.nn.Module):
        model_dim = d

        def forward(self, seqs, padding_mask):
            for i in range(cfg.num_layers):
                seqs = S.conformer_block(seqs, p, i, cfg, _key_pad(padding_mask))
            return seqs, padding_mask

    class PoolFrontend(torch.nn.Module):   # TransformerEmbeddingFrontend: embed * sqrt(d) + sinusoidal position 0
        def forward(self, seqs, padding_mask, state_bag=None):
            assert padding_mask is None
            x = p["encoder_pooler.decoder_frontend.embed.weight"][seqs].float() * math.sqrt(d)
            return x + sinusoidal_table(seqs.shape[1], d).unsqueeze(0), None

    class PoolDecoder(torch.nn.Module):
        def forward(self, seqs, padding_mask, encoder_output, encoder_padding_mask, state_bag=None):
            assert padding_mask is None
            for i in range(cfg.pooler_layers):
                seqs = S.pooler_layer(seqs, p, i, cfg, encoder_output, _key_pad(encoder_padding_mask))
            return seqs, None

    proj = torch.nn.Linear(d, d, bias=False)
    ln = torch.nn.LayerNorm(d, eps=cfg.ln_eps)
    with torch.no_grad():
        proj.weight.copy_(p["encoder_pooler.projection_out.weight"])
        ln.weight.copy_(p["layer_norm.weight"])
        ln.bias.copy_(p["layer_norm.bias"])
    pooler = pool.AttentionEncoderOutputPooler(PoolFrontend(), PoolDecoder(), proj, bos_idx=cfg.bos_idx)
    model = speech.SonarSpeechEncoderModel(Frontend(), Encoder(), ln, final_dropout_p=0.1, encoder_pooler=pooler).eval()
    g = torch.Generator().manual_seed(77)
    out = []
    for lens in ([24, 24, 24], [24, 10, 17]):
        fbank = torch.randn(3, 24, cfg.feature_dim // 2, generator=g)
        lens_t = torch.tensor(lens)
        for i, n in enumerate(lens):
            fbank[i, n:] = 0
        masked = lens != [24, 24, 24]
        with torch.no_grad():
            o = model(SequenceBatch(fbank.clone(), PaddingMask(lens_t.clone(), 24) if masked else None))
        out.append({"fbank": fbank, "fbank_lens": lens_t if masked else None, "encoded_seqs": o.encoded_seqs,
                    "sentence_embeddings": o.sentence_embeddings,
                    "padding_mask_seq_lens": None if o.padding_mask is None else o.padding_mask.seq_lens})
    cfg_dict = {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}
    return {"config": cfg_dict, "seed": 31, "cases": out}


def decoder_case(cond, trans):
    from oracle import text_decoder as D
    from oracle.text_encoder import sinusoidal_table

    cfg = D.OracleTextDecoderConfig(model_dim=32, ffn_inner_dim=64, num_layers=2, num_heads=4, vocab_size=50)
    p = D.make_synthetic_params(cfg, seed=5)
    d = cfg.model_dim
    E = p["decoder_frontend.embed.weight"]

    class Frontend(torch.nn.Module):
        def forward(self, seqs, padding_mask, state_bag=None):
            scale = 1.0 if cfg.no_scale_embedding else math.sqrt(d)
            t = seqs.shape[1]
            x = E[seqs].float() * scale + sinusoidal_table(cfg.pos_offset + t, d)[cfg.pos_offset:].unsqueeze(0)
            return x, padding_mask

    class Decoder(torch.nn.Module):
        model_dim = d

        def forward(self, seqs, padding_mask, encoder_output, encoder_padding_mask, state_bag=None):
            assert encoder_padding_mask is None and encoder_output.shape[1] == 1   # what the Dummy encoder hands over
            x = seqs
            for i in range(cfg.num_layers):
                q = f"decoder.layers.{i}."
                h = D._ln(x, p[q + "self_attn_layer_norm.weight"], p[q + "self_attn_layer_norm.bias"], cfg.ln_eps)
                x = x + D._mha(p, q + "self_attn.", h, h, cfg.num_heads, causal=True)
                h = D._ln(x, p[q + "encoder_decoder_attn_layer_norm.weight"], p[q + "encoder_decoder_attn_layer_norm.bias"], cfg.ln_eps)
                x = x + D._mha(p, q + "encoder_decoder_attn.", h, encoder_output, cfg.num_heads, causal=False)
                h = D._ln(x, p[q + "ffn_layer_norm.weight"], p[q + "ffn_layer_norm.bias"], cfg.ln_eps)
                h = F.relu(F.linear(h, p[q + "ffn.inner_proj.weight"], p[q + "ffn.inner_proj.bias"]))
                x = x + F.linear(h, p[q + "ffn.output_proj.weight"], p[q + "ffn.output_proj.bias"])
            return D._ln(x, p["decoder.layer_norm.weight"], p["decoder.layer_norm.bias"], cfg.ln_eps), padding_mask

    class TiedProjection(torch.nn.Module):
        def forward(self, x):
            return F.linear(x, E)

    vocab = VocabularyInfo(cfg.vocab_size, 1, 2, 3, 0)
    decoder = cond.ConditionalTransformerDecoderModel(Frontend(), Decoder(), TiedProjection(), max_target_seq_len=64,
                                                      target_vocab_info=vocab)
    model = trans.SonarEncoderDecoderModel(trans.DummyEncoderModel(d), decoder).eval()
    g = torch.Generator().manual_seed(9)
    emb = torch.randn(3, cfg.cond_dim, generator=g)
    prev = torch.randint(4, cfg.vocab_size, (3, 6), generator=g)
    prev[:, 0] = 3
    with torch.no_grad():
        enc, enc_pm = model.encode(emb.clone(), None)
        dec, dec_pm = model.decode(prev, None, enc, enc_pm)
        logits = model.project(dec, dec_pm)
        # the conditional decoder's own entry points (what fairseq2's generators call on a bare decoder model)
        enc2, enc_pm2 = decoder.encode(enc, enc_pm)
        dec2, dec_pm2 = decoder.decode(prev, None, enc2, enc_pm2)
        assert torch.equal(decoder.project(dec2, dec_pm2).logits, logits.logits)
    cfg_dict = {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}
    return {"config": cfg_dict, "seed": 5, "embeddings": emb, "prev_tokens": prev, "logits": logits.logits, "pad_idx": logits.pad_idx,
            "encoder_output_shape": list(enc.shape), "encoder_padding_mask_is_none": enc_pm is None}


def main():
    pool, cond, speech, trans = load_reference()
    fix = {"speech": speech_case(pool, speech), "decoder": decoder_case(cond, trans)}
    torch.save(fix, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
