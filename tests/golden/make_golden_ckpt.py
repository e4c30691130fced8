"""Generate tests/golden/ckpt_reference.pt by RUNNING THE REFERENCE'S OWN CONVERTERS in this container:

  * convert_sonar_text_encoder_checkpoint / convert_sonar_text_decoder_checkpoint from
    /root/reference/sonar/models/sonar_text/handler.py (:52-94, :122-172)
  * convert_sonar_speech_checkpoint from /root/reference/sonar/models/sonar_speech/handler.py (:46-110)

imported by path.  Their bodies are plain dict / regex / torch code; what they import and this image lacks is stubbed:
the handler base class and the config / factory / model classes (only named in class bodies that are never
instantiated here), and fairseq2's `convert_fairseq_checkpoint(checkpoint, key_map)`, which is restated below from
fairseq2 v0.4 (fairseq2/models/utils/checkpoint.py [fs2-recall]: the keys of checkpoint["model"] are renamed by the FIRST
pattern of the map, in dict order, whose `re.sub` changes the key; the result is {"model": renamed}).

Inputs: small synthetic fairseq1-layout checkpoints (tests/ckpt_layouts.py; the text encoder's additionally carries the
top-level `embed_tokens` module the reference reads at handler.py:86, its weight aliasing the state-dict entry as in a
fairseq training checkpoint).  The fixture stores the inputs and, for the reference's outputs, every key with its tensor,
so the test needs neither /root/reference nor this script.  Run in the build container:
    python tests/golden/make_golden_ckpt.py
"""
import importlib.util
import io
import os
import re
import sys
import types

sys.dont_write_bytecode = True   # importing the reference by path must not leave __pycache__ in /root/reference (read-only by contract)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/sonar/models"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ckpt_reference.pt")


def convert_fairseq_checkpoint(checkpoint, key_map):
    """fairseq2 v0.4 semantics [fs2-recall]: first pattern (dict order) whose substitution changes the key wins."""
    def new_key(old):
        for pat, rep in key_map.items():
            k = re.sub(pat, rep, old)
            if k != old:
                return k
        return old

    return {"model": {new_key(k): v for k, v in checkpoint["model"].items()}}


def clone_ckpt(ck):
    """A private copy that keeps storage sharing (embed_tokens module <-> state dict entry; the decoder's tied
    output_projection): what torch.load gives for a checkpoint file.  copy.deepcopy does not keep Parameter / .data sharing."""
    buf = io.BytesIO()
    torch.save(ck, buf)
    buf.seek(0)
    return torch.load(buf, weights_only=False)


def _stub(name, **attrs):
    mod = sys.modules.get(name) or types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def load_reference_handlers():
    dummy = lambda n: type(n, (), {})
    _stub("fairseq2")
    _stub("fairseq2.models", AbstractModelHandler=dummy("AbstractModelHandler"))
    _stub("fairseq2.models.utils")
    _stub("fairseq2.models.utils.checkpoint", convert_fairseq_checkpoint=convert_fairseq_checkpoint)
    for pkg in ("sonar", "sonar.models", "sonar.models.sonar_text", "sonar.models.sonar_speech", "sonar.nn"):
        _stub(pkg)
    _stub("sonar.models.sonar_text.config", SonarTextDecoderConfig=dummy("SonarTextDecoderConfig"),
          SonarTextEncoderConfig=dummy("SonarTextEncoderConfig"))
    _stub("sonar.models.sonar_text.factory", SonarTextDecoderFactory=dummy("F1"), SonarTextEncoderFactory=dummy("F2"))
    _stub("sonar.models.sonar_text.model", SonarTextTransformerEncoderModel=dummy("M1"))
    _stub("sonar.nn.conditional_decoder_model", ConditionalTransformerDecoderModel=dummy("M2"))
    _stub("sonar.models.sonar_speech.config", SonarSpeechEncoderConfig=dummy("SonarSpeechEncoderConfig"))
    _stub("sonar.models.sonar_speech.factory", SonarSpeechEncoderFactory=dummy("F3"))
    _stub("sonar.models.sonar_speech.model", SonarSpeechEncoderModel=dummy("M3"))

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    return (load("ref_sonar_text_handler", f"{REF}/sonar_text/handler.py"),
            load("ref_sonar_speech_handler", f"{REF}/sonar_speech/handler.py"))


def make_inputs():
    from oracle import speech_encoder as OS
    from oracle import text_decoder as OD
    from oracle import text_encoder as OE
    from tests.ckpt_layouts import speech_encoder_to_fairseq1, text_decoder_to_fairseq1, text_encoder_to_fairseq1

    enc = text_encoder_to_fairseq1(OE.make_synthetic_params(
        OE.OracleTextEncoderConfig(model_dim=32, num_layers=2, num_heads=1, ffn_inner_dim=64, vocab_size=24), seed=41))
    # handler.py:86 reads checkpoint["embed_tokens"].weight: the embedding MODULE a fairseq training checkpoint carries,
    # whose weight is the state dict's tensor
    emb = torch.nn.Embedding(24, 32)
    emb.weight = torch.nn.Parameter(enc["state_dict"]["embed_tokens.weight"], requires_grad=False)
    enc["state_dict"]["embed_tokens.weight"] = emb.weight.data
    enc["embed_tokens"] = emb
    dec = text_decoder_to_fairseq1(OD.make_synthetic_params(
        OD.OracleTextDecoderConfig(model_dim=32, num_layers=2, num_heads=1, ffn_inner_dim=64, vocab_size=24, max_seq_len=16),
        seed=42, std=0.1))
    sp = speech_encoder_to_fairseq1(OS.make_synthetic_params(
        OS.OracleSpeechEncoderConfig(model_dim=32, num_layers=2, num_heads=1, ffn_inner_dim=64, conv_kernel=7,
                                     pooler_layers=2, pooler_heads=1, pooler_ffn_dim=48, pooler_vocab=16), seed=43, std=0.06))
    return enc, dec, sp


def main():
    text, speech = load_reference_handlers()
    enc, dec, sp = make_inputs()
    fixture = {"inputs": {"text_encoder": clone_ckpt(enc), "text_decoder": clone_ckpt(dec),
                          "speech_encoder": clone_ckpt(sp)}}
    # the reference converters modify their argument in place (key deletion, the in-place row permutation): they get
    # their own copies
    r_enc = text.convert_sonar_text_encoder_checkpoint(clone_ckpt(enc))
    r_dec = text.convert_sonar_text_decoder_checkpoint(clone_ckpt(dec))
    cfg = types.SimpleNamespace(w2v2_encoder_config=types.SimpleNamespace(use_conformer=True))
    r_sp = speech.convert_sonar_speech_checkpoint(clone_ckpt(sp), cfg)
    # the encoder converter permutes checkpoint["embed_tokens"].weight IN PLACE (handler.py:86-91) and leaves it at the TOP
    # level of its result (:92) next to "model"; the table inside "model" is permuted because it shares that storage
    fixture["reference"] = {
        "text_encoder": {"model": dict(r_enc["model"]),
                         "top_level_embed": r_enc["encoder_frontend.embed.weight"].clone()},
        "text_decoder": {"model": dict(r_dec["model"])},
        "speech_encoder": {"model": dict(r_sp["model"])},
    }
    # fairseq2-layout pass-through (handler.py:54-58, :124-128, sonar_speech/handler.py:52-53): the converters return
    # their argument itself
    for name, fn, key in (("text_encoder", text.convert_sonar_text_encoder_checkpoint, "encoder_frontend.embed.weight"),
                          ("text_decoder", text.convert_sonar_text_decoder_checkpoint, "decoder_frontend.embed.weight")):
        ck = {"model": {key: torch.zeros(2, 2)}}
        assert fn(ck) is ck, name
    ck = {"model": {"encoder_frontend.model_dim_proj": torch.zeros(1)}}
    assert speech.convert_sonar_speech_checkpoint(ck, cfg) is ck
    torch.save(fixture, OUT)
    for name, r in fixture["reference"].items():
        print(name, len(r["model"]), "keys")
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
