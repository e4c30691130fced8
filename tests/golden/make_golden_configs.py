"""Generate tests/golden/configs_reference.json by RUNNING THE REFERENCE'S OWN config code in this container:
`register_sonar_text_encoder_configs`, `register_sonar_text_decoder_configs` (/root/reference/sonar/models/sonar_text/config.py:
87-127, 192-255) and `register_sonar_speech_encoder_configs` (sonar_speech/config.py:54-100), imported by path with a recording
stand-in for fairseq2's RuntimeContext / config registry and a plain dataclass for `VocabularyInfo`; the speech configs' nested
w2v-BERT "600m" encoder config comes from fairseq2 itself (absent here) and is recorded as a sentinel -- only the SONAR-level
fields are pinned.  The fixture holds every field of every registered arch.
Run in the build container:  python tests/golden/make_golden_configs.py
"""
import dataclasses
import enum
import importlib.util
import json
import os
import sys
import types

sys.dont_write_bytecode = True   # importing the reference by path must not leave __pycache__ in /root/reference

REF = "/root/reference/sonar/models"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs_reference.json")


@dataclasses.dataclass
class VocabularyInfo:
    size: int
    unk_idx: int
    bos_idx: int
    eos_idx: int
    pad_idx: int


class TransformerNormOrder(enum.Enum):
    POST = 0
    PRE = 1
    PRE_WITH_NORMFORMER = 2


class Registry:
    def __init__(self):
        self.archs = {}

    def decorator(self, name):
        def wrap(fn):
            self.archs[name] = fn
            return fn
        return wrap

    def get(self, name):   # the w2v-BERT registry: not available offline
        return types.SimpleNamespace(w2v2_config=types.SimpleNamespace(encoder_config=f"<fairseq2 w2vbert {name} encoder config>"))


class Context:
    def __init__(self):
        self.registries = {}

    def get_config_registry(self, kls):
        return self.registries.setdefault(kls.__name__, Registry())


def _stub(name, **attrs):
    mod = sys.modules.get(name) or types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def plain(v):
    if dataclasses.is_dataclass(v):
        return {f.name: plain(getattr(v, f.name)) for f in dataclasses.fields(v)}
    if isinstance(v, enum.Enum):
        return v.name
    return v


def main():
    dummy = lambda n: type(n, (), {})
    _stub("fairseq2")
    _stub("fairseq2.context", RuntimeContext=Context)
    _stub("fairseq2.data", VocabularyInfo=VocabularyInfo)
    _stub("fairseq2.models")
    _stub("fairseq2.models.w2vbert", W2VBertConfig=dummy("W2VBertConfig"))
    _stub("fairseq2.models.wav2vec2", Wav2Vec2EncoderConfig=dummy("Wav2Vec2EncoderConfig"))
    _stub("fairseq2.nn")
    _stub("fairseq2.nn.transformer", TransformerNormOrder=TransformerNormOrder)
    text = load("ref_sonar_text_config", f"{REF}/sonar_text/config.py")
    speech = load("ref_sonar_speech_config", f"{REF}/sonar_speech/config.py")
    ctx = Context()
    text.register_sonar_text_encoder_configs(ctx)
    text.register_sonar_text_decoder_configs(ctx)
    speech.register_sonar_speech_encoder_configs(ctx)
    out = {}
    for kls, reg in ctx.registries.items():
        if not reg.archs:
            continue
        out[kls] = {name: plain(fn()) for name, fn in reg.archs.items()}
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, sorted(v))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
