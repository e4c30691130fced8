"""CPU tests of the host-side pieces added in round 2: WAV decoding (C ABI, host code), card-name
resolution, the generator's length rule, tokenizer <unk> rendering."""
import io
import os
import struct
import wave
from pathlib import Path

import numpy as np
import pytest
import torch

DATA = Path(__file__).parent / "golden" / "reference_data"


def _wav_bytes(samples: np.ndarray, rate=16000, fmt="pcm16", extensible=False) -> bytes:
    """samples: float64 [frames, channels] in [-1, 1)."""
    frames, ch = samples.shape
    if fmt == "pcm8":
        tag, bits, data = 1, 8, (np.clip(np.round(samples * 128) + 128, 0, 255)).astype(np.uint8).tobytes()
    elif fmt == "pcm16":
        tag, bits, data = 1, 16, np.round(samples * 32767).astype("<i2").tobytes()
    elif fmt == "pcm24":
        v = np.round(samples * 8388607).astype("<i4")
        b = v.reshape(-1, 1).view(np.uint8).reshape(-1, 4)[:, :3]
        tag, bits, data = 1, 24, b.tobytes()
    elif fmt == "pcm32":
        tag, bits, data = 1, 32, np.round(samples * 2147483647).astype("<i4").tobytes()
    elif fmt == "f32":
        tag, bits, data = 3, 32, samples.astype("<f4").tobytes()
    elif fmt == "f64":
        tag, bits, data = 3, 64, samples.astype("<f8").tobytes()
    else:
        raise ValueError(fmt)
    align = ch * bits // 8
    if extensible:
        guid = struct.pack("<H", tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt_chunk = struct.pack("<HHIIHHHHI", 0xFFFE, ch, rate, rate * align, align, bits, 22, bits, 0) + guid
    else:
        fmt_chunk = struct.pack("<HHIIHH", tag, ch, rate, rate * align, align, bits)
    junk = b"LIST" + struct.pack("<I", 5) + b"hello" + b"\0"      # an odd-sized chunk the parser must skip
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk + junk + b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


@pytest.mark.parametrize("fmt,tol", [("pcm8", 1 / 100), ("pcm16", 1e-4), ("pcm24", 1e-6), ("pcm32", 1e-7),
                                     ("f32", 1e-7), ("f64", 1e-7)])
@pytest.mark.parametrize("channels", [1, 2])
def test_wav_decode_formats(fmt, tol, channels):
    from sonar_amd.inference_pipelines.speech import decode_wav_bytes

    rng = np.random.default_rng(3)
    x = rng.uniform(-0.9, 0.9, size=(777, channels))
    for ext in (False, True):
        got, rate = decode_wav_bytes(_wav_bytes(x, 16000, fmt, ext))
        assert rate == 16000 and got.shape == (777, channels) and got.dtype == torch.float32
        assert np.abs(got.numpy() - x).max() <= tol + 1e-7


def test_wav_decode_matches_stdlib_on_the_reference_clips(tmp_path):
    from sonar_amd.inference_pipelines.speech import read_wav

    for name, n in (("audio_1.wav", 80640), ("audio_2.wav", 76800)):
        got = read_wav(DATA / name)
        with wave.open(str(DATA / name), "rb") as w:
            raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        assert got.shape == (1, n)
        assert np.array_equal(got[0].numpy(), raw.astype(np.float32) / 32768.0)
    # wrong sample rate and non-WAV input are errors, not silent garbage
    p = tmp_path / "r8k.wav"
    p.write_bytes(_wav_bytes(np.zeros((100, 1)), rate=8000))
    with pytest.raises(ValueError, match="16 kHz"):
        read_wav(p)
    q = tmp_path / "x.flac"
    q.write_bytes(b"fLaC" + bytes(64))            # a FLAC marker without a STREAMINFO block (real streams: test_audio_decode_cpu.py)
    with pytest.raises(ValueError, match="STREAMINFO"):
        read_wav(q)
    q = tmp_path / "x.mp3"
    q.write_bytes(b"\xff\xfb\x90\x00" + bytes(64))
    with pytest.raises(ValueError, match="MPEG audio is not covered"):
        read_wav(q)
    t = tmp_path / "trunc.wav"
    t.write_bytes(_wav_bytes(np.zeros((100, 1)))[:30])
    with pytest.raises(ValueError):
        read_wav(t)


def test_card_resolution(tmp_path, monkeypatch):
    from sonar_amd import cards

    monkeypatch.setenv("SONAR_CHECKPOINT_DIR", str(tmp_path))
    monkeypatch.setenv("HOME", str(tmp_path / "nohome"))
    assert cards.is_card_name("text_sonar_basic_encoder") and cards.is_card_name("sonar_speech_encoder_fra")
    assert not cards.is_card_name("/some/path.pt")
    with pytest.raises(FileNotFoundError, match="sonar_text_encoder.pt"):
        cards.resolve_card("text_sonar_basic_encoder")
    for f in ("sonar_text_encoder.pt", "sonar_text_decoder.pt", "spenc.eng.pt", "spenc.v5ap.hin.pt", "spenc.v3ap.fra.pt",
              cards.NLLB_SPM, "mutox.pt"):
        (tmp_path / f).write_bytes(b"x")
    r = cards.resolve_card("text_sonar_basic_encoder")
    assert r.checkpoint.name == "sonar_text_encoder.pt" and r.arch == "basic" and r.tokenizer.name == cards.NLLB_SPM
    assert cards.resolve_card("text_sonar_basic_decoder").checkpoint.name == "sonar_text_decoder.pt"
    e = cards.resolve_card("sonar_speech_encoder_eng")
    assert (e.checkpoint.name, e.arch) == ("spenc.eng.pt", "english")        # sonar/cards/sonar_speech_encoder.yaml
    h = cards.resolve_card("sonar_speech_encoder_hin")
    assert (h.checkpoint.name, h.arch) == ("spenc.v5ap.hin.pt", "non_english")
    assert cards.resolve_card("sonar_speech_encoder_fra").checkpoint.name == "spenc.v3ap.fra.pt"
    assert cards.resolve_tokenizer("text_sonar_basic_decoder").name == cards.NLLB_SPM
    assert cards.resolve_checkpoint(str(tmp_path / "whatever.pt"), "basic") == (tmp_path / "whatever.pt", "basic")
    assert cards.resolve_card("sonar_mutox").checkpoint.name == "mutox.pt"
    with pytest.raises(KeyError):
        cards.resolve_card("not_a_card")


def test_card_table_matches_the_reference_cards():
    """The name -> file table of sonar_amd/cards.py against the reference's YAML cards (skipped on the GPU box,
    where /root/reference does not exist)."""
    yaml = pytest.importorskip("yaml")
    ref = Path("/root/reference/sonar/cards")
    if not ref.is_dir():
        pytest.skip("reference checkout not present")
    from sonar_amd import cards

    docs = []
    for f in ("text_sonar_basic_encoder.yaml", "text_sonar_basic_decoder.yaml", "text_sonar_finetuned_decoder.yaml",
              "sonar_speech_encoder.yaml", "sonar_mutox.yaml"):
        docs += list(yaml.safe_load_all(open(ref / f)))
    n = 0
    for d in docs:
        if "checkpoint" not in d:
            continue
        base = d["checkpoint"].rsplit("/", 1)[1]
        name = d["name"]
        assert cards.is_card_name(name), name
        if name in cards._TEXT_CARDS:
            assert cards._TEXT_CARDS[name][0] == base
            assert d["tokenizer"].rsplit("/", 1)[1] == cards.NLLB_SPM
        elif name in cards._HEAD_CARDS:
            assert cards._HEAD_CARDS[name][0] == base
        else:
            lang = cards._SPEECH_RE.match(name).group(1)
            assert base in (["spenc.eng.pt"] if lang == "eng" else [f"spenc.v3ap.{lang}.pt", f"spenc.v5ap.{lang}.pt"])
        n += 1
    assert n >= 60


def test_generator_length_rule_of_the_oracle():
    """a * source_len + b with source_len = model_dim for sentence vectors (fairseq2 Seq2SeqGenerator)."""
    from oracle import text_decoder as OD

    cfg = OD.OracleTextDecoderConfig(model_dim=64, num_layers=1, num_heads=1, ffn_inner_dim=128, vocab_size=50, max_seq_len=40)
    params = OD.make_synthetic_params(cfg, seed=1, std=0.2)
    emb = torch.randn(1, 64, generator=torch.Generator().manual_seed(0))
    # default (1, 128) on a sentence vector: the cap is the decoder's max_seq_len, EOS forced at its last slot
    h = OD.beam_search(params, cfg, emb, [3, 7], beam_size=1, min_gen_len=200)
    assert len(h[0][0].seq) == 40 - 2 and h[0][0].seq[-1].item() == 3
    h = OD.beam_search(params, cfg, emb, [3, 7], beam_size=1, min_gen_len=200, max_gen_len=(1, 4), source_len=6)
    assert len(h[0][0].seq) == 10
    s = OD.sampling_generate(params, cfg, emb, [3, 7], ("top_k", 3), seed=1, min_gen_len=200, max_gen_len=(1, 4), source_len=6)
    assert len(s[0][0]) == 10


def test_tokenizer_decode_renders_unk(tmp_path):
    import sentencepiece as spm

    from sonar_amd.tokenizer import NllbTokenizer

    corpus = tmp_path / "c.txt"
    corpus.write_text("\n".join(["hello world my name is paul"] * 50))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "toy"), vocab_size=20, model_type="unigram",
                                   hard_vocab_limit=False, bos_id=1, eos_id=2, unk_id=0, pad_id=-1, minloglevel=2)
    tok = NllbTokenizer(str(tmp_path / "toy.model"))
    ids = tok.create_encoder(lang="eng_Latn")("hello zzz").tolist()
    assert 1 in ids                                             # the unknown characters map to <unk> = 1
    text = tok.decode(ids)
    assert text == tok.sp.decode([i - 1 for i in ids if 4 <= i < tok.lang_base or i == 1])
    assert "⁇" in text and "hello" in text                # rendered, not dropped
    assert tok.decode([tok.lang_idx("eng_Latn"), 3, 0, 2]) == ""  # control symbols never render


def test_arch_configs_equal_the_reference_registrations():
    """Every field of every architecture this package registers (`basic` / `small` text encoder; `basic` / `small` / `toy`
    decoder; `english` / `non_english` speech encoder) against the values the REFERENCE'S OWN registration functions return
    (sonar_text/config.py:87-127, 192-255; sonar_speech/config.py:54-100), executed by path in the build container
    (tests/golden/make_golden_configs.py -> configs_reference.json).  The reference's field set is a subset of ours (we add
    `input_dim` to the decoder, which its factory takes as an argument, and flatten the w2v-BERT encoder sub-config)."""
    import dataclasses
    import json
    import os

    from sonar_amd.speech_encoder import get_speech_encoder_config
    from sonar_amd.text_decoder import get_text_decoder_config
    from sonar_amd.text_encoder import get_text_encoder_config

    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "configs_reference.json")))

    def plain(v):
        return {f.name: plain(getattr(v, f.name)) for f in dataclasses.fields(v)} if dataclasses.is_dataclass(v) else v

    for kls, getter in (("SonarTextEncoderConfig", get_text_encoder_config), ("SonarTextDecoderConfig", get_text_decoder_config)):
        assert sorted(ref[kls]) == sorted({"SonarTextEncoderConfig": ["basic", "small"],
                                           "SonarTextDecoderConfig": ["basic", "small", "toy"]}[kls])
        for arch, want in ref[kls].items():
            got = plain(getter(arch))
            for field, value in want.items():
                assert got[field] == value, (kls, arch, field, got[field], value)
    # speech: the SONAR-level fields (the nested w2v-BERT "600m" encoder config is fairseq2's, recorded as a sentinel)
    names = {"model_dim": "model_dim", "max_seq_len": "max_seq_len", "pad_idx": "pad_idx", "bos_idx": "bos_idx",
             "num_decoder_layers": "num_decoder_layers", "num_decoder_attn_heads": "num_decoder_attn_heads",
             "ffn_inner_dim": "decoder_ffn_inner_dim"}
    assert sorted(ref["SonarSpeechEncoderConfig"]) == ["english", "non_english"]
    for arch, want in ref["SonarSpeechEncoderConfig"].items():
        got = plain(get_speech_encoder_config(arch))
        assert want["decoder_norm_order"] == "POST"          # the pooler's layers are POST-norm: what speech.hip implements
        for rf, ours in names.items():
            assert got[ours] == want[rf], (arch, rf, got[ours], want[rf])
