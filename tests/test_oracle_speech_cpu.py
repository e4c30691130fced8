"""CPU: pin the oracle's conformer block / pooler layer against the HuggingFace twins and check
the filterbank restatement's structural properties (its numerics are unpinned, see the oracle header)."""
import math
import os

import torch
from torch.testing import assert_close

from oracle import speech_encoder as OS

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "conformer_twin.pt")


def _cfg(fx):
    d = fx["dims"]
    return OS.OracleSpeechEncoderConfig(model_dim=d["model_dim"], num_layers=1, num_heads=d["num_heads"],
                                        ffn_inner_dim=d["ffn_inner_dim"], conv_kernel=d["conv_kernel"],
                                        pooler_layers=1, pooler_heads=d["num_heads"], pooler_ffn_dim=d["ffn_inner_dim"])


def test_rel_pos_encoding_matches_hf():
    fx = torch.load(GOLDEN, weights_only=False)
    t = fx["block_in"].shape[1]
    assert_close(OS.rel_pos_encoding(t, fx["dims"]["model_dim"]), fx["rel_pos"], atol=1e-6, rtol=0)


def test_conformer_block_matches_hf_twin():
    fx = torch.load(GOLDEN, weights_only=False)
    cfg = _cfg(fx)
    y = OS.conformer_block(fx["block_in"], fx["block_params"], 0, cfg, None)
    assert_close(y, fx["block_out"], atol=3e-5, rtol=1e-4)


def test_pooler_layer_matches_bart_twin():
    fx = torch.load(GOLDEN, weights_only=False)
    cfg = _cfg(fx)
    t = fx["pooler_enc"].shape[1]
    pad = torch.arange(t).unsqueeze(0) >= fx["pooler_lens"].unsqueeze(1)
    y = OS.pooler_layer(fx["pooler_q"], fx["pooler_params"], 0, cfg, fx["pooler_enc"], pad)
    assert_close(y, fx["pooler_out"], atol=3e-5, rtol=1e-4)


def test_fbank_shape_and_invariances():
    g = torch.Generator().manual_seed(4)
    wav = torch.rand(16000, generator=g) * 2 - 1
    fb = OS.kaldi_fbank(wav)
    assert fb.shape == (1 + (16000 - 400) // 160, 80)            # snip_edges frame count
    assert OS.kaldi_fbank(torch.rand(160000, generator=g)).shape[0] == 998   # 10 s -> 998 frames
    # standardised per utterance: zero mean / unit (unbiased) std over time
    assert_close(fb.mean(0), torch.zeros(80), atol=1e-4, rtol=0)
    assert_close(fb.std(0), torch.ones(80), atol=1e-4, rtol=0)
    # gain invariance of the standardised log-mel features, DC invariance (remove_dc_offset)
    assert_close(OS.kaldi_fbank(wav * 0.5), fb, atol=2e-3, rtol=0)
    assert_close(OS.kaldi_fbank(wav * 0.5 + 0.1, standardize=False),
                 OS.kaldi_fbank(wav * 0.5, standardize=False), atol=2e-3, rtol=0)
    # a pure tone lands in the mel bin whose triangle covers it
    tone = torch.sin(2 * math.pi * 1000.0 * torch.arange(16000) / 16000.0)
    raw = OS.kaldi_fbank(tone, standardize=False)
    banks = OS.mel_banks()
    k = round(1000.0 / (16000 / 512))
    assert int(raw.mean(0).argmax()) == int(banks[:, k].argmax())


def test_mel_banks_are_kaldi_shaped():
    banks = OS.mel_banks()
    assert banks.shape == (80, 256)
    assert (banks >= 0).all() and (banks <= 1).all()
    assert (banks.sum(1) > 0).all()
    assert banks[:, 0].sum() == 0                                  # DC bin below low_freq = 20 Hz
    peaks = banks.argmax(1)
    assert (peaks[1:] >= peaks[:-1]).all()


def test_full_model_masks_padding():
    cfg = OS.OracleSpeechEncoderConfig(model_dim=64, num_layers=2, num_heads=4, ffn_inner_dim=128, conv_kernel=7,
                                       pooler_layers=2, pooler_heads=4, pooler_ffn_dim=128, pooler_vocab=64)
    p = OS.make_synthetic_params(cfg, seed=3, std=0.1)
    g = torch.Generator().manual_seed(5)
    fb = torch.randn(2, 40, 80, generator=g)
    lens = torch.tensor([40, 26])
    fb[1, 26:] = 0
    enc, emb = OS.speech_encoder_forward(p, cfg, fb, lens)
    enc1, emb1 = OS.speech_encoder_forward(p, cfg, fb[1:, :26], None)
    assert_close(emb[1], emb1[0], atol=1e-5, rtol=1e-4)            # batching / padding invariance
    assert_close(enc[1, :13], enc1[0], atol=1e-5, rtol=1e-4)
    assert emb.shape == (2, 64)


def test_fbank_matches_hf_seamless_feature_extractor():
    """The Kaldi filterbank restatement against HuggingFace SeamlessM4TFeatureExtractor (an independent
    implementation of the same front end for the same w2v-BERT encoder family) on the committed twin
    (tests/golden/make_golden_fbank.py): log-mel features, then the per-utterance standardisation and
    the 2-frame stacking the encoder frontend consumes."""
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "fbank_seamless_twin.pt"), weights_only=False)
    raw = OS.kaldi_fbank(fx["waveform"], standardize=False)
    assert raw.shape == fx["fbank"].shape == (128, 80)
    # log-mel values span ~[5, 30]; the two implementations differ by fp32 FFT / mel summation order
    assert (raw - fx["fbank"]).abs().max().item() < 2e-3
    assert (raw - fx["fbank"]).abs().mean().item() < 2e-5
    std = OS.kaldi_fbank(fx["waveform"], standardize=True)
    stacked = std[: std.shape[0] // 2 * 2].reshape(-1, 160)          # frames (2t, 2t+1) side by side
    assert stacked.shape == fx["normalized_stacked"].shape
    assert (stacked - fx["normalized_stacked"]).abs().max().item() < 2e-3


def test_speech_path_end_to_end_matches_hf_composition():
    """waveform -> sentence embedding against the composition of independent implementations
    (HF SeamlessM4TFeatureExtractor, torch LayerNorm/Linear, HF Wav2Vec2ConformerEncoderLayer, HF
    BartDecoderLayer) on the GPU-sized twin -- tests/golden/make_golden_speech_e2e.py."""
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "speech_e2e_twin.pt"), weights_only=False)
    cfg = OS.OracleSpeechEncoderConfig(**fx["config"])
    p = OS.make_synthetic_params(cfg, seed=fx["seed"], std=fx["std"])
    for i, wav in enumerate(fx["waveforms"]):
        fb = OS.kaldi_fbank(wav)
        fb = fb[: fb.shape[0] // 2 * 2].unsqueeze(0)
        enc, emb = OS.speech_encoder_forward(p, cfg, fb, None)
        want = fx["embeddings"][i]
        assert (emb[0] - want).abs().max().item() <= 2e-3 * want.abs().max().item()
        assert 1 - torch.nn.functional.cosine_similarity(emb[0], want, dim=0).item() <= 1e-6
        if i == 0:
            assert (enc[0] - fx["encoder_out_first_clip"]).abs().max().item() <= 5e-3
