"""GPU parity of the SpeechToEmbedding hot path (filterbank, conformer encoder, attention pooler,
through the C ABI) against the CPU oracle."""
import wave

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cfgs(layers=2, pool=2):
    from oracle.speech_encoder import OracleSpeechEncoderConfig
    from sonar_amd.speech_encoder import SonarSpeechEncoderConfig

    o = OracleSpeechEncoderConfig(model_dim=256, num_layers=layers, num_heads=4, ffn_inner_dim=512, conv_kernel=7,
                                  pooler_layers=pool, pooler_heads=4, pooler_ffn_dim=384, pooler_vocab=64)
    c = SonarSpeechEncoderConfig(model_dim=256, num_encoder_layers=layers, num_encoder_attn_heads=4, ffn_inner_dim=512,
                                 depthwise_conv_kernel_size=7, num_decoder_layers=pool, num_decoder_attn_heads=4,
                                 decoder_ffn_inner_dim=384, max_frames=512)
    return o, c


def _cos_err(a, b):
    return (1 - F.cosine_similarity(a.float().cpu(), b.float().cpu(), dim=-1)).abs().max().item()


@pytest.mark.parametrize("nsamples", [400, 16000, 80640, 33333])
@pytest.mark.parametrize("standardize", [False, True])
def test_fbank_vs_oracle(nsamples, standardize):
    from oracle import speech_encoder as OS
    from sonar_amd.speech_encoder import waveform_to_fbank

    if nsamples == 400 and standardize:
        pytest.skip("a single frame has no std")
    g = torch.Generator().manual_seed(nsamples)
    wav = torch.rand(nsamples, generator=g) * 2 - 1
    wav = wav * 0.3 + 0.2 * torch.sin(torch.arange(nsamples) * 0.05)
    ref = OS.kaldi_fbank(wav, standardize=standardize)
    got = waveform_to_fbank(wav.cuda(), standardize=standardize).cpu()
    assert got.shape == ref.shape == (1 + (nsamples - 400) // 160, 80)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= (3e-3 if standardize else 2e-3)


@pytest.mark.parametrize("ragged", [True, False])
@pytest.mark.parametrize("fp16_residual", [True, False])
def test_speech_encoder_vs_oracle(ragged, fp16_residual):
    from oracle import speech_encoder as OS
    from sonar_amd.speech_encoder import SpeechEncoderEngine

    ocfg, cfg = _cfgs()
    params = OS.make_synthetic_params(ocfg, seed=99, std=0.06)
    g = torch.Generator().manual_seed(7)
    n, t = 5, 300
    fb = torch.randn(n, t, 80, generator=g)
    lens = torch.tensor([300, 97, 2, 158, 299]) if ragged else None
    if ragged:
        for i, L in enumerate(lens.tolist()):
            fb[i, L:] = 0
    _, ref = OS.speech_encoder_forward(params, ocfg, fb, lens)
    eng = SpeechEncoderEngine(cfg, params, device="cuda:0", fp16_residual=fp16_residual)
    emb = eng.forward(fb.cuda(), lens, torch.float32)
    torch.cuda.synchronize()
    assert emb.shape == (n, 256) and torch.isfinite(emb).all()
    assert _cos_err(emb, ref) <= 1e-3, _cos_err(emb, ref)
    assert (emb.cpu() - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
    emb16 = eng.forward(fb.cuda(), lens, torch.float16)
    assert emb16.dtype == torch.float16 and _cos_err(emb16, ref) <= 1e-3
    # batching invariance: each clip alone gives the same vector (reference: speech pipeline tests)
    i = 1 if ragged else 0
    L = int(lens[i]) if ragged else t
    L -= L % 2
    one = eng.forward(fb[i:i + 1, :L].cuda(), None, torch.float32)
    assert _cos_err(one, emb[i:i + 1]) <= 1e-5


def test_bf16_speech_model_vs_oracle():
    """`dtype=torch.bfloat16` (the reference's pipelines take any dtype, sonar/inference_pipelines/speech.py:402-474 ->
    `model.to(device, dtype)`): bf16 weights are exact fp16 operands, the residual stream is fp32 (a bf16 model's
    activations have fp32 range), embeddings come back in bf16 -- within the north_star bound of the fp32 oracle run on the
    SAME (bf16-representable) weights, and equal to the fp16 model's up to the stream's and the output's roundings."""
    from oracle import speech_encoder as OS
    from sonar_amd.speech_encoder import SonarSpeechEncoderModel
    from sonar_amd.text_encoder import PaddingMask, SequenceBatch

    ocfg, cfg = _cfgs()
    params = {k: v.to(torch.bfloat16) for k, v in OS.make_synthetic_params(ocfg, seed=31, std=0.06).items()}
    g = torch.Generator().manual_seed(9)
    lens = torch.tensor([300, 97, 158, 299])
    fb = torch.randn(len(lens), 300, 80, generator=g)
    for i, L in enumerate(lens.tolist()):
        fb[i, L:] = 0
    _, ref = OS.speech_encoder_forward({k: v.float() for k, v in params.items()}, ocfg, fb, lens)
    batch = SequenceBatch(fb.cuda(), PaddingMask(lens, 300))
    out = SonarSpeechEncoderModel(cfg, params, device="cuda:0", dtype=torch.bfloat16)(batch).sentence_embeddings
    assert out.dtype == torch.bfloat16 and out.shape == (4, 256) and torch.isfinite(out.float()).all()
    err = _cos_err(out, ref)
    print(f"bf16 speech model: max (1 - cos) vs oracle = {err:.2e}")
    assert err <= 1e-3, err
    out16 = SonarSpeechEncoderModel(cfg, params, device="cuda:0", dtype=torch.float16)(batch).sentence_embeddings
    assert out16.dtype == torch.float16 and _cos_err(out16, out) <= 1e-4


def test_speech_encoder_10s_frames_ragged_vs_oracle():
    """The shape BASELINE configs[3] is timed on (998 filterbank frames = 499 conformer frames per clip: eight 64-key
    tiles and relative positions out to +-498 in `relpos_attention_kernel`), as a RAGGED batch at small width."""
    from oracle import speech_encoder as OS
    from sonar_amd.speech_encoder import SpeechEncoderEngine

    ocfg, cfg = _cfgs()
    params = OS.make_synthetic_params(ocfg, seed=123, std=0.06)
    g = torch.Generator().manual_seed(17)
    lens = torch.tensor([998, 640, 130, 996, 386])
    fb = torch.randn(len(lens), 998, 80, generator=g)
    for i, L in enumerate(lens.tolist()):
        fb[i, L:] = 0
    _, ref = OS.speech_encoder_forward(params, ocfg, fb, lens)
    for fp16_residual in (True, False):
        eng = SpeechEncoderEngine(cfg, params, device="cuda:0", fp16_residual=fp16_residual)
        emb = eng.forward(fb.cuda(), lens, torch.float32)
        assert torch.isfinite(emb).all()
        err = _cos_err(emb, ref)
        print(f"5 clips of 499 / 320 / 65 / 498 / 193 frames, fp16_residual={fp16_residual}: max (1 - cos) {err:.2e}")
        assert err <= 1e-3, err
        assert (emb.cpu() - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
        one = eng.forward(fb[:1].cuda(), None, torch.float32)     # the full-length clip alone
        assert _cos_err(one, emb[:1]) <= 1e-5
        del eng


def test_speech_pipeline_end_to_end(tmp_path):
    from oracle import speech_encoder as OS
    from sonar_amd.inference_pipelines import SpeechToEmbeddingModelPipeline
    from sonar_amd.speech_encoder import SonarSpeechEncoderModel

    ocfg, cfg = _cfgs(layers=1, pool=1)
    params = OS.make_synthetic_params(ocfg, seed=5, std=0.06)
    model = SonarSpeechEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32)
    pipe = SpeechToEmbeddingModelPipeline(model, device=torch.device("cuda:0"))
    g = torch.Generator().manual_seed(3)
    wavs = [torch.rand(1, 24000, generator=g) * 2 - 1, torch.rand(1, 17777, generator=g) * 2 - 1,
            torch.rand(1, 24000, generator=g) * 2 - 1]
    # the third clip also goes through a 16-bit WAV file (tensor-vs-file equality is a reference test:
    # tests/integration_tests/test_sonar_speech_pipeline_models.py:28-40)
    pcm = (wavs[2][0] * 32767).round().clamp(-32768, 32767).to(torch.int16)
    path = tmp_path / "clip.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.numpy().tobytes())
    wavs[2] = (pcm.float() / 32768.0).unsqueeze(0)
    out = pipe.predict([wavs[0], wavs[1], str(path)], batch_size=2)
    out_t = pipe.predict(wavs, batch_size=3)
    assert out.shape == (3, 256)
    assert _cos_err(out, out_t) <= 1e-5
    # oracle on the oracle's own filterbank
    feats = [OS.kaldi_fbank(w[0]) for w in wavs]
    for i, f in enumerate(feats):
        t = f.shape[0] + f.shape[0] % 2
        fb = torch.zeros(1, t, 80)
        fb[0, : f.shape[0]] = f
        _, ref = OS.speech_encoder_forward(params, ocfg, fb, torch.tensor([f.shape[0]]))
        assert _cos_err(out[i:i + 1], ref) <= 1e-3


def test_reference_unit_tests_zeros_waveform_and_file(tmp_path):
    """The reference's own pipeline unit tests (tests/unit_tests/test_sonar_speech.py:29-47) on a small model: an all-zero clip,
    a random clip and a WAV file of 175 920 samples each give one embedding row.  The reference asserts the shape only; here the
    random clip and the file are also compared with the oracle, and the all-zero clip -- whose standardised filterbank is 0 / 0
    (std_mean over constant frames, no epsilon: fairseq2's converter and the oracle alike) -- must come out non-finite from the
    engine exactly as it does from the oracle, not as a plausible-looking vector."""
    from oracle import speech_encoder as OS
    from sonar_amd.inference_pipelines import SpeechToEmbeddingModelPipeline
    from sonar_amd.speech_encoder import SonarSpeechEncoderModel

    import dataclasses

    ocfg, cfg = _cfgs(layers=1, pool=1)
    cfg = dataclasses.replace(cfg, max_frames=1024)   # 175 920 samples = 1 098 frames = 549 stacked frames
    params = OS.make_synthetic_params(ocfg, seed=8, std=0.06)
    model = SonarSpeechEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32)
    pipe = SpeechToEmbeddingModelPipeline(model, device=torch.device("cuda:0"))
    n = 175920

    def oracle(w):
        f = OS.kaldi_fbank(w)
        t = f.shape[0] + f.shape[0] % 2
        fb = torch.zeros(1, t, 80)
        fb[0, : f.shape[0]] = f
        return OS.speech_encoder_forward(params, ocfg, fb, torch.tensor([f.shape[0]]))[1]

    zeros = pipe.predict([torch.zeros(1, n)])
    assert zeros.shape == (1, 256)
    assert not torch.isfinite(oracle(torch.zeros(n))).any()
    assert not torch.isfinite(zeros).any()
    g = torch.Generator().manual_seed(11)
    fake = torch.rand(1, n, generator=g)
    emb = pipe.predict([fake])
    assert emb.shape == (1, 256) and torch.isfinite(emb).all()
    assert _cos_err(emb, oracle(fake[0])) <= 1e-3
    pcm = (fake[0] * 32767).round().clamp(-32768, 32767).to(torch.int16)
    path = tmp_path / "audio.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.numpy().tobytes())
    emb_f = pipe.predict([str(path.resolve())])
    assert emb_f.shape == (1, 256)
    assert _cos_err(emb_f, oracle(pcm.float() / 32768.0)) <= 1e-3


def test_fbank_of_digital_silence_is_not_a_number_as_in_the_oracle():
    """A constant clip has constant log-mel columns: mean == the value, deviations 0, unbiased std 0, 0 / 0.  torch.std_mean (the
    oracle; at::std_mean in the reference's converter [fs2-recall]) returns NaN features; the kernels sum around the column's
    first value so that they do too, on the single-clip path (one and several workgroups per pass) and inside a batch, where the
    neighbouring clip must stay untouched."""
    from oracle import speech_encoder as OS
    from sonar_amd.speech_encoder import waveform_to_fbank, waveforms_to_fbank_batch

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    noise = (torch.rand(40000, generator=g) * 2 - 1)
    for n in (4000, 33333, 175920):
        want = OS.kaldi_fbank(torch.zeros(n))
        got = waveform_to_fbank(torch.zeros(n, device=dev))
        assert got.shape == want.shape and torch.isnan(want).all() and torch.isnan(got).all(), n
        assert torch.isnan(waveform_to_fbank(torch.full((n,), 0.25, device=dev))).all()   # any constant: DC is removed per frame
    fb, lens = waveforms_to_fbank_batch([torch.zeros(33333, device=dev), noise.to(dev)])
    assert torch.isnan(fb[0, : lens[0]]).all() and (fb[0, lens[0]:] == 0).all()
    single = waveform_to_fbank(noise.to(dev))
    assert torch.isfinite(fb[1, : lens[1]]).all()
    assert (fb[1, : lens[1]] - single).abs().max().item() <= 1e-4
    assert (single.cpu() - OS.kaldi_fbank(noise)).abs().max().item() <= 2e-3


def test_batched_fbank_equals_per_clip_and_pads_with_zeros():
    from sonar_amd.speech_encoder import waveform_to_fbank, waveforms_to_fbank_batch

    g = torch.Generator().manual_seed(11)
    clips = [torch.rand(n, generator=g) * 2 - 1 for n in (16000, 399, 24123, 400, 8000)]
    fb, lens = waveforms_to_fbank_batch([c.cuda() for c in clips])
    assert lens == [98, 0, 149, 1, 48] and fb.shape == (5, 150, 80)
    for i, c in enumerate(clips):
        if lens[i] >= 2:
            one = waveform_to_fbank(c.cuda())
            assert (fb[i, : lens[i]] - one).abs().max().item() <= 2e-4
        assert (fb[i, lens[i]:] == 0).all()
    raw, _ = waveforms_to_fbank_batch([c.cuda() for c in clips], standardize=False)
    assert torch.equal(raw[3, :1], waveform_to_fbank(clips[3].cuda(), standardize=False))   # a single frame
    assert torch.equal(raw[0, :98], waveform_to_fbank(clips[0].cuda(), standardize=False))


def test_tsv_driven_pipelines_on_the_reference_clips(tmp_path):
    """The TSV-driven pipelines (sonar/inference_pipelines/speech.py:42-274; the reference's golden test drives the
    encoder through them, tests/integration_tests/test_sonar_speech_encoder.py:27-78) on the reference's own TSV and
    clips with a small random-init encoder: `build_pipeline(params)` must yield, per bucket, the embeddings
    SpeechToEmbeddingModelPipeline.predict returns for the same files, in the reference's element structure."""
    from pathlib import Path

    from oracle import speech_encoder as OS
    from sonar_amd.inference_pipelines import (SpeechInferenceParams, SpeechToEmbeddingModelPipeline,
                                               SpeechToEmbeddingPipeline)
    from sonar_amd.speech_encoder import SonarSpeechEncoderModel

    data = Path(__file__).parent / "golden" / "reference_data"
    ocfg, cfg = _cfgs(layers=1, pool=1)
    params = OS.make_synthetic_params(ocfg, seed=5, std=0.06)
    model = SonarSpeechEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32)
    dev = torch.device("cuda:0")
    ctx = SpeechInferenceParams(data_file=data / "audio_ref.tsv", audio_root_dir=data, audio_path_index=1,
                                target_lang="fra_Latn", batch_size=4, pad_idx=0, device=dev,
                                fbank_dtype=torch.float32, n_parallel=1)
    dp = SpeechToEmbeddingPipeline(model).build_pipeline(ctx)
    items = list(dp)
    assert len(items) == 1 and [Path(p).name for p in items[0]["audio"]["path"]] == ["audio_1.wav", "audio_2.wav"]
    got = items[0]["audio"]["data"].sentence_embeddings
    want = SpeechToEmbeddingModelPipeline(model, device=dev).predict(
        [str(data / "audio_1.wav"), str(data / "audio_2.wav")], batch_size=4)
    assert got.shape == (2, 256) and _cos_err(got, want) <= 1e-6
    # batch_size 1: two buckets, same rows; the pipeline object can be iterated again
    ctx1 = SpeechInferenceParams(data_file=data / "audio_ref.tsv", audio_root_dir=data, audio_path_index=1,
                                 batch_size=1, device=dev, n_parallel=2, n_prefetched_batches=1)
    dp1 = SpeechToEmbeddingPipeline(model).build_pipeline(ctx1)
    for _ in range(2):
        rows = torch.cat([it["audio"]["data"].sentence_embeddings for it in dp1])
        assert _cos_err(rows, want) <= 1e-5
