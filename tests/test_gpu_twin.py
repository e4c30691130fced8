"""GPU: the HIP engines against an INDEPENDENT implementation, not only against this repository's oracle.
HuggingFace M2M100Encoder / M2M100Decoder / generate(num_beams=1) were run in the build container on the
GPU-sized twins of tests/golden/twin_weights.py (tests/golden/make_golden_gpu_twin.py); the weights are
rebuilt here without transformers and go through the reference's fairseq checkpoint converters."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)


@pytest.fixture(scope="module")
def fx():
    return torch.load(os.path.join(GOLDEN, "gpu_twin_outputs.pt"), weights_only=False)


@pytest.mark.parametrize("fp16_residual", [False, True])
def test_text_encoder_matches_hf_m2m100_encoder(fx, fp16_residual):
    import twin_weights as TW
    from sonar_amd.text_encoder import (PaddingMask, SequenceBatch, SonarTextEncoderConfig,
                                        SonarTextTransformerEncoderModel, VocabularyInfo,
                                        convert_sonar_text_encoder_checkpoint)

    cfg = SonarTextEncoderConfig(model_dim=TW.D, num_encoder_layers=TW.L, num_encoder_attn_heads=TW.H,
                                 ffn_inner_dim=TW.F, vocab_info=VocabularyInfo(size=TW.V), pooling="mean",
                                 max_seq_len=TW.MAXPOS - 2, _from_fairseq=True)
    params = convert_sonar_text_encoder_checkpoint(TW.fairseq_checkpoint("encoder"))
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32,
                                             fp16_residual=fp16_residual)
    ids, lens = fx["enc_ids"], fx["enc_lens"]
    emb = model(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))).sentence_embeddings.cpu()
    want = fx["enc_pooled"]
    # north_star tolerance: <= 1e-3 (1 - cosine) against the fp32 reference computation
    assert (1 - F.cosine_similarity(emb, want, dim=-1)).abs().max().item() <= 1e-3
    assert (emb - want).abs().max().item() <= 3e-2 * want.abs().max().item()


@pytest.fixture(scope="module")
def decoder():
    import twin_weights as TW
    from sonar_amd.text_decoder import (SonarTextDecoderConfig, TextDecoderEngine,
                                        convert_sonar_text_decoder_checkpoint)
    from sonar_amd.text_encoder import VocabularyInfo

    cfg = SonarTextDecoderConfig(model_dim=TW.D, num_decoder_layers=TW.L, num_decoder_attn_heads=TW.H,
                                 ffn_inner_dim=TW.F, vocab_info=VocabularyInfo(size=TW.V), max_seq_len=TW.MAXPOS - 2)
    params = convert_sonar_text_decoder_checkpoint(TW.fairseq_checkpoint("decoder"))
    return TextDecoderEngine(cfg, params, device="cuda:0")


def test_decoder_logits_match_hf_m2m100_decoder(fx, decoder):
    got = decoder.logits(fx["dec_emb"].cuda(), fx["dec_prev"].cuda()).cpu()
    want = fx["dec_logits"]
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1.5e-2 * want.abs().max().item()


def test_greedy_generation_matches_hf_generate(fx, decoder):
    """beam_size = 1 against HF generate(num_beams=1): token for token up to the first position where
    HF's own margin between its best and second-best token is within fp16 reach (then the two greedy
    paths may legitimately part); EOS is forced at the cap by the reference where HF just stops."""
    toks, lens, _ = decoder.generate(fx["gen_emb"].cuda(), fx["gen_prompt"].tolist(), beam_size=1,
                                     max_gen_len=(0, 10))
    toks, lens = toks[:, 0].cpu(), lens[:, 0].cpu()
    want, margin = fx["gen_tokens"], fx["gen_margin"]
    compared = 0
    for i in range(want.shape[0]):
        n = min(int(lens[i]), want.shape[1]) - 1        # the engine's last token may be the forced EOS
        for j in range(n):
            if toks[i, j].item() != want[i, j].item():
                assert margin[i, j].item() < 5e-2, (i, j, margin[i, j].item())   # a near-tie may go either way ...
                break                                                            # ... and the paths part here
            compared += 1
    assert compared >= 80


@pytest.mark.parametrize("fp16_residual", [False, True])
def test_speech_pipeline_matches_hf_composition(fx_speech, fp16_residual):
    """waveform -> sentence embedding through SpeechToEmbeddingModelPipeline.predict against the
    composition of independent implementations of tests/golden/make_golden_speech_e2e.py."""
    from oracle import speech_encoder as OS      # only for the seeded weights
    from sonar_amd.inference_pipelines import SpeechToEmbeddingModelPipeline
    from sonar_amd.speech_encoder import SonarSpeechEncoderConfig, SonarSpeechEncoderModel

    c = fx_speech["config"]
    p = OS.make_synthetic_params(OS.OracleSpeechEncoderConfig(**c), seed=fx_speech["seed"], std=fx_speech["std"])
    cfg = SonarSpeechEncoderConfig(model_dim=c["model_dim"], num_encoder_layers=c["num_layers"],
                                   num_encoder_attn_heads=c["num_heads"], ffn_inner_dim=c["ffn_inner_dim"],
                                   depthwise_conv_kernel_size=c["conv_kernel"], num_decoder_layers=c["pooler_layers"],
                                   num_decoder_attn_heads=c["pooler_heads"], decoder_ffn_inner_dim=c["pooler_ffn_dim"],
                                   max_frames=512)
    model = SonarSpeechEncoderModel(cfg, p, device="cuda:0", dtype=torch.float32, fp16_residual=fp16_residual)
    pipe = SpeechToEmbeddingModelPipeline(model, device=torch.device("cuda:0"))
    out = pipe.predict([w.unsqueeze(0) for w in fx_speech["waveforms"]], batch_size=2).cpu()
    want = fx_speech["embeddings"]
    assert (1 - F.cosine_similarity(out, want, dim=-1)).abs().max().item() <= 1e-3
    assert (out - want).abs().max().item() <= 3e-2 * want.abs().max().item()


@pytest.fixture(scope="module")
def fx_speech():
    return torch.load(os.path.join(GOLDEN, "speech_e2e_twin.pt"), weights_only=False)
