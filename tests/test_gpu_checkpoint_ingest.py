"""Checkpoint ingest on the GPU: fairseq1-layout files written with torch.save are loaded
through `load_sonar_*` (key conversion, control-token row permutation, moved LayerNorm, tied projection,
packed cache) and must drive the engine to the same outputs as the fairseq2-named weights handed over in
memory -- and to the CPU oracle's."""
import pytest
import torch

from tests.ckpt_layouts import speech_encoder_to_fairseq1, text_decoder_to_fairseq1, text_encoder_to_fairseq1

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _cache(tmp_path, monkeypatch):
    monkeypatch.setenv("SONAR_AMD_CACHE", str(tmp_path / "packed"))


def test_text_encoder_from_disk(tmp_path):
    from oracle import text_encoder as OE
    from sonar_amd.text_encoder import (PaddingMask, SequenceBatch, SonarTextEncoderConfig, VocabularyInfo,
                                        load_sonar_text_encoder)

    ocfg = OE.OracleTextEncoderConfig(model_dim=256, num_layers=2, num_heads=4, ffn_inner_dim=512, vocab_size=1000)
    cfg = SonarTextEncoderConfig(model_dim=256, num_encoder_layers=2, num_encoder_attn_heads=4, ffn_inner_dim=512,
                                 vocab_info=VocabularyInfo(size=1000), _from_fairseq=True)
    params = OE.make_synthetic_params(ocfg, seed=11, std=0.08)
    f = tmp_path / "sonar_text_encoder.pt"
    torch.save(text_encoder_to_fairseq1(params), f)
    ids, lens = OE.synthetic_batch(6, 3, 40, ocfg.vocab_size, seed=0)
    ids[0, :4] = torch.tensor([0, 1, 2, 3])          # the permuted control rows are actually read
    lens[0] = max(int(lens[0]), 4)
    _, ref = OE.text_encoder_forward(params, ocfg, ids, lens)
    batch = SequenceBatch(ids.to(DEV), PaddingMask(lens, ids.shape[1]))
    outs = []
    for expect in ("miss", "hit"):                   # the second load comes from the packed cache
        st = {}
        model = load_sonar_text_encoder(str(f), config=cfg, device=DEV, dtype=torch.float32, load_stats=st)
        assert st["cache"] == expect
        outs.append(model(batch).sentence_embeddings.float().cpu())
        del model
    assert torch.equal(outs[0], outs[1])             # cache hit == fresh conversion, bit for bit
    cos = torch.nn.functional.cosine_similarity(outs[0], ref, dim=-1)
    assert (1 - cos).max().item() <= 1e-4
    assert (outs[0] - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()


def test_text_decoder_from_disk_is_tied(tmp_path):
    """test_tied_weights.py:40-78 on the engine: the output projection IS the (permuted) embedding table."""
    from oracle import text_decoder as OD
    from sonar_amd.text_decoder import SonarTextDecoderConfig, load_sonar_text_decoder
    from sonar_amd.text_encoder import VocabularyInfo

    ocfg = OD.OracleTextDecoderConfig(model_dim=256, num_layers=2, num_heads=4, ffn_inner_dim=512, vocab_size=1000,
                                      max_seq_len=32)
    cfg = SonarTextDecoderConfig(model_dim=256, num_decoder_layers=2, num_decoder_attn_heads=4, ffn_inner_dim=512,
                                 vocab_info=VocabularyInfo(size=1000), max_seq_len=32)
    params = OD.make_synthetic_params(ocfg, seed=12, std=0.09)
    emb = torch.randn(3, 256, generator=torch.Generator().manual_seed(1)) * 0.3
    prev = torch.tensor([[3, 0, 1, 2, 700], [3, 5, 6, 7, 8], [3, 1, 1, 9, 3]])   # control tokens as inputs too
    ref = OD.decoder_logits(params, ocfg, emb, prev)
    scale = ref.abs().max().item()
    got = []
    for tied_storage in (True, False):
        f = tmp_path / f"dec{int(tied_storage)}.pt"
        torch.save(text_decoder_to_fairseq1(params, tied_storage), f)
        model = load_sonar_text_decoder(str(f), config=cfg, device=DEV)
        lg = model.engine.logits(emb.to(DEV), prev.to(DEV)).cpu()
        # columns 0..3 (PAD, UNK, BOS, EOS) are the rows the converter permuted: tied projection
        assert (lg - ref).abs().max().item() <= 1.5e-2 * scale
        assert (lg[..., :4] - ref[..., :4]).abs().max().item() <= 1.5e-2 * scale
        got.append(lg)
        del model
    assert torch.equal(got[0], got[1])


def test_speech_encoder_from_disk(tmp_path):
    from oracle import speech_encoder as OS
    from sonar_amd.speech_encoder import SonarSpeechEncoderConfig, load_sonar_speech_encoder
    from sonar_amd.text_encoder import PaddingMask, SequenceBatch

    so = OS.OracleSpeechEncoderConfig(model_dim=256, num_layers=2, num_heads=4, ffn_inner_dim=512, conv_kernel=7,
                                      pooler_layers=2, pooler_heads=4, pooler_ffn_dim=384, pooler_vocab=64)
    scfg = SonarSpeechEncoderConfig(model_dim=256, num_encoder_layers=2, num_encoder_attn_heads=4, ffn_inner_dim=512,
                                    depthwise_conv_kernel_size=7, num_decoder_layers=2, num_decoder_attn_heads=4,
                                    decoder_ffn_inner_dim=384, max_frames=512)
    params = OS.make_synthetic_params(so, seed=13, std=0.06)
    f = tmp_path / "spenc.eng.pt"
    torch.save(speech_encoder_to_fairseq1(params), f)
    g = torch.Generator().manual_seed(3)
    fb = torch.randn(3, 60, 80, generator=g)
    lens = [60, 44, 52]
    for i, l in enumerate(lens):
        fb[i, l:] = 0
    ref = OS.speech_encoder_forward(params, so, fb, torch.tensor(lens))
    ref = ref[1] if isinstance(ref, tuple) else ref
    outs = []
    for expect in ("miss", "hit"):
        st = {}
        model = load_sonar_speech_encoder(str(f), config=scfg, device=DEV, dtype=torch.float32, load_stats=st)
        assert st["cache"] == expect
        out = model(SequenceBatch(fb.to(DEV), PaddingMask(torch.tensor(lens, dtype=torch.int32), 60)))
        outs.append(out.sentence_embeddings.float().cpu())
        del model
    assert torch.equal(outs[0], outs[1])
    cos = torch.nn.functional.cosine_similarity(outs[0], ref, dim=-1)
    assert (1 - cos).max().item() <= 1e-3


def test_card_names_resolve_to_loaded_pipelines(tmp_path, monkeypatch):
    """The reference's constructor calls with card NAMES (text.py:157-167) work once the files are in
    $SONAR_CHECKPOINT_DIR -- here tiny synthetic ones with an explicit config via the loaders' seam."""
    import sentencepiece as spm

    from oracle import text_encoder as OE
    from sonar_amd import cards
    from sonar_amd import text_encoder as TE
    from sonar_amd.inference_pipelines import TextToEmbeddingModelPipeline

    corpus = tmp_path / "c.txt"
    corpus.write_text("\n".join(["hello world my name is paul and i am a teacher"] * 80))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "sp"), vocab_size=30, model_type="unigram",
                                   hard_vocab_limit=False, bos_id=1, eos_id=2, unk_id=0, pad_id=-1, minloglevel=2)
    (tmp_path / cards.NLLB_SPM).write_bytes((tmp_path / "sp.model").read_bytes())
    from sonar_amd.tokenizer import NllbTokenizer

    vocab = NllbTokenizer(str(tmp_path / cards.NLLB_SPM)).vocab_info.size
    ocfg = OE.OracleTextEncoderConfig(model_dim=256, num_layers=1, num_heads=4, ffn_inner_dim=512, vocab_size=vocab)
    cfg = TE.SonarTextEncoderConfig(model_dim=256, num_encoder_layers=1, num_encoder_attn_heads=4, ffn_inner_dim=512,
                                    vocab_info=TE.VocabularyInfo(size=vocab), _from_fairseq=True)
    params = OE.make_synthetic_params(ocfg, seed=2, std=0.08)
    torch.save(text_encoder_to_fairseq1(params), tmp_path / "sonar_text_encoder.pt")
    monkeypatch.setenv("SONAR_CHECKPOINT_DIR", str(tmp_path))
    monkeypatch.setitem(TE.TEXT_ENCODER_ARCHS, "basic", lambda: cfg)    # the card says arch "basic"
    pipe = TextToEmbeddingModelPipeline("text_sonar_basic_encoder", "text_sonar_basic_encoder", device=torch.device(DEV))
    out = pipe.predict(["hello world", "my name is paul"], source_lang="eng_Latn")
    assert out.shape == (2, 256) and torch.isfinite(out.float()).all()
    monkeypatch.delenv("SONAR_CHECKPOINT_DIR")
    monkeypatch.setenv("HOME", str(tmp_path / "nohome"))
    with pytest.raises(FileNotFoundError, match="SONAR_CHECKPOINT_DIR"):
        TextToEmbeddingModelPipeline("text_sonar_basic_encoder", "text_sonar_basic_encoder", device=torch.device(DEV))
