"""GPU: the part of the reference's configuration space outside the released models' shapes -- restating
/root/reference/tests/unit_tests/test_low_dimension_text_models.py:20-73 (a decoder with model_dim 32 / 4 heads of 8
conditioned on 256-d vectors; an encoder with model_dim 32 pooled by attention into 256 dimensions) as parity tests
against the fp32 oracle.  These shapes run on the library's generic-dimension kernels (csrc/flex.hip); the
conditioning width input_dim != model_dim is also covered on the MFMA path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dec_cfgs(d, heads, ffn, layers, vocab, input_dim, max_seq_len=64):
    from oracle.text_decoder import OracleTextDecoderConfig
    from sonar_amd.text_decoder import SonarTextDecoderConfig
    from sonar_amd.text_encoder import VocabularyInfo

    o = OracleTextDecoderConfig(model_dim=d, num_layers=layers, num_heads=heads, ffn_inner_dim=ffn, vocab_size=vocab,
                                max_seq_len=max_seq_len, input_dim=input_dim)
    c = SonarTextDecoderConfig(model_dim=d, num_decoder_layers=layers, num_decoder_attn_heads=heads, ffn_inner_dim=ffn,
                               vocab_info=VocabularyInfo(size=vocab), max_seq_len=max_seq_len, input_dim=input_dim)
    return o, c


def test_low_dim_decoder_toy_arch():
    """test_low_dimension_text_models.py:46-73: `toy` arch (model_dim 32, 4 heads, F 128, V 1024), input_dim 256,
    prefix [0, 1, 2, 3, 4] for 3 sentences -> logits [3, 5, V]; here also held to the oracle, then decoded."""
    from oracle import text_decoder as OD
    from sonar_amd.text_decoder import TextDecoderEngine, get_text_decoder_config

    cfg = get_text_decoder_config("toy")
    cfg.input_dim = 256
    assert (cfg.model_dim, cfg.num_decoder_attn_heads, cfg.ffn_inner_dim, cfg.vocab_info.size) == (32, 4, 128, 1024)
    ocfg = OD.OracleTextDecoderConfig(model_dim=32, num_layers=2, num_heads=4, ffn_inner_dim=128, vocab_size=1024,
                                      max_seq_len=512, input_dim=256)
    params = OD.make_synthetic_params(ocfg, seed=31, std=0.2)
    eng = TextDecoderEngine(cfg, params, device="cuda:0")
    embeds = torch.rand(3, 256, generator=torch.Generator().manual_seed(1))
    prefix = torch.tensor([[0, 1, 2, 3, 4]] * 3)
    got = eng.logits(embeds.cuda(), prefix.cuda()).cpu()
    assert got.shape == (3, 5, 1024)
    ref = OD.decoder_logits(params, ocfg, embeds, prefix)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"toy decoder (d 32, 4 heads of 8, input_dim 256): max |logit diff| / scale = {err:.2e}")
    assert err <= 1e-4                       # the generic kernels compute in fp32
    # a longer prefix: 40 positions through the fp32 KV cache and the ancestry-gathered attention
    prev = torch.randint(4, 1024, (3, 40), generator=torch.Generator().manual_seed(2))
    prev[:, 0] = 3
    ref = OD.decoder_logits(params, ocfg, embeds, prev)
    got = eng.logits(embeds.cuda(), prev.cuda()).cpu()
    assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    # and generation: the device beam search runs on top of the generic step unchanged
    for beam in (1, 5):
        kw = dict(beam_size=beam, max_gen_len=(0, 11))
        want = OD.beam_search(params, ocfg, embeds, [3, 900], **kw)
        toks, lens, scores = eng.generate(embeds.cuda(), [3, 900], **kw)
        toks, lens, scores = toks.cpu(), lens.cpu(), scores.cpu()
        for i in range(3):
            assert toks[i, 0, : int(lens[i, 0])].tolist() == want[i][0].seq.tolist(), (beam, i)
            assert abs(scores[i, 0].item() - want[i][0].score) <= 1e-4
    # the sampling generator too (top-k 1 == greedy)
    from sonar_amd.generation import TopKSampler

    st, sl, _ = eng.sample(embeds.cuda(), [3, 900], TopKSampler(1), seed=3, max_gen_len=(0, 11))
    g = OD.beam_search(params, ocfg, embeds, [3, 900], beam_size=1, max_gen_len=(0, 11))
    for i in range(3):
        assert st[i, : int(sl[i])].tolist() == g[i][0].seq.tolist()


@pytest.mark.parametrize("d,heads,ffn", [(48, 3, 100), (64, 2, 96), (96, 12, 200)])
def test_generic_decoder_shapes(d, heads, ffn):
    """head_dim 16 / 32 / 8, odd FFN widths, vocabulary not a multiple of anything."""
    from oracle import text_decoder as OD
    from sonar_amd.text_decoder import TextDecoderEngine

    ocfg, cfg = _dec_cfgs(d, heads, ffn, 2, 777, d + 24)
    params = OD.make_synthetic_params(ocfg, seed=d, std=0.15)
    eng = TextDecoderEngine(cfg, params, device="cuda:0")
    g = torch.Generator().manual_seed(d)
    emb = torch.randn(4, d + 24, generator=g) * 0.3
    prev = torch.randint(4, 777, (4, 9), generator=g)
    ref = OD.decoder_logits(params, ocfg, emb, prev)
    got = eng.logits(emb.cuda(), prev.cuda()).cpu()
    assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    want = OD.beam_search(params, ocfg, emb, [3, 700], beam_size=3, max_gen_len=(0, 8))
    toks, lens, _ = eng.generate(emb.cuda(), [3, 700], beam_size=3, max_gen_len=(0, 8))
    for i in range(4):
        assert toks[i, 0, : int(lens[i, 0])].tolist() == want[i][0].seq.tolist()


def test_input_dim_differs_from_model_dim_on_the_mfma_path():
    """factory.py:264, 276-282: the cross-attention's K / V projections take `input_dim` columns.  model_dim 256 /
    4 heads of 64 stays on the MFMA engines; the conditioning vectors are 128- and 320-dimensional."""
    from oracle import text_decoder as OD
    from sonar_amd.text_decoder import TextDecoderEngine

    for input_dim in (128, 320):
        ocfg, cfg = _dec_cfgs(256, 4, 512, 2, 1000, input_dim)
        params = OD.make_synthetic_params(ocfg, seed=input_dim, std=0.09)
        eng = TextDecoderEngine(cfg, params, device="cuda:0")
        g = torch.Generator().manual_seed(7)
        emb = torch.randn(5, input_dim, generator=g) * 0.3
        prev = torch.randint(4, 1000, (5, 10), generator=g)
        prev[:, 0] = 3
        ref = OD.decoder_logits(params, ocfg, emb, prev)
        got = eng.logits(emb.cuda(), prev.cuda()).cpu()
        assert (got - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
        with pytest.raises(ValueError):
            eng.logits(torch.randn(5, 256).cuda(), prev.cuda())


def _enc_cfgs(**kw):
    """(oracle config, engine config) for one SonarTextEncoderConfig variant."""
    from oracle.text_encoder import OracleTextEncoderConfig
    from sonar_amd.text_encoder import SonarTextEncoderConfig, VocabularyInfo

    d, heads, ffn, layers, vocab = kw.pop("d"), kw.pop("heads"), kw.pop("ffn"), kw.pop("layers"), kw.pop("vocab")
    pooling = kw.pop("pooling", "mean")
    embedding_dim = kw.pop("embedding_dim", None)
    dec_layers, dec_heads, dec_ffn = kw.pop("dec_layers", 0), kw.pop("dec_heads", 0), kw.pop("dec_ffn", None)
    o = OracleTextEncoderConfig(model_dim=d, num_layers=layers, num_heads=heads, ffn_inner_dim=ffn, vocab_size=vocab,
                                pooling=pooling, embedding_dim=embedding_dim, pooler_layers=dec_layers, pooler_heads=dec_heads,
                                pooler_ffn_dim=dec_ffn or ffn, **kw)
    c = SonarTextEncoderConfig(model_dim=d, num_encoder_layers=layers, num_decoder_layers=dec_layers,
                               num_encoder_attn_heads=heads, num_decoder_attn_heads=dec_heads or heads, ffn_inner_dim=ffn,
                               decoder_ffn_inner_dim=dec_ffn, vocab_info=VocabularyInfo(size=vocab), pooling=pooling,
                               embedding_dim=embedding_dim, _from_fairseq=True, **kw)
    return o, c


def _run_encoder(o, c, ragged=True, seed=0, n=6, smax=19):
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import PaddingMask, SequenceBatch, SonarTextTransformerEncoderModel

    params = O.make_synthetic_params(o, seed=seed + 1, std=0.12)
    ids, lens = O.synthetic_batch(n, 3, smax, o.vocab_size, seed=seed)
    if not ragged:
        lens = None
        ids = ids.clamp(min=4)
    enc_ref, ref = O.text_encoder_forward(params, o, ids, lens)
    model = SonarTextTransformerEncoderModel(c, params, device="cuda:0", dtype=torch.float32, return_encoded_seqs=True)
    out = model(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]) if lens is not None else None))
    return out, enc_ref, ref, lens


def test_low_dim_encoder_attention_pooling():
    """test_low_dimension_text_models.py:20-43: `basic` with model_dim 32, embedding_dim 256, 5 encoder layers, 2 pooler
    layers, pooling "attention" (16 heads of 2 in the encoder, 16 of 16 in the pooler; post-norm pooler layers because
    `basic` has normalize_before False), tokens [[0, 1, 2, 3, 4]] x 3 -> sentence_embeddings [3, 256]."""
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import SequenceBatch, SonarTextTransformerEncoderModel, get_text_encoder_config

    cfg = get_text_encoder_config("basic")
    cfg.model_dim, cfg.embedding_dim, cfg.num_encoder_layers, cfg.num_decoder_layers, cfg.pooling = 32, 256, 5, 2, "attention"
    cfg.ffn_inner_dim, cfg.vocab_info.size = 64, 300          # (the reference test keeps F = 8192 and V = 256206: same code path)
    o = O.OracleTextEncoderConfig(model_dim=32, num_layers=5, num_heads=16, ffn_inner_dim=64, vocab_size=300,
                                  pooling="attention", embedding_dim=256, pooler_layers=2, pooler_heads=16, pooler_ffn_dim=64)
    params = O.make_synthetic_params(o, seed=9, std=0.15)
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32)
    tokens = torch.tensor([[0, 1, 2, 3, 4]] * 3)
    out = model(SequenceBatch(tokens.cuda(), None)).sentence_embeddings
    assert out.shape == (3, 256)
    _, ref = O.text_encoder_forward(params, o, tokens, None)
    err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"low-dim encoder, attention pooling 32 -> 256: max |diff| / scale = {err:.2e}")
    assert err <= 1e-4


@pytest.mark.parametrize("variant", ["attention_pre", "attention_post_ragged", "mean_headdim8", "max_learned_pos",
                                     "last_no_pos_ln_embed", "normalize_before_mean", "no_scale"])
def test_generic_encoder_option_matrix(variant):
    """Every builder option of SonarTextEncoderFactory.create_model (factory.py:72-120) against the oracle."""
    base = dict(d=48, heads=6, ffn=80, layers=2, vocab=211)
    v = {
        "attention_pre": dict(pooling="attention", embedding_dim=96, dec_layers=2, dec_heads=4, dec_ffn=72, normalize_before=True),
        "attention_post_ragged": dict(pooling="attention", embedding_dim=40, dec_layers=1, dec_heads=5),
        "mean_headdim8": dict(pooling="mean"),
        "max_learned_pos": dict(pooling="max", learned_pos=True),
        "last_no_pos_ln_embed": dict(pooling="last", no_token_positional_embeddings=True, layernorm_embedding=True),
        "normalize_before_mean": dict(pooling="mean", normalize_before=True),
        "no_scale": dict(pooling="mean", no_scale_embedding=True),
    }[variant]
    o, c = _enc_cfgs(**base, **v)
    out, enc_ref, ref, lens = _run_encoder(o, c, ragged=variant != "attention_pre", seed=len(variant))
    emb = out.sentence_embeddings.cpu()
    assert emb.shape == ref.shape
    assert (emb - ref).abs().max().item() <= 1e-4 * ref.abs().max().item(), variant
    enc = out.encoded_seqs.cpu()
    if lens is not None:
        keep = (torch.arange(enc.shape[1]).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)
        assert (enc[~keep.expand_as(enc)] == 0).all()
        enc_ref = torch.where(keep, enc_ref, torch.zeros_like(enc_ref))
    assert (enc - enc_ref).abs().max().item() <= 1e-4 * enc_ref.abs().max().item()


def test_generic_encoder_through_the_pipeline_types():
    """fp16 / bf16 outputs and the out-of-vocabulary report work on the generic path too."""
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import PaddingMask, SequenceBatch, SonarTextTransformerEncoderModel

    o, c = _enc_cfgs(d=40, heads=5, ffn=64, layers=1, vocab=97)
    params = O.make_synthetic_params(o, seed=4, std=0.1)
    ids, lens = O.synthetic_batch(4, 3, 9, 97, seed=2)
    _, ref = O.text_encoder_forward(params, o, ids, lens)
    for dt in (torch.float16, torch.bfloat16):
        m = SonarTextTransformerEncoderModel(c, params, device="cuda:0", dtype=dt)
        out = m(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))).sentence_embeddings
        assert out.dtype == dt
        assert (1 - F.cosine_similarity(out.float().cpu(), ref, dim=-1)).abs().max().item() <= 1e-3
    bad = ids.clone()
    bad[0, 0] = 97
    # the model object reports out-of-vocabulary ids from forward() itself, as the reference's embedding lookup does ...
    with pytest.raises(IndexError):
        m(SequenceBatch(bad.cuda(), PaddingMask(lens, ids.shape[1])))
    m(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1])))          # the flag was cleared: a valid batch passes
    # ... unless its caller queues batches and checks once at the end (predict(), bench.py)
    with pytest.raises(IndexError):
        with m.deferring_check():
            m(SequenceBatch(bad.cuda(), PaddingMask(lens, ids.shape[1])))
            m(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1])))
