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
