"""GPU parity of the text-encoder hot path (through the C ABI) against the CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cfgs(d=256, heads=4, ffn=512, layers=2, vocab=1000, pooling="mean"):
    from oracle.text_encoder import OracleTextEncoderConfig
    from sonar_amd.text_encoder import SonarTextEncoderConfig, VocabularyInfo

    o = OracleTextEncoderConfig(model_dim=d, num_layers=layers, num_heads=heads, ffn_inner_dim=ffn,
                                vocab_size=vocab, pooling=pooling)
    c = SonarTextEncoderConfig(model_dim=d, num_encoder_layers=layers, num_encoder_attn_heads=heads,
                               ffn_inner_dim=ffn, vocab_info=VocabularyInfo(size=vocab), pooling=pooling,
                               _from_fairseq=True)
    return o, c


def _cos_err(a, b):
    return (1 - F.cosine_similarity(a.float().cpu(), b.float().cpu(), dim=-1)).abs().max().item()


@pytest.mark.parametrize("pooling", ["mean", "max", "last"])
@pytest.mark.parametrize("ragged", [True, False])
@pytest.mark.parametrize("fp16_residual", [False, True])
def test_encoder_small_vs_oracle(pooling, ragged, fp16_residual):
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, SequenceBatch, PaddingMask

    ocfg, cfg = _cfgs(pooling=pooling)
    params = O.make_synthetic_params(ocfg, seed=1234, std=0.08)
    if ragged:
        ids, lens = O.synthetic_batch(9, 3, 70, ocfg.vocab_size, seed=0)
        lens[0] = 1
    else:
        ids, lens = O.synthetic_batch(5, 40, 40, ocfg.vocab_size, seed=1)
    enc_ref, emb_ref = O.text_encoder_forward(params, ocfg, ids, lens if ragged else None)

    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32,
                                             return_encoded_seqs=True, fp16_residual=fp16_residual)
    mask = PaddingMask(lens, ids.shape[1]) if ragged else None
    out = model(SequenceBatch(ids.cuda(), mask))
    torch.cuda.synchronize()
    emb = out.sentence_embeddings
    assert emb.shape == (ids.shape[0], ocfg.model_dim) and emb.dtype == torch.float32
    assert torch.isfinite(emb).all()
    # north_star tolerance: <= 1e-3 (1 - cosine) against the fp32 CPU path
    assert _cos_err(emb, emb_ref) <= 1e-3
    assert (emb.cpu() - emb_ref).abs().max().item() <= 3e-2 * emb_ref.abs().max().item()
    # encoded_seqs on valid positions
    enc = out.encoded_seqs.cpu()
    for i, L in enumerate(lens.tolist() if ragged else [ids.shape[1]] * ids.shape[0]):
        assert _cos_err(enc[i, :L], enc_ref[i, :L]) <= 2e-3
        if ragged:
            assert (enc[i, L:] == 0).all()


def test_encoder_batching_invariance():
    """Reference property (tests/integration_tests/test_text_sonar.py:120-161):
    embeddings do not depend on how sentences are batched."""
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, SequenceBatch, PaddingMask

    ocfg, cfg = _cfgs()
    params = O.make_synthetic_params(ocfg, seed=7, std=0.08)
    ids, lens = O.synthetic_batch(6, 2, 50, ocfg.vocab_size, seed=3)
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32)
    full = model(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))).sentence_embeddings
    for i in range(ids.shape[0]):
        L = int(lens[i])
        one = model(SequenceBatch(ids[i:i + 1, :L].cuda(), None)).sentence_embeddings
        assert (one - full[i:i + 1]).abs().max().item() <= 1e-5 * max(1.0, full.abs().max().item())


def test_encoder_fp16_out_and_errors():
    from oracle import text_encoder as O
    from sonar_amd import _lib
    from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, SequenceBatch

    ocfg, cfg = _cfgs()
    params = O.make_synthetic_params(ocfg, seed=7, std=0.08)
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float16)
    ids, _ = O.synthetic_batch(4, 16, 16, ocfg.vocab_size, seed=5)
    _, emb_ref = O.text_encoder_forward(params, ocfg, ids, None)
    emb = model(SequenceBatch(ids.cuda(), None)).sentence_embeddings
    assert emb.dtype == torch.float16
    assert _cos_err(emb, emb_ref) <= 1e-3
    too_long = torch.zeros(1, cfg.model_max_seq_len + 1, dtype=torch.int64)
    with pytest.raises(_lib.SmiError):
        model(SequenceBatch(too_long.cuda(), None))


@pytest.mark.parametrize("nx,ny,k", [(300, 517, 1), (1000, 3000, 4), (129, 128, 8), (2048, 4096, 2)])
def test_xsim_topk_vs_oracle(nx, ny, k):
    from oracle import xsim as OX
    from sonar_amd import xsim

    g = torch.Generator().manual_seed(nx + ny + k)
    d = 256
    y = torch.randn(ny, d, generator=g)
    x = y[torch.randint(0, ny, (nx,), generator=g)] + 0.8 * torch.randn(nx, d, generator=g)
    xh, yh = x.half(), y.half()
    ref_s, ref_i = OX.cosine_topk(xh.float(), yh.float(), k)
    s, i = xsim.topk(xh.cuda(), yh.cuda(), k)
    torch.cuda.synchronize()
    s, i = s.cpu(), i.cpu().long()
    # scores agree to fp16-normalisation precision; indices agree wherever the gap is resolvable
    assert (s - ref_s).abs().max().item() <= 3e-3
    full = F.normalize(xh.float(), dim=-1) @ F.normalize(yh.float(), dim=-1).T
    picked = full.gather(1, i)
    assert (picked - ref_s).abs().max().item() <= 3e-3
    assert (i[:, 0] == ref_i[:, 0]).float().mean().item() >= 0.995
    for r in range(nx):
        assert len(set(i[r].tolist())) == k


def test_xsim_error_rate_matches_oracle():
    from oracle import xsim as OX
    from sonar_amd import xsim

    x, y, perm = OX.synthetic_pairs(1500, d=256, noise=1.0, seed=2)
    y_al = y[perm]  # aligned: x[i] <-> y_al[i]
    for margin in ("cosine", "ratio"):
        ref = OX.xsim_error_rate(x.half().float(), y_al.half().float(), margin=margin)
        got, _ = xsim.xsim_error(x.half().cuda(), y_al.half().cuda(), margin=margin)
        assert abs(got - ref) <= 2e-3, (margin, got, ref)


def test_predict_pipeline_end_to_end(tmp_path):
    """TextToEmbeddingModelPipeline.predict() on the engine == oracle on the same token ids,
    in input order, for several batchings (reference: test_text_sonar.py:120-161)."""
    import sentencepiece as spm

    from oracle import text_encoder as O
    from sonar_amd.inference_pipelines import TextToEmbeddingModelPipeline
    from sonar_amd.text_encoder import SonarTextTransformerEncoderModel
    from sonar_amd.tokenizer import NllbTokenizer

    words = ["hello", "world", "my", "name", "is", "paul", "teacher", "working", "bonjour", "monde",
             "je", "travaille", "comme", "professeur", "the", "quick", "brown", "fox", "jumps", "over"]
    g = torch.Generator().manual_seed(0)
    corpus = tmp_path / "c.txt"
    with open(corpus, "w") as fh:
        for _ in range(400):
            n = int(torch.randint(2, 12, (1,), generator=g))
            fh.write(" ".join(words[int(i)] for i in torch.randint(0, len(words), (n,), generator=g)) + "\n")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "toy"), vocab_size=48,
                                   model_type="unigram", hard_vocab_limit=False, bos_id=1, eos_id=2,
                                   unk_id=0, pad_id=-1, minloglevel=2)
    tok = NllbTokenizer(str(tmp_path / "toy.model"))
    ocfg, cfg = _cfgs(vocab=tok.vocab_info.size)
    params = O.make_synthetic_params(ocfg, seed=11, std=0.08)
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32)
    pipe = TextToEmbeddingModelPipeline(model, tok, device=torch.device("cuda:0"))
    texts = ["hello world my name is paul", "hello", "the quick brown fox jumps over the teacher",
             "bonjour monde", "je travaille comme professeur", "fox"]
    enc = tok.create_encoder(lang="eng_Latn")
    ref = []
    for t in texts:
        ids = enc(t).unsqueeze(0)
        ref.append(O.text_encoder_forward(params, ocfg, ids, None)[1])
    ref = torch.cat(ref)
    for kw in (dict(batch_size=2), dict(batch_size=1), dict(batch_size=None, batch_max_tokens=5), dict(batch_size=6)):
        out = pipe.predict(texts, source_lang="eng_Latn", **kw)
        assert out.shape == ref.shape and out.device.type == "cuda"
        assert _cos_err(out, ref) <= 1e-3, kw
        assert (out.cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert pipe.predict(texts[:2], "eng_Latn", target_device=torch.device("cpu")).device.type == "cpu"


def test_encoder_empty_sentences_in_batch():
    """seq_lens[i] = 0 is legal at the C ABI: the empty sentence pools to zeros and does not disturb its
    neighbours (packed rows: it simply owns no row); an all-empty batch returns zeros."""
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, SequenceBatch, PaddingMask

    ocfg, cfg = _cfgs()
    params = O.make_synthetic_params(ocfg, seed=7, std=0.08)
    ids, lens = O.synthetic_batch(5, 4, 30, ocfg.vocab_size, seed=8)
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32)
    full = model(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))).sentence_embeddings
    lens0 = lens.clone()
    lens0[1] = 0
    lens0[4] = 0
    out = model(SequenceBatch(ids.cuda(), PaddingMask(lens0, ids.shape[1]))).sentence_embeddings
    assert (out[1] == 0).all() and (out[4] == 0).all()
    for i in (0, 2, 3):
        assert (out[i] - full[i]).abs().max().item() <= 1e-5 * max(1.0, full.abs().max().item())
    none = model(SequenceBatch(ids.cuda(), PaddingMask(torch.zeros_like(lens), ids.shape[1]))).sentence_embeddings
    assert (none == 0).all()


def test_encoder_max_seq_len_514():
    """The longest sequence the model accepts (512 + pad_idx + 1 = 514 positions, factory.py:56-59) next to
    short ones, against the oracle; one more token is refused (text.py:202-209 raises for max_seq_len)."""
    from oracle import text_encoder as O
    from sonar_amd import _lib
    from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, SequenceBatch, PaddingMask

    ocfg, cfg = _cfgs()
    assert cfg.model_max_seq_len == 514
    params = O.make_synthetic_params(ocfg, seed=3, std=0.08)
    g = torch.Generator().manual_seed(12)
    lens = torch.tensor([514, 1, 257, 64])
    ids = torch.zeros(4, 514, dtype=torch.int64)
    for i, L in enumerate(lens.tolist()):
        ids[i, :L] = torch.randint(4, ocfg.vocab_size, (L,), generator=g)
    _, ref = O.text_encoder_forward(params, ocfg, ids, lens)
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32)
    emb = model(SequenceBatch(ids.cuda(), PaddingMask(lens, 514))).sentence_embeddings
    assert _cos_err(emb, ref) <= 1e-3
    with pytest.raises(_lib.SmiError):
        model(SequenceBatch(torch.zeros(1, 515, dtype=torch.int64).cuda(), None))


def test_smi_cast_matches_torch_rounding():
    """The boundary cast (smi_cast): bf16 <-> fp32 <-> fp16, round-to-nearest-even incl. ties, +-0, inf, NaN, subnormals."""
    from sonar_amd import _lib

    g = torch.Generator().manual_seed(5)
    x = torch.randn(100003, generator=g) * torch.logspace(-30, 30, 100003)
    special = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1.0, 1.00390625, 1.01171875,   # ties at bf16
                            3.3895313892515355e+38, 1e-40, -1e-45, 65504.0, 65520.0])
    x = torch.cat([x, special]).cuda()
    for dst in (torch.bfloat16, torch.float16, torch.float32):
        for src in (torch.float32, torch.bfloat16, torch.float16):
            a = x.to(src)
            got = _lib.cast(a, dst)
            want = a.to(dst)
            assert got.dtype == dst and got.shape == a.shape
            nan = torch.isnan(want.float())
            assert torch.equal(torch.isnan(got.float()), nan), (src, dst)
            assert torch.equal(got.float()[~nan], want.float()[~nan]), (src, dst)


def test_bf16_model_at_the_boundary():
    """`dtype=torch.bfloat16` (the reference's pipelines accept any dtype, text.py:36-54, 161-162): bf16 weights are
    exact fp16 operands, embeddings come back in bf16 and stay within the north_star bound of the fp32 oracle run on
    the SAME (bf16-representable) weights."""
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import (PaddingMask, SequenceBatch, SonarTextEncoderConfig,
                                        SonarTextTransformerEncoderModel, VocabularyInfo)

    ocfg = O.OracleTextEncoderConfig(model_dim=256, num_layers=2, num_heads=4, ffn_inner_dim=512, vocab_size=1000)
    cfg = SonarTextEncoderConfig(model_dim=256, num_encoder_layers=2, num_encoder_attn_heads=4, ffn_inner_dim=512,
                                 vocab_info=VocabularyInfo(size=1000), _from_fairseq=True)
    params = {k: v.to(torch.bfloat16) for k, v in O.make_synthetic_params(ocfg, seed=77, std=0.08).items()}
    ids, lens = O.synthetic_batch(9, 5, 40, ocfg.vocab_size, seed=3)
    _, ref = O.text_encoder_forward({k: v.float() for k, v in params.items()}, ocfg, ids, lens)
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.bfloat16)
    out = model(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))).sentence_embeddings
    assert out.dtype == torch.bfloat16 and out.shape == (9, 256)
    err = (1 - F.cosine_similarity(out.float().cpu(), ref, dim=-1)).abs().max().item()
    print(f"bf16 model: max (1 - cos) vs oracle = {err:.2e}")
    assert err <= 1e-3
    # the same weights served as an fp16 model (fp16 residual stream instead of the bf16 model's fp32 one): equal up to
    # the stream's roundings and the final rounding to bf16
    m16 = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float16)
    o16 = m16(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))).sentence_embeddings
    assert (out.float() - o16.float()).abs().max().item() <= 2 ** -6 * o16.float().abs().max().item()


def test_bf16_model_keeps_activations_that_overflow_fp16():
    """A bf16 model's activations have fp32 range (the reference runs bf16 arithmetic under `dtype=torch.bfloat16`,
    text.py:36-54): a residual stream beyond fp16's 65504 -- here a large embedding table times sqrt(d), the
    "massive activation" pattern -- must survive.  The bf16 model keeps its residual stream in fp32 (round 4) and
    stays within the north_star bound of the fp32 oracle; the same weights forced onto an fp16 stream do not."""
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import (PaddingMask, SequenceBatch, SonarTextEncoderConfig,
                                        SonarTextTransformerEncoderModel, VocabularyInfo)

    ocfg = O.OracleTextEncoderConfig(model_dim=256, num_layers=2, num_heads=4, ffn_inner_dim=512, vocab_size=1000)
    cfg = SonarTextEncoderConfig(model_dim=256, num_encoder_layers=2, num_encoder_attn_heads=4, ffn_inner_dim=512,
                                 vocab_info=VocabularyInfo(size=1000), _from_fairseq=True)
    params = O.make_synthetic_params(ocfg, seed=78, std=0.08)
    ek = [k for k in params if "embed" in k and params[k].shape == (1000, 256)]
    assert len(ek) == 1
    params[ek[0]] = params[ek[0]] * 2.5e4                     # |E| up to ~8e3 (an fp16 number); E * 16 up to ~1.3e5
    params = {k: v.to(torch.bfloat16) for k, v in params.items()}
    ids, lens = O.synthetic_batch(7, 5, 40, ocfg.vocab_size, seed=4)
    x_max = (params[ek[0]].float().abs().max() * 16).item()
    assert x_max > 65504 and params[ek[0]].float().abs().max().item() < 65504
    _, ref = O.text_encoder_forward({k: v.float() for k, v in params.items()}, ocfg, ids, lens)
    batch = SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))
    out = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.bfloat16)(batch).sentence_embeddings
    assert out.dtype == torch.bfloat16 and torch.isfinite(out.float()).all()
    err = (1 - F.cosine_similarity(out.float().cpu(), ref, dim=-1)).abs().max().item()
    print(f"bf16 model, residual stream up to {x_max:.3g}: max (1 - cos) vs oracle = {err:.2e}")
    assert err <= 1e-3
    forced = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.bfloat16, fp16_residual=True)
    bad = forced(batch).sentence_embeddings.float()
    assert not torch.isfinite(bad).all()                      # what the fp32 stream is for


@pytest.mark.parametrize("fp16_residual", [True, False])
def test_small_batch_layer_schedule_vs_oracle(fp16_residual):
    """Small batches (a handful of row tiles, F a multiple of 1024) take the decoder-shaped layer: split-K attention-output
    and FFN-output projections whose slabs the fused sum + LayerNorm kernel folds into the (fp16 or fp32) residual stream.
    The reference's default call (`batch_size=5`, text.py:178) lives on this path."""
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import (PaddingMask, SequenceBatch, SonarTextEncoderConfig,
                                        SonarTextTransformerEncoderModel, VocabularyInfo)

    ocfg = O.OracleTextEncoderConfig(model_dim=256, num_layers=3, num_heads=4, ffn_inner_dim=2048, vocab_size=500)
    cfg = SonarTextEncoderConfig(model_dim=256, num_encoder_layers=3, num_encoder_attn_heads=4, ffn_inner_dim=2048,
                                 vocab_info=VocabularyInfo(size=500), _from_fairseq=True)
    params = O.make_synthetic_params(ocfg, seed=21, std=0.06)
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32, fp16_residual=fp16_residual)
    for n, lo, hi in ((5, 8, 64), (1, 3, 3), (40, 2, 30)):
        ids, lens = O.synthetic_batch(n, lo, hi, ocfg.vocab_size, seed=n)
        _, ref = O.text_encoder_forward(params, ocfg, ids, lens)
        emb = model(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))).sentence_embeddings
        err = (1 - F.cosine_similarity(emb.float().cpu(), ref, dim=-1)).abs().max().item()
        assert err <= 1e-3, (n, err)
        assert (emb.float().cpu() - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
