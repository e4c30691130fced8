"""GPU parity of the EmbeddingToText hot path (decoder step + on-device beam search, through the
C ABI) against the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfgs(d=256, heads=4, ffn=512, layers=2, vocab=1000, max_seq_len=64):
    from oracle.text_decoder import OracleTextDecoderConfig
    from sonar_amd.text_decoder import SonarTextDecoderConfig
    from sonar_amd.text_encoder import VocabularyInfo

    o = OracleTextDecoderConfig(model_dim=d, num_layers=layers, num_heads=heads, ffn_inner_dim=ffn,
                                vocab_size=vocab, max_seq_len=max_seq_len)
    c = SonarTextDecoderConfig(model_dim=d, num_decoder_layers=layers, num_decoder_attn_heads=heads,
                               ffn_inner_dim=ffn, vocab_info=VocabularyInfo(size=vocab), max_seq_len=max_seq_len)
    return o, c


@pytest.fixture(scope="module")
def setup():
    from oracle import text_decoder as OD
    from sonar_amd.text_decoder import TextDecoderEngine

    ocfg, cfg = _cfgs()
    params = OD.make_synthetic_params(ocfg, seed=4321, std=0.09)
    eng = TextDecoderEngine(cfg, params, device="cuda:0")
    return OD, ocfg, params, eng


def test_decoder_logits_vs_oracle(setup):
    OD, ocfg, params, eng = setup
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(5, ocfg.model_dim, generator=g) * 0.3
    prev = torch.randint(4, ocfg.vocab_size, (5, 11), generator=g)
    prev[:, 0] = 3
    ref = OD.decoder_logits(params, ocfg, emb, prev)
    got = eng.logits(emb.cuda(), prev.cuda()).cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 1.5e-2 * scale, ((got - ref).abs().max().item(), scale)
    # fp16 embeddings in, same answer
    got16 = eng.logits(emb.half().cuda(), prev.cuda()).cpu()
    assert (got16 - ref).abs().max().item() <= 2e-2 * scale
    # the reference test's shape: one sentence, prev tokens [[3, 333]] (test_text_sonar.py:69)
    one = eng.logits(emb[:1].cuda(), torch.tensor([[3, 333]]).cuda()).cpu()
    assert (one - OD.decoder_logits(params, ocfg, emb[:1], torch.tensor([[3, 333]]))).abs().max().item() <= 1.5e-2 * scale


def _rescored(OD, params, ocfg, e, prompt, seq):
    full = torch.tensor([list(prompt) + seq])
    lp = torch.log_softmax(OD.decoder_logits(params, ocfg, e.unsqueeze(0), full[:, :-1]), dim=-1)
    return lp[0, torch.arange(full.shape[1] - 1), full[0, 1:]].sum().item()


# "Exact token-id match for greedy decode" (BASELINE north_star), stated honestly: every token of every
# best hypothesis equals the fp32 CPU oracle's, EXCEPT for a sentence whose decision margin -- measured
# BY THE ORACLE on its own candidate lists (tests/neartie.py; the engine's report, smi_text_decoder_last_margins,
# is only cross-checked against it) -- is below EPS_REL of the logit range, i.e. where two candidates are
# closer than fp16 arithmetic can separate.  Such a sentence must still return a hypothesis whose oracle
# score equals the oracle's best within the same epsilon.
EPS_REL = 1e-3


def _logit_range(OD, params, ocfg, emb, prompt):
    lg = OD.decoder_logits(params, ocfg, emb, torch.tensor([list(prompt)] * emb.shape[0]))
    return (lg.max() - lg.min()).item()


@pytest.mark.parametrize("beam", [1, 3, 5])
def test_beam_search_vs_oracle(setup, beam):
    from tests.neartie import check_engine_margin, oracle_excuses

    OD, ocfg, params, eng = setup
    g = torch.Generator().manual_seed(10 + beam)
    n = 24
    emb = torch.randn(n, ocfg.model_dim, generator=g) * 0.3
    prompt = [3, 700]
    kw = dict(beam_size=beam, max_gen_len=(0, 13))
    om = []
    ref = OD.beam_search_incremental(params, ocfg, emb, prompt, margins_out=om, **kw)
    toks, lens, scores = eng.generate(emb.cuda(), prompt, **kw)
    margins = eng.last_margins(n).cpu()
    torch.cuda.synchronize()
    toks, lens, scores = toks.cpu(), lens.cpu(), scores.cpu()
    eps = EPS_REL * _logit_range(OD, params, ocfg, emb, prompt)
    excused = []
    for i in range(n):
        L = int(lens[i, 0])
        seq = toks[i, 0, :L].tolist()
        assert L >= 2 and seq[-1] == 3 and 0 not in seq and all(t >= 0 for t in seq)
        assert (toks[i, 0, L:] == -1).all()
        # the engine's best hypothesis is a valid hypothesis with the score the oracle assigns to it
        total = _rescored(OD, params, ocfg, emb[i], prompt, seq)
        norm = total / (len(prompt) + L - 1)
        assert abs(norm - scores[i, 0].item()) <= 5e-3, (norm, scores[i, 0].item())
        # hypotheses come out best first
        k = int((lens[i] > 0).sum())
        assert k == beam
        assert all(scores[i, j] >= scores[i, j + 1] - 1e-6 for j in range(k - 1))
        assert margins[i, 0] >= 0 and (beam == 1 or margins[i, 1] >= 0)
        check_engine_margin(margins[i], om[i], eps, f"beam {beam}, sentence {i}")
        if seq != ref[i][0].seq.tolist():
            step_gap, final_gap = om[i][:2]
            assert oracle_excuses(om[i], eps), (
                f"sentence {i}: tokens differ from the oracle although every decision margin the ORACLE "
                f"measured (step {step_gap:.3e}, final {final_gap:.3e}) is above eps {eps:.3e}; engine's: {margins[i].tolist()}")
            # a measured near-tie: the returned hypothesis must be as good as the oracle's best
            assert abs(norm - ref[i][0].score) <= 2 * eps, (norm, ref[i][0].score)
            excused.append((i, step_gap, final_gap))
    print(f"beam {beam}: {n - len(excused)}/{n} best hypotheses token-identical to the oracle; "
          f"measured near-ties (< {eps:.2e}): {excused}")
    # near-ties are rare events, not a blanket excuse
    assert len(excused) <= max(1, n // 8), excused


def test_greedy_margins_are_the_top2_logprob_gaps(setup):
    """The margin the exactness test relies on is itself checked against the oracle: for beam 1 it must
    equal the smallest top-1 / top-2 log-prob gap along the oracle's greedy path."""
    OD, ocfg, params, eng = setup
    emb = torch.randn(6, ocfg.model_dim, generator=torch.Generator().manual_seed(21)) * 0.3
    prompt = [3, 700]
    toks, lens, _ = eng.generate(emb.cuda(), prompt, beam_size=1, max_gen_len=(0, 9))
    margins = eng.last_margins(6).cpu()
    toks, lens = toks.cpu(), lens.cpu()
    for i in range(6):
        seq = toks[i, 0, : int(lens[i, 0])].tolist()
        full = torch.tensor([prompt + seq])
        lp = torch.log_softmax(OD.decoder_logits(params, ocfg, emb[i:i + 1], full[:, :-1]), dim=-1)[0]
        lp[:, 0] = -torch.inf                      # PAD is never a candidate
        gaps = []
        for t in range(len(prompt) - 1, full.shape[1] - 1):
            row = lp[t].clone()
            if t == len(prompt) - 1:               # min_gen_len = 1: EOS blocked on the first free step
                row[3] = -torch.inf
            if t == len(prompt) - 1 + 8:           # forced EOS at the cap: no decision
                continue
            top2 = row.topk(2).values
            gaps.append((top2[0] - top2[1]).item())
        assert abs(min(gaps) - margins[i, 0].item()) <= 2e-2, (min(gaps), margins[i, 0].item())
        assert margins[i, 1].item() == float("inf")


def test_generation_cap_follows_source_length(setup):
    """fairseq2: max_gen_len = a * source_len + b.  A sentence vector handed to the generator as
    `source_seqs` [n, model_dim] has "source length" model_dim, so the default (1, 128) cap is the
    decoder's max_seq_len, not 129 tokens; text / speech sources pass their own length."""
    OD, ocfg, params, eng = setup
    emb = (torch.randn(2, ocfg.model_dim, generator=torch.Generator().manual_seed(31)) * 0.3).cuda()
    prompt = [3, 700]
    # EOS blocked for the whole run: the output length is the cap itself
    toks, lens, _ = eng.generate(emb, prompt, beam_size=1, min_gen_len=1000 if False else 40, max_gen_len=(1, 8),
                                 source_len=32)
    assert int(lens.max()) == 40 and toks.shape[2] == 2 + 40           # 1 * 32 + 8
    toks, lens, _ = eng.generate(emb, prompt, beam_size=1, min_gen_len=60, max_gen_len=(1, 8))
    assert toks.shape[2] == ocfg.max_seq_len and int(lens.max()) == ocfg.max_seq_len - 2   # capped by the model
    with pytest.raises(ValueError):
        eng.generate(emb, prompt, beam_size=1, max_gen_len=(0, 0))
    ref = OD.beam_search(params, ocfg, emb.cpu().float(), prompt, beam_size=1, min_gen_len=40, max_gen_len=(1, 8),
                         source_len=32)
    assert len(ref[0][0].seq) == 40


def test_generate_is_repeatable_across_calls(setup):
    """Workspace reuse across calls (grow-only buffers, KV cache, tile statistics) must not leak state:
    the same request gives the same hypotheses after other requests ran in between."""
    OD, ocfg, params, eng = setup
    emb = (torch.randn(6, ocfg.model_dim, generator=torch.Generator().manual_seed(77)) * 0.3).cuda()
    kw = dict(beam_size=3, max_gen_len=(0, 11))
    first = [t.cpu() for t in eng.generate(emb, [3, 702], **kw)]
    other = [t.cpu() for t in eng.generate(emb, [3, 703], **kw)]
    eng.generate(emb[:2], [3, 702], beam_size=5, max_gen_len=(0, 21))
    again = [t.cpu() for t in eng.generate(emb, [3, 702], **kw)]
    for a, b in zip(first, again):
        assert torch.equal(a, b)
    assert not torch.equal(first[0], other[0])


@pytest.mark.parametrize("chains", [2, 3])
def test_independent_chains_return_the_single_chain_hypotheses(setup, chains):
    """smi_text_decoder_set_chains: a batch decoded as 2 / 3 independent sentence groups (own workspace, KV cache,
    beam state, stream and host thread each) must return what the single chain returns -- hypotheses, lengths,
    scores, decision margins -- sentence for sentence and in input order, an uneven last group included.
    With the per-launch tile choices pinned (the split-K part count and the FFN engine follow a call's row count
    otherwise, i.e. the fp32 summation order would differ) the results are BIT-identical."""
    OD, ocfg, params, eng = setup
    n, beam = 301, 5          # 1505 rows; 2 chains: 151 + 150 sentences, 3 chains: 101 + 101 + 99
    emb = (torch.randn(n, ocfg.model_dim, generator=torch.Generator().manual_seed(99)) * 0.3).cuda()
    kw = dict(beam_size=beam, max_gen_len=(0, 30))
    from sonar_amd import _lib

    _lib.set_tuning(DEC_KS_OUT=2, DEC_FFN1_ENGINE=1)   # read by the engine at the start of every generate() call
    try:
        eng.set_chains(1)
        one = [t.cpu() for t in eng.generate(emb, [3, 702], **kw)]
        m_one = eng.last_margins(n).cpu()
        eng.set_chains(chains)
        got = [t.cpu() for t in eng.generate(emb, [3, 702], **kw)]
        m_got = eng.last_margins(n).cpu()
        again = [t.cpu() for t in eng.generate(emb, [3, 702], **kw)]   # chain workspaces are reused
        small_one, small = [], []
        for c, dst in ((1, small_one), (chains, small)):   # a small batch is never split (>= 384 rows per chain)
            eng.set_chains(c)
            dst.extend(t.cpu() for t in eng.generate(emb[:20], [3, 702], **kw))
        # the engine's own choice of tile shapes: only near-ties may differ (summation order of the split-K slabs)
        _lib.set_tuning(DEC_KS_OUT=None, DEC_FFN1_ENGINE=None)
        eng.set_chains(1)
        free_one = eng.generate(emb, [3, 702], **kw)[0].cpu()
        eng.set_chains(chains)
        free_got = eng.generate(emb, [3, 702], **kw)[0].cpu()
    finally:
        eng.set_chains(0)
        _lib.set_tuning(DEC_KS_OUT=None, DEC_FFN1_ENGINE=None)
    for a, b in zip(got, again):
        assert torch.equal(a, b)
    for a, b in zip(one, got):
        assert torch.equal(a, b)
    assert torch.equal(m_one, m_got)
    for a, b in zip(small, small_one):
        assert torch.equal(a, b)
    same = int((free_one[:, 0] == free_got[:, 0]).all(dim=1).sum())
    print(f"chains {chains}: bit-identical with pinned tile shapes; {same}/{n} best hypotheses identical with free ones")
    assert same >= n - max(3, n // 50)   # observed (r05, r06): 300 / 301; the difference is a near-tie under another summation order


def test_beam_logits_storage_type(setup):
    """smi_text_decoder_set_beam_logits_dtype (round 4): the beam search of an fp16 model stores its logits in fp16 (tile-major,
    statistics of the ROUNDED values), as the reference's fp16 final_proj produces them; fp32 keeps the accumulators.  Both
    re-score to the oracle's log-probabilities; their scores differ by at most the fp16 rounding of the logits; hypotheses are
    equal wherever the run's own decision margin exceeds that rounding."""
    OD, ocfg, params, eng = setup
    g = torch.Generator().manual_seed(23)
    n = 24
    emb = (torch.randn(n, ocfg.model_dim, generator=g) * 0.3).cuda()
    prompt = [3, 17]
    kw = dict(beam_size=5, min_gen_len=3, max_gen_len=(0, 14))
    out = {}
    try:
        for dt in (torch.float32, torch.float16):
            eng.set_beam_logits_dtype(dt)
            eng.set_slab_dtype(dt)
            toks, lens, scores = [t.cpu() for t in eng.generate(emb, prompt, **kw)]
            out[dt] = (toks, lens, scores, eng.last_margins(n).cpu())
    finally:
        eng.set_beam_logits_dtype(torch.float16)
        eng.set_slab_dtype(torch.float16)
    with pytest.raises(ValueError):
        eng.set_beam_logits_dtype(torch.bfloat16)
    t32, l32, s32, m32 = out[torch.float32]
    t16, l16, s16, m16 = out[torch.float16]
    logit_scale = OD.decoder_logits(params, ocfg, emb[:1].cpu(), torch.tensor([prompt])).abs().max().item()
    ulp = logit_scale * 2.0 ** -10          # fp16 rounding of a logit of that size, and of the normaliser's inputs
    same = 0
    for i in range(n):
        for dt, (toks, lens, scores, _) in out.items():
            L = int(lens[i, 0])
            total = _rescored(OD, params, ocfg, emb[i].cpu(), prompt, toks[i, 0, :L].tolist())
            assert abs(total / (len(prompt) + L - 1) - scores[i, 0].item()) <= 5e-3, (dt, i)
        if torch.equal(t32[i, 0], t16[i, 0]):
            same += 1
            assert abs(s32[i, 0].item() - s16[i, 0].item()) <= 4 * ulp
        else:
            assert min(m32[i, 0].item(), m16[i, 0].item()) <= 8 * ulp, (i, m32[i].tolist(), m16[i].tolist())
    assert same >= n - 3, same
    print(f"fp16 / fp32 beam logits: {same}/{n} best hypotheses identical, logit scale {logit_scale:.2f}")


def test_bf16_decoder_model_vs_oracle():
    """A bf16 decoder (`dtype=torch.bfloat16`, bf16 weights and bf16 sentence vectors; sonar/inference_pipelines/text.py:
    305-346 moves the model with `.to(device, dtype)`): bf16 weights are exact fp16 operands, the stream is fp32, the beam
    search keeps fp32 logits (fp16 storage is an fp16 model's, test_beam_logits_storage_type).  Teacher-forced logits and the
    beam-5 hypotheses against the fp32 oracle run on the SAME (bf16-representable) weights and inputs."""
    from oracle import text_decoder as OD
    from sonar_amd.text_decoder import ConditionalTransformerDecoderModel

    ocfg, cfg = _cfgs()
    params = {k: v.to(torch.bfloat16) for k, v in OD.make_synthetic_params(ocfg, seed=77, std=0.09).items()}
    fparams = {k: v.float() for k, v in params.items()}
    model = ConditionalTransformerDecoderModel(cfg, params, device="cuda:0", dtype=torch.bfloat16)
    eng = model.engine
    g = torch.Generator().manual_seed(5)
    emb = (torch.randn(6, ocfg.model_dim, generator=g) * 0.3).to(torch.bfloat16)
    prev = torch.randint(4, ocfg.vocab_size, (6, 9), generator=g)
    prev[:, 0] = 3
    ref = OD.decoder_logits(fparams, ocfg, emb.float(), prev)
    got = eng.logits(emb.cuda(), prev.cuda()).cpu()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 1.5e-2 * scale, ((got - ref).abs().max().item(), scale)
    prompt = [3, 17]
    toks, lens, scores = [t.cpu() for t in eng.generate(emb.cuda(), prompt, beam_size=5, min_gen_len=3, max_gen_len=(0, 12))]
    for i in range(6):
        L = int(lens[i, 0])
        total = _rescored(OD, fparams, ocfg, emb[i].float(), prompt, toks[i, 0, :L].tolist())
        assert abs(total / (len(prompt) + L - 1) - scores[i, 0].item()) <= 5e-3, (i, total, scores[i, 0].item())


def test_beam_search_forced_eos_and_min_len(setup):
    OD, ocfg, params, eng = setup
    emb = torch.randn(3, ocfg.model_dim, generator=torch.Generator().manual_seed(5)) * 0.3
    toks, lens, scores = eng.generate(emb.cuda(), [3, 701], beam_size=2, max_gen_len=(0, 4), min_gen_len=2)
    lens = lens.cpu()
    toks = toks.cpu()
    assert toks.shape[2] == 2 + 4
    for i in range(3):
        for j in range(2):
            L = int(lens[i, j])
            assert 3 <= L <= 4 and toks[i, j, L - 1].item() == 3   # >= min_gen_len tokens before EOS, <= cap
            assert 3 not in toks[i, j, :L - 1].tolist()


def test_embedding_to_text_pipeline(setup, tmp_path):
    import sentencepiece as spm

    from sonar_amd.inference_pipelines import EmbeddingToTextModelPipeline
    from sonar_amd.text_decoder import ConditionalTransformerDecoderModel
    from sonar_amd.tokenizer import NllbTokenizer

    OD, _, _, _ = setup
    words = ["hello", "world", "my", "name", "is", "paul", "teacher", "working", "bonjour", "monde"]
    corpus = tmp_path / "c.txt"
    g = torch.Generator().manual_seed(0)
    with open(corpus, "w") as fh:
        for _ in range(300):
            n = int(torch.randint(2, 10, (1,), generator=g))
            fh.write(" ".join(words[int(i)] for i in torch.randint(0, len(words), (n,), generator=g)) + "\n")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "toy"), vocab_size=40,
                                   model_type="unigram", hard_vocab_limit=False, bos_id=1, eos_id=2,
                                   unk_id=0, pad_id=-1, minloglevel=2)
    tok = NllbTokenizer(str(tmp_path / "toy.model"))
    ocfg, cfg = _cfgs(vocab=tok.vocab_info.size)
    params = OD.make_synthetic_params(ocfg, seed=77, std=0.09)
    model = ConditionalTransformerDecoderModel(cfg, params, device="cuda:0")
    pipe = EmbeddingToTextModelPipeline(model, tok, device=torch.device("cuda:0"))
    emb = torch.randn(6, ocfg.model_dim, generator=torch.Generator().manual_seed(9)) * 0.3
    texts = pipe.predict(emb, target_lang="fra_Latn", batch_size=4, max_gen_len=(0, 9))
    assert len(texts) == 6 and all(isinstance(t, str) for t in texts)
    prompt = tok.create_encoder(lang="fra_Latn", mode="target").prefix
    ref = OD.beam_search(params, ocfg, emb, prompt, beam_size=5, max_gen_len=(0, 9))
    same = sum(texts[i] == tok.decode(ref[i][0].seq) for i in range(6))
    assert same >= 5
    with pytest.raises(NotImplementedError):
        pipe.predict(emb, target_lang="fra_Latn", sampler=object())
    # a sampler switches predict() to the sampling generator (text.py:315-320); torch's global seed
    # makes it repeatable as it does for the reference
    from sonar_amd.generation import TopPSampler

    torch.manual_seed(4)
    s1 = pipe.predict(emb, target_lang="fra_Latn", batch_size=4, max_gen_len=(0, 9), sampler=TopPSampler(0.9))
    torch.manual_seed(4)
    s2 = pipe.predict(emb, target_lang="fra_Latn", batch_size=4, max_gen_len=(0, 9), sampler=TopPSampler(0.9))
    assert s1 == s2 and len(s1) == 6 and all(isinstance(t, str) for t in s1)


def test_text_to_text_and_speech_to_text_chain(setup, tmp_path):
    """TextToText / SpeechToText pipelines == decode(encode(.)) of the separate pipelines."""
    import sentencepiece as spm

    from oracle import speech_encoder as OS
    from oracle import text_encoder as OE
    from sonar_amd.inference_pipelines import (EmbeddingToTextModelPipeline, SpeechToTextModelPipeline,
                                               TextToEmbeddingModelPipeline, TextToTextModelPipeline)
    from sonar_amd.speech_encoder import SonarSpeechEncoderConfig, SonarSpeechEncoderModel
    from sonar_amd.text_decoder import ConditionalTransformerDecoderModel
    from sonar_amd.text_encoder import SonarTextEncoderConfig, SonarTextTransformerEncoderModel, VocabularyInfo
    from sonar_amd.tokenizer import NllbTokenizer

    OD, _, _, _ = setup
    words = ["hello", "world", "my", "name", "is", "paul", "teacher", "working", "bonjour", "monde"]
    corpus = tmp_path / "c.txt"
    g = torch.Generator().manual_seed(0)
    with open(corpus, "w") as fh:
        for _ in range(300):
            n = int(torch.randint(2, 10, (1,), generator=g))
            fh.write(" ".join(words[int(i)] for i in torch.randint(0, len(words), (n,), generator=g)) + "\n")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "toy"), vocab_size=40,
                                   model_type="unigram", hard_vocab_limit=False, bos_id=1, eos_id=2,
                                   unk_id=0, pad_id=-1, minloglevel=2)
    tok = NllbTokenizer(str(tmp_path / "toy.model"))
    v = tok.vocab_info.size
    ocfg, cfg = _cfgs(vocab=v)
    dec = ConditionalTransformerDecoderModel(cfg, OD.make_synthetic_params(ocfg, seed=77, std=0.09), device="cuda:0")
    oe = OE.OracleTextEncoderConfig(model_dim=256, num_layers=2, num_heads=4, ffn_inner_dim=512, vocab_size=v)
    ecfg = SonarTextEncoderConfig(model_dim=256, num_encoder_layers=2, num_encoder_attn_heads=4, ffn_inner_dim=512,
                                  vocab_info=VocabularyInfo(size=v), _from_fairseq=True)
    enc = SonarTextTransformerEncoderModel(ecfg, OE.make_synthetic_params(oe, seed=3, std=0.08), device="cuda:0",
                                           dtype=torch.float16)
    dev = torch.device("cuda:0")
    texts = ["hello world", "my name is paul", "bonjour monde"]
    t2t = TextToTextModelPipeline(enc, dec, tok, device=dev)
    got = t2t.predict(texts, source_lang="eng_Latn", target_lang="fra_Latn", batch_size=2, max_gen_len=(0, 7))
    emb = TextToEmbeddingModelPipeline(enc, tok, device=dev).predict(texts, source_lang="eng_Latn")
    want = EmbeddingToTextModelPipeline(dec, tok, device=dev).predict(emb, target_lang="fra_Latn", max_gen_len=(0, 7))
    assert got == want and len(got) == 3

    so = OS.OracleSpeechEncoderConfig(model_dim=256, num_layers=1, num_heads=4, ffn_inner_dim=512, conv_kernel=7,
                                      pooler_layers=1, pooler_heads=4, pooler_ffn_dim=384, pooler_vocab=64)
    scfg = SonarSpeechEncoderConfig(model_dim=256, num_encoder_layers=1, num_encoder_attn_heads=4, ffn_inner_dim=512,
                                    depthwise_conv_kernel_size=7, num_decoder_layers=1, num_decoder_attn_heads=4,
                                    decoder_ffn_inner_dim=384, max_frames=512)
    senc = SonarSpeechEncoderModel(scfg, OS.make_synthetic_params(so, seed=5, std=0.06), device="cuda:0")
    wavs = [torch.rand(1, 16000, generator=g) * 2 - 1, torch.rand(1, 20000, generator=g) * 2 - 1]
    s2t = SpeechToTextModelPipeline(senc, dec, tok, device=dev)
    out = s2t.predict(wavs, target_lang="eng_Latn", max_gen_len=(0, 6))
    assert len(out) == 2 and all(isinstance(t, str) for t in out)
