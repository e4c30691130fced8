"""GPU parity of the sampling generator (csrc/sampling.hip, smi_text_decoder_sample) against the CPU
restatement in oracle/text_decoder.py: the kept set and the draw of one step on given logits, and the
generation loop's invariants.  (A sampled SEQUENCE cannot be compared token for token: the fp16 engine's
probabilities differ from the fp32 oracle's by ~1e-2, which moves span boundaries under the same word.)"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sample_rows(logits, sampler, z, temperature=1.0, pad=0, eos=3, block_eos=False, unk=1, unk_penalty=0.0):
    from sonar_amd import _lib

    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    rows, v = logits.shape
    ld = (v + 255) // 256 * 256
    buf = torch.full((rows, ld), 7.5, dtype=torch.float32, device="cuda")   # padding columns hold junk
    buf[:, :v] = logits.cuda()
    zt = torch.tensor([w - (1 << 64) if w >= (1 << 63) else w for w in z], dtype=torch.int64, device="cuda")
    tok = torch.empty(rows, dtype=torch.int32, device="cuda")
    lp = torch.empty(rows, dtype=torch.float32, device="cuda")
    mass = torch.empty(rows, dtype=torch.int64, device="cuda")
    cnt = torch.empty(rows, dtype=torch.int32, device="cuda")
    kind = _lib.SMI_SAMPLER_TOP_K if sampler[0] == "top_k" else _lib.SMI_SAMPLER_TOP_P
    _lib.check(lib.smi_sample_rows(buf.data_ptr(), ld, rows, v, kind, int(sampler[1]) if kind == 0 else 1,
                                   float(sampler[1]) if kind == 1 else 1.0, temperature, pad, eos, int(block_eos),
                                   unk, unk_penalty, zt.data_ptr(), tok.data_ptr(), lp.data_ptr(), mass.data_ptr(), cnt.data_ptr(),
                                   _lib.current_stream_ptr()))
    torch.cuda.synchronize()
    return tok.cpu().tolist(), lp.cpu(), mass.cpu().tolist(), cnt.cpu().tolist()


def _check_against_oracle(logits, sampler, seed, temperature=1.0, block_eos=False, exact_count=False):
    from oracle import text_decoder as OD

    rows = logits.shape[0]
    z = [OD.splitmix_word(seed, r, 5) for r in range(rows)]
    tok, lp, mass, cnt = _sample_rows(logits, sampler, z, temperature, block_eos=block_eos)
    checked = 0
    for r in range(rows):
        keep = OD.sample_filter(logits[r], sampler, temperature, block_eos=block_eos).numpy()
        masses, _ = OD.q40_masses(logits[r], temperature)
        probs = OD.sampling_probs(logits[r], temperature, block_eos=block_eos)
        assert keep[tok[r]] or not exact_count, (r, tok[r])
        assert tok[r] != 0 and not (block_eos and tok[r] == 3)
        assert lp[r].item() == pytest.approx(float(torch.log(probs[tok[r]])), abs=2e-4)
        if sampler[0] == "top_p":
            # the nucleus edge: the oracle's fp32 cumsum and the engine's integer sums may disagree on
            # tokens whose exclusive mass sits within rounding of p
            sp = torch.sort(probs, descending=True).values.double()
            excl = torch.cumsum(sp, 0) - sp
            npos = int((sp > 0).sum())
            lo = int((excl[:npos] <= sampler[1] - 2e-6).sum())
            hi = int((excl[:npos] <= sampler[1] + 2e-6).sum())
            assert lo <= cnt[r] <= hi, (r, lo, cnt[r], hi)
            # the draw is then checked on the engine's own nucleus: its cnt[r] best-ranked tokens
            keep = OD.sample_filter(logits[r], ("top_k", cnt[r]), temperature, block_eos=block_eos).numpy()
        assert cnt[r] == int(keep.sum()), (r, cnt[r], int(keep.sum()))
        want_mass = int(masses[keep].astype(object).sum())
        assert abs(mass[r] - want_mass) <= 4e-6 * want_mass + 64     # v_exp_f32 vs numpy exp2 per token
        want_tok, margin = OD.sample_draw(masses, keep, z[r])
        if margin > 2e-6:   # clear of the span boundaries by 10x the per-token exp rounding
            assert tok[r] == want_tok, (r, tok[r], want_tok)
            checked += 1
    assert checked >= rows // 2


@pytest.mark.parametrize("sampler", [("top_k", 1), ("top_k", 5), ("top_k", 50), ("top_k", 3000), ("top_p", 0.3),
                                     ("top_p", 0.9), ("top_p", 0.999)])
def test_step_kept_set_and_draw_vs_oracle(sampler):
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(24, 5003, generator=g) * 3.0
    _check_against_oracle(logits, sampler, seed=17)
    _check_against_oracle(logits, sampler, seed=18, temperature=0.7, block_eos=True)


def test_step_full_nllb_vocabulary():
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(6, 256206, generator=g) * 2.5
    logits[:, 0] += 20.0                       # pad would dominate: it must never be drawn
    _check_against_oracle(logits, ("top_p", 0.9), seed=5)
    _check_against_oracle(logits, ("top_k", 40), seed=6)


def test_value_ties_at_the_threshold_keep_the_lowest_ids():
    g = torch.Generator().manual_seed(8)
    logits = (torch.randn(16, 4099, generator=g) * 2.0).round() * 0.5      # ~20 distinct values: ties everywhere
    for sampler in (("top_k", 7), ("top_k", 300), ("top_p", 0.5), ("top_p", 0.95)):
        _check_against_oracle(logits, sampler, seed=21, exact_count=sampler[0] == "top_k")
    # a flat row: top-k keeps the k lowest unmasked ids, each with the same chance
    flat = torch.zeros(2000, 64)
    z = [(i * 0x9E3779B97F4A7C15) & ((1 << 64) - 1) for i in range(2000)]
    tok, lp, mass, cnt = _sample_rows(flat, ("top_k", 4), z)
    assert set(cnt) == {4} and set(tok) == {1, 2, 3, 4}                   # id 0 is pad
    assert np.abs(np.bincount(tok, minlength=5)[1:] / 2000 - 0.25).max() < 0.05
    assert lp.tolist() == pytest.approx([-np.log(64)] * 2000, abs=1e-5)
    tok, _, _, cnt = _sample_rows(flat[:50], ("top_p", 0.5), z[:50])
    # exclusive mass i/64 <= 0.5 over the 63 unmasked tokens in id order (pad's 1/64 stays in the normaliser)
    assert set(cnt) == {33} and max(tok) <= 33 and min(tok) >= 1


def test_unk_penalty_lowers_the_unk_probability():
    """probs[unk] -= unk_penalty before the filter (fairseq2's sampling generator; fs2-recall): the kept set, its
    integer mass, the draw and the step score against the oracle, incl. a penalty that removes the token."""
    from oracle import text_decoder as OD

    g = torch.Generator().manual_seed(31)
    logits = torch.randn(16, 3001, generator=g) * 2.0
    logits[:, 1] = logits.max(dim=-1).values + 1.0          # UNK is the most probable token of every row
    z = [OD.splitmix_word(7, r, 2) for r in range(16)]
    p_unk = torch.softmax(logits, -1)[:, 1]
    for sampler in (("top_k", 4), ("top_p", 0.7)):
        for pen in (0.05, float(p_unk.min()) * 0.5, 2.0, -0.1):
            tok, lp, mass, cnt = _sample_rows(logits, sampler, z, unk_penalty=pen)
            checked = 0
            for r in range(16):
                probs = OD.sampling_probs(logits[r], unk_penalty=pen)
                masses, _ = OD.q40_masses(logits[r], unk_penalty=pen)
                keep = OD.sample_filter(logits[r], sampler, unk_penalty=pen).numpy()
                if sampler[0] == "top_p":          # the nucleus edge: judge the draw on the engine's own count
                    keep = OD.sample_filter(logits[r], ("top_k", cnt[r]), unk_penalty=pen).numpy()
                if pen >= 1.0:
                    assert tok[r] != 1 and not keep[1]
                assert cnt[r] == int(keep.sum()), (sampler, pen, r, cnt[r], int(keep.sum()))
                want_mass = int(masses[keep].astype(object).sum())
                assert abs(mass[r] - want_mass) <= 4e-6 * want_mass + 64
                assert lp[r].item() == pytest.approx(float(torch.log(probs[tok[r]])), abs=3e-4)
                want_tok, margin = OD.sample_draw(masses, keep, z[r])
                if margin > 2e-6:
                    assert tok[r] == want_tok, (sampler, pen, r, tok[r], want_tok)
                    checked += 1
            assert checked >= 8
    # the penalty moves the draw: with top-1 the UNK (most probable) is drawn without it and cannot be once it is removed
    tok0, *_ = _sample_rows(logits, ("top_k", 1), z)
    tok1, *_ = _sample_rows(logits, ("top_k", 1), z, unk_penalty=2.0)
    assert set(tok0) == {1} and 1 not in tok1


def test_draw_frequencies_follow_the_kept_probabilities():
    from oracle import text_decoder as OD

    g = torch.Generator().manual_seed(12)
    row = torch.randn(1, 700, generator=g) * 2.0
    n = 6000
    z = [OD.splitmix_word(99, r, 1) for r in range(n)]
    tok, _, _, cnt = _sample_rows(row.expand(n, -1).contiguous(), ("top_p", 0.8), z)
    keep = OD.sample_filter(row[0], ("top_p", 0.8))
    probs = OD.sampling_probs(row[0]) * keep
    probs = (probs / probs.sum()).numpy()
    freq = np.bincount(tok, minlength=700) / n
    assert freq[~keep.numpy()].sum() == 0
    assert np.abs(freq - probs).max() < 4 * np.sqrt(probs.max() / n) + 1e-3


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


def test_top_k_1_sampling_equals_beam_1(setup):
    from sonar_amd.generation import TopKSampler

    OD, ocfg, params, eng = setup
    emb = torch.randn(9, ocfg.model_dim, generator=torch.Generator().manual_seed(31)) * 0.3
    st, sl, ss = eng.sample(emb.cuda(), [3, 701], TopKSampler(1), seed=5, max_gen_len=(0, 12))
    # the sampling generator works on fp32 logits; the beam search of an fp16 model stores fp16 logits (round 4,
    # smi_text_decoder_set_beam_logits_dtype): same tokens, scores equal up to that rounding (2^-11 of a logit of ~8 per token);
    # with fp32 storage the two paths agree to fp32 noise
    try:
        for dt, tol in ((torch.float32, 1e-4), (torch.float16, 2e-3)):
            eng.set_beam_logits_dtype(dt)
            eng.set_slab_dtype(dt)          # its own setting since round 5 (the sampling generator keeps fp32 partial sums)
            bt, bl, bs = eng.generate(emb.cuda(), [3, 701], beam_size=1, max_gen_len=(0, 12))
            assert torch.equal(bl[:, 0].cpu(), sl.cpu())
            for i in range(9):
                L = int(sl[i])
                assert torch.equal(bt[i, 0, :L].cpu(), st[i, :L].cpu())
                assert (st[i, L:] == -1).all()
            assert torch.allclose(bs[:, 0].cpu(), ss.cpu(), atol=tol, rtol=tol), (dt, (bs[:, 0].cpu() - ss.cpu()).abs().max())
    finally:
        eng.set_beam_logits_dtype(torch.float16)
        eng.set_slab_dtype(torch.float16)


def test_sampled_sequences_scores_and_reproducibility(setup):
    from sonar_amd.generation import TopPSampler

    OD, ocfg, params, eng = setup
    emb = torch.randn(7, ocfg.model_dim, generator=torch.Generator().manual_seed(32)) * 0.3
    prompt = [3, 555]
    kw = dict(min_gen_len=3, max_gen_len=(0, 10), temperature=1.3)
    t1, l1, s1 = eng.sample(emb.cuda(), prompt, TopPSampler(0.95), seed=77, **kw)
    t2, l2, s2 = eng.sample(emb.cuda(), prompt, TopPSampler(0.95), seed=77, **kw)
    t3, l3, s3 = eng.sample(emb.cuda(), prompt, TopPSampler(0.95), seed=78, **kw)
    assert torch.equal(t1, t2) and torch.equal(l1, l2) and torch.equal(s1, s2)
    assert not torch.equal(t1, t3)
    # a sentence's random stream does not depend on the batch it is decoded in
    t4, l4, _ = eng.sample(emb[2:5].cuda(), prompt, TopPSampler(0.95), seed=77, sentence_offset=2, **kw)
    assert torch.equal(t4.cpu()[:, : t1.shape[1]], t1.cpu()[2:5]) and torch.equal(l4.cpu(), l1.cpu()[2:5])
    t1, l1, s1 = t1.cpu(), l1.cpu(), s1.cpu()
    for i in range(7):
        L = int(l1[i])
        seq = t1[i, :L].tolist()
        assert 4 <= L <= 10 and seq[-1] == 3 and 3 not in seq[:-1] and 0 not in seq and min(seq) >= 0
        full = torch.tensor([prompt + seq])
        lp = torch.log_softmax(OD.decoder_logits(params, ocfg, emb[i:i + 1], full[:, :-1]) / 1.3, dim=-1)
        ref = lp[0, torch.arange(full.shape[1] - 1), full[0, 1:]].sum().item() / (full.shape[1] - 1)
        assert s1[i].item() == pytest.approx(ref, abs=3e-2)


def test_first_sampled_token_follows_the_model_distribution(setup):
    from sonar_amd.generation import TopKSampler

    OD, ocfg, params, eng = setup
    e = torch.randn(1, ocfg.model_dim, generator=torch.Generator().manual_seed(33)) * 0.3
    n = 4096
    toks, lens, _ = eng.sample(e.expand(n, -1).contiguous().cuda(), [3, 444], TopKSampler(6), seed=3,
                               max_gen_len=(0, 2), temperature=2.0)
    first = toks[:, 0].cpu().numpy()
    logits = OD.decoder_logits(params, ocfg, e, torch.tensor([[3, 444]]))[0, -1]
    keep = OD.sample_filter(logits, ("top_k", 6), temperature=2.0, block_eos=True)
    probs = OD.sampling_probs(logits, 2.0, block_eos=True) * keep
    probs = (probs / probs.sum()).numpy()
    freq = np.bincount(first, minlength=ocfg.vocab_size) / n
    assert freq[~keep.numpy()].sum() == 0
    assert np.abs(freq - probs).max() < 0.04


def test_sampler_objects():
    from sonar_amd import _lib
    from sonar_amd.generation import TopKSampler, TopPSampler, resolve_sampler

    with pytest.raises(ValueError):
        TopKSampler(0)
    with pytest.raises(ValueError):
        TopPSampler(1.5)
    with pytest.raises(NotImplementedError):
        resolve_sampler(object())
    assert resolve_sampler(TopPSampler()) == (_lib.SMI_SAMPLER_TOP_P, 1, pytest.approx(0.9))
    assert resolve_sampler(TopKSampler(7))[:2] == (_lib.SMI_SAMPLER_TOP_K, 7)


def test_sampling_argument_errors(setup):
    from sonar_amd.generation import TopKSampler

    OD, ocfg, params, eng = setup
    emb = torch.zeros(2, ocfg.model_dim).cuda()
    # an UNK penalty that removes the token: generation runs and never emits UNK (id 1)
    toks, lens, _ = eng.sample(emb, [3, 5], TopKSampler(50), unk_penalty=1.0, max_gen_len=(0, 12), seed=4)
    assert int((toks == 1).sum()) == 0 and int(lens.min()) >= 1
    with pytest.raises(ValueError):
        eng.sample(emb, [3, 5], TopKSampler(2), max_seq_len=10 ** 6)
    with pytest.raises(RuntimeError):
        eng.sample(emb, [3, 5], TopKSampler(2), temperature=0.0)
