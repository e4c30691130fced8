"""GPU parity of the decoder BEYOND toy lengths (round 3; VERDICT r2 "decoder parity stops at 13 tokens").

`launch_dec_attention` (csrc/decoder.hip) picks, per step, `passes = pos / 64 + 1` online-softmax passes and
the chunk size NG = ceil((pos / 8 + 1) / passes) in 1..8; the beam-merged variant of round 3 adds a
per-chunk "all beams share this ancestor" fast path.  Every one of those code paths is held to the fp32 CPU
oracle here: teacher-forced logits over 17 / 65 / 130 / 511 positions (all NG, 1..8 passes, identity
ancestry), greedy runs of >= 70 and >= 140 forced steps checked DECISION BY DECISION on the engine's own
prefix (a single near-tie cannot cascade into a different suffix), and beam-5 runs of the same lengths whose
returned hypotheses are re-scored by the oracle -- a hypothesis' cumulative score was accumulated through the
ancestry table and the KV cache, so a wrong ancestor anywhere in 140 steps shows up as a score mismatch --
and compared with the oracle's own beam search under the measured-margin rule of tests/test_gpu_decoder.py.
Reference call path: sonar/inference_pipelines/text.py:305-346 (BeamSearchSeq2SeqGenerator)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS_REL = 1e-3          # of the logit range: what fp16 operands cannot separate (same constant as test_gpu_decoder.py)


def _cfgs(d=256, heads=4, ffn=512, layers=2, vocab=1000, max_seq_len=512):
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
    params = OD.make_synthetic_params(ocfg, seed=8642, std=0.09)
    eng = TextDecoderEngine(cfg, params, device="cuda:0")
    return OD, ocfg, params, eng


@pytest.mark.parametrize("t", [17, 65, 130, 511])
def test_teacher_forced_logits_long_prefix(setup, t):
    """positions 0..t-1: NG = 1..8 in one pass (t <= 64), then 2..8 passes with the online-softmax carry."""
    OD, ocfg, params, eng = setup
    g = torch.Generator().manual_seed(100 + t)
    n = 3
    emb = torch.randn(n, ocfg.model_dim, generator=g) * 0.3
    prev = torch.randint(4, ocfg.vocab_size, (n, t), generator=g)
    prev[:, 0] = 3
    ref = OD.decoder_logits(params, ocfg, emb, prev)
    got = eng.logits(emb.cuda(), prev.cuda()).cpu()
    assert got.shape == ref.shape == (n, t, ocfg.vocab_size) and torch.isfinite(got).all()
    scale = ref.abs().max().item()
    err = (got - ref).abs().amax(dim=(0, 2))            # per position
    print(f"t={t}: max |logit diff| / scale = {err.max().item() / scale:.2e} (worst position {int(err.argmax())}), "
          f"last position {err[-1].item() / scale:.2e}")
    assert err.max().item() <= 1.5e-2 * scale, (err.max().item(), scale, int(err.argmax()))
    agree = (got.argmax(-1) == ref.argmax(-1)).float().mean().item()
    assert agree >= 0.97, agree


def _limits(plen, min_gen_len, max_gen_len, model_max):
    max_len = min(plen + max_gen_len[1], model_max)
    return max_len, min(plen + min_gen_len, max_len)


def _check_decisions(OD, params, ocfg, e, prompt, seq, max_len, min_len, eps):
    """Every token of `seq` (generated part, with the final EOS) must be the oracle's arg-max on the SAME
    prefix under fairseq2's masks, or lose to it by less than eps.  Returns (near ties used, summed log-prob)."""
    plen = len(prompt)
    full = torch.tensor([list(prompt) + seq])
    lp = torch.log_softmax(OD.decoder_logits(params, ocfg, e.unsqueeze(0), full[:, :-1]), dim=-1, dtype=torch.float32)[0]
    total = lp[torch.arange(full.shape[1] - 1), full[0, 1:]].sum().item()
    ties = 0
    for t in range(plen - 1, full.shape[1] - 1):
        step_nr = t + 1                                   # index of the token chosen at this step
        chosen = int(full[0, t + 1])
        if step_nr == max_len - 1:                        # forced EOS at the cap: no decision
            assert chosen == 3
            continue
        row = lp[t].clone()
        row[0] = -torch.inf                               # PAD never
        if step_nr < min_len:
            row[3] = -torch.inf                           # EOS blocked below the minimum length
        best = int(row.argmax())
        if chosen != best:
            gap = (row[best] - row[chosen]).item()
            assert gap < eps, f"step {step_nr}: token {chosen} chosen, oracle prefers {best} by {gap:.3e} >= eps {eps:.3e}"
            ties += 1
    return ties, total


def _logit_range(OD, params, ocfg, emb, prompt):
    lg = OD.decoder_logits(params, ocfg, emb, torch.tensor([list(prompt)] * emb.shape[0]))
    return (lg.max() - lg.min()).item()


@pytest.mark.parametrize("forced", [70, 141])
def test_greedy_long_runs_decision_by_decision(setup, forced):
    OD, ocfg, params, eng = setup
    g = torch.Generator().manual_seed(300 + forced)
    n = 8
    emb = torch.randn(n, ocfg.model_dim, generator=g) * 0.3
    prompt = [3, 700]
    kw = dict(min_gen_len=forced, max_gen_len=(0, forced + 12))
    toks, lens, scores = eng.generate(emb.cuda(), prompt, beam_size=1, **kw)
    toks, lens, scores = toks.cpu(), lens.cpu(), scores.cpu()
    max_len, min_len = _limits(len(prompt), forced, kw["max_gen_len"], ocfg.max_seq_len)
    eps = EPS_REL * _logit_range(OD, params, ocfg, emb, prompt)
    ties_total = 0
    for i in range(n):
        L = int(lens[i, 0])
        seq = toks[i, 0, :L].tolist()
        assert forced + 1 <= L <= forced + 12 and seq[-1] == 3 and 3 not in seq[:-1] and 0 not in seq
        ties, total = _check_decisions(OD, params, ocfg, emb[i], prompt, seq, max_len, min_len, eps)
        ties_total += ties
        assert abs(total / (len(prompt) + L - 1) - scores[i, 0].item()) <= 5e-3
    decisions = int(lens[:, 0].sum())
    print(f"greedy, {forced} forced steps: {decisions - ties_total}/{decisions} decisions equal the oracle's arg-max on the "
          f"same prefix; {ties_total} within eps {eps:.2e}")
    assert ties_total <= max(2, decisions // 100)


@pytest.mark.parametrize("forced", [70, 141])
def test_beam5_long_runs_rescored_and_vs_oracle_search(setup, forced):
    OD, ocfg, params, eng = setup
    g = torch.Generator().manual_seed(500 + forced)
    n, beam = 6, 5
    emb = torch.randn(n, ocfg.model_dim, generator=g) * 0.3
    prompt = [3, 701]
    kw = dict(min_gen_len=forced, max_gen_len=(0, forced + 10))
    toks, lens, scores = eng.generate(emb.cuda(), prompt, beam_size=beam, **kw)
    margins = eng.last_margins(n).cpu()
    toks, lens, scores = toks.cpu(), lens.cpu(), scores.cpu()
    from tests.neartie import check_engine_margin, oracle_excuses

    om = []
    ref = OD.beam_search_incremental(params, ocfg, emb, prompt, beam_size=beam, margins_out=om, **kw)
    eps = EPS_REL * _logit_range(OD, params, ocfg, emb, prompt)
    same, excused = 0, []
    for i in range(n):
        assert int((lens[i] > 0).sum()) == beam
        for j in range(beam):                              # EVERY returned hypothesis carries the oracle's score for it
            L = int(lens[i, j])
            seq = toks[i, j, :L].tolist()
            assert L >= forced + 1 and seq[-1] == 3 and 3 not in seq[:-1] and 0 not in seq
            full = torch.tensor([prompt + seq])
            lp = torch.log_softmax(OD.decoder_logits(params, ocfg, emb[i:i + 1], full[:, :-1]), dim=-1, dtype=torch.float32)
            total = lp[0, torch.arange(full.shape[1] - 1), full[0, 1:]].sum().item()
            assert abs(total / (len(prompt) + L - 1) - scores[i, j].item()) <= 5e-3, (i, j, total, scores[i, j].item())
        assert all(scores[i, j] >= scores[i, j + 1] - 1e-6 for j in range(beam - 1))
        best = toks[i, 0, : int(lens[i, 0])].tolist()
        check_engine_margin(margins[i], om[i], eps, f"{forced} forced steps, sentence {i}")
        if best == ref[i][0].seq.tolist():
            same += 1
            continue
        step_gap, final_gap = om[i][:2]
        assert oracle_excuses(om[i], eps), (
            f"sentence {i}: best hypothesis differs from the oracle's although every decision margin the ORACLE "
            f"measured (step {step_gap:.3e}, final {final_gap:.3e}) is above eps {eps:.3e}; engine's: {margins[i].tolist()}")
        # a measured near tie somewhere in > 700 candidate rankings: the returned hypothesis must still be as
        # good as the oracle's best
        assert abs(scores[i, 0].item() - ref[i][0].score) <= 2 * eps, (scores[i, 0].item(), ref[i][0].score)
        excused.append((i, step_gap, final_gap))
    print(f"beam 5, {forced} forced steps: {same}/{n} best hypotheses token-identical to the oracle's beam search; "
          f"measured near-ties (< {eps:.2e}): {excused}")
    assert same >= n // 2, (same, excused)
