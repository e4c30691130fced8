"""CPU: pin the oracle DECODER against the HuggingFace M2M100Decoder twin fixture and check
the beam-search restatement's self-consistency."""
import os

import torch
from torch.testing import assert_close

from oracle import text_decoder as OD

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "m2m100_decoder_twin.pt")


def _from_fixture():
    from sonar_amd.text_decoder import convert_sonar_text_decoder_checkpoint

    fx = torch.load(GOLDEN, weights_only=False)
    c = fx["config"]
    cfg = OD.OracleTextDecoderConfig(model_dim=c["model_dim"], num_layers=c["num_layers"], num_heads=c["num_heads"],
                                     ffn_inner_dim=c["ffn_inner_dim"], vocab_size=c["vocab_size"],
                                     max_seq_len=c["max_seq_len"])
    params = convert_sonar_text_decoder_checkpoint(fx["checkpoint"])
    assert set(OD.param_names(cfg)) == set(params.keys())
    return fx, cfg, params


def test_decoder_logits_match_hf_twin():
    fx, cfg, params = _from_fixture()
    logits = OD.decoder_logits(params, cfg, fx["embeddings"], fx["prev_tokens"])
    assert_close(logits, fx["logits"], atol=3e-4, rtol=1e-5)


def test_decoder_is_causal_and_incremental():
    fx, cfg, params = _from_fixture()
    full = OD.decoder_logits(params, cfg, fx["embeddings"], fx["prev_tokens"])
    part = OD.decoder_logits(params, cfg, fx["embeddings"], fx["prev_tokens"][:, :4])
    assert_close(full[:, :4], part, atol=1e-5, rtol=1e-5)


def test_cross_attention_collapses_to_a_per_sentence_constant():
    # one key => softmax == 1 => cross_attn(x) = W_o (W_v e + b_v) + b_o for every query
    fx, cfg, params = _from_fixture()
    p = "decoder.layers.0.encoder_decoder_attn."
    e = fx["embeddings"]
    const = torch.nn.functional.linear(
        torch.nn.functional.linear(e, params[p + "v_proj.weight"], params[p + "v_proj.bias"]),
        params[p + "output_proj.weight"], params[p + "output_proj.bias"])
    q = torch.randn(e.shape[0], 5, cfg.model_dim)
    got = OD._mha(params, p, q, e.unsqueeze(1), cfg.num_heads, causal=False)
    assert_close(got, const.unsqueeze(1).expand_as(got), atol=1e-5, rtol=1e-5)


def test_beam1_equals_greedy_and_scores_are_logprobs():
    cfg = OD.OracleTextDecoderConfig(model_dim=64, num_layers=2, num_heads=4, ffn_inner_dim=128, vocab_size=120,
                                     max_seq_len=64)
    params = OD.make_synthetic_params(cfg, seed=5, std=0.25)
    emb = torch.randn(3, 64, generator=torch.Generator().manual_seed(1))
    prompt = [3, 57]
    hyps = OD.beam_search(params, cfg, emb, prompt, beam_size=1, max_gen_len=(0, 11))
    greedy = OD.greedy_decode(params, cfg, emb, prompt, max_new=11)
    for h, g, e in zip(hyps, greedy, emb):
        m = len(h[0].seq) - 1  # the last beam token may be the forced EOS at max length
        assert h[0].seq.tolist()[:m] == g[:m]
        # step scores are the teacher-forced log-probs of the produced tokens
        seq = torch.tensor([prompt + h[0].seq.tolist()])
        lp = torch.log_softmax(OD.decoder_logits(params, cfg, e.unsqueeze(0), seq[:, :-1]), dim=-1)
        ref = lp[0, torch.arange(len(prompt) - 1, seq.shape[1] - 1), seq[0, len(prompt):]]
        if h[0].seq[-1].item() == 3 and len(h[0].seq) < 11:
            assert_close(h[0].step_scores, ref, atol=1e-4, rtol=1e-4)


def test_beam_search_properties():
    cfg = OD.OracleTextDecoderConfig(model_dim=64, num_layers=2, num_heads=4, ffn_inner_dim=128, vocab_size=120,
                                     max_seq_len=64)
    params = OD.make_synthetic_params(cfg, seed=6, std=0.3)
    emb = torch.randn(2, 64, generator=torch.Generator().manual_seed(2))
    hyps = OD.beam_search(params, cfg, emb, [3, 60], beam_size=4, max_gen_len=(0, 7))
    for hs in hyps:
        assert len(hs) == 4
        assert all(h.seq[-1].item() == 3 for h in hs)                    # every hypothesis ends with EOS
        assert all(len(h.seq) >= 2 for h in hs)                           # min_gen_len=1: EOS not first
        assert all(len(h.seq) <= 1 + 6 + 0 for h in hs)                   # source_len*1 + 6 generated tokens max
        assert [h.score for h in hs] == sorted((h.score for h in hs), reverse=True)
        assert all(0 not in h.seq.tolist() for h in hs)                   # PAD never generated


GREEDY = os.path.join(os.path.dirname(__file__), "golden", "m2m100_greedy_twin.pt")


def test_greedy_generation_matches_hf_generate():
    """The generation loop (prompt forcing, incremental steps, EOS stop, min-length EOS
    suppression) against HuggingFace generate(num_beams=1) on a tied twin
    (tests/golden/make_golden_generate.py).  At the length cap the reference forces EOS
    (fairseq2 beam search) where HF just stops: that last token is excluded."""
    from sonar_amd.text_decoder import convert_sonar_text_decoder_checkpoint

    fx = torch.load(GREEDY, weights_only=False)
    c = fx["config"]
    cfg = OD.OracleTextDecoderConfig(model_dim=c["model_dim"], num_layers=c["num_layers"], num_heads=c["num_heads"],
                                     ffn_inner_dim=c["ffn_inner_dim"], vocab_size=c["vocab_size"],
                                     max_seq_len=c["max_seq_len"])
    params = convert_sonar_text_decoder_checkpoint(fx["checkpoint"])
    stopped = capped = 0
    for run in fx["runs"]:
        hyps = OD.beam_search(params, cfg, fx["embeddings"], run["prompt"], beam_size=1,
                              min_gen_len=run["min_gen_len"], max_gen_len=(0, run["max_new"]))
        greedy = OD.greedy_decode(params, cfg, fx["embeddings"], run["prompt"], max_new=run["max_new"])
        for h, g, want in zip(hyps, greedy, run["generated"]):
            got = h[0].seq.tolist()                       # generated part only
            if want[-1] == 3:
                assert got == want
                assert len(want) > run["min_gen_len"]
                stopped += 1
            else:
                assert len(want) == run["max_new"] and len(got) == run["max_new"]
                assert got[:-1] == want[:-1] and got[-1] == 3
                capped += 1
            if run["min_gen_len"] == 1:
                m = len(want) - 1
                assert g[:m] == want[:m]
    assert stopped >= 10 and capped >= 10     # the fixture exercises both exits


def test_incremental_beam_search_equals_the_quadratic_restatement():
    """oracle.beam_search_incremental (K/V cache + index_select re-ordering, the reference's evaluation order; the
    variant bench.py times as the C5 CPU baseline) returns the hypotheses of oracle.beam_search."""
    import torch

    from oracle import text_decoder as OD

    cfg = OD.OracleTextDecoderConfig(model_dim=64, num_layers=2, num_heads=4, ffn_inner_dim=96, vocab_size=200, max_seq_len=40)
    params = OD.make_synthetic_params(cfg, seed=5, std=0.12)
    emb = torch.randn(4, 64, generator=torch.Generator().manual_seed(1)) * 0.4
    for beam, kw in ((1, dict(max_gen_len=(0, 12))), (3, dict(max_gen_len=(0, 17), min_gen_len=9)), (5, dict(max_gen_len=(0, 9)))):
        a = OD.beam_search(params, cfg, emb, [3, 150], beam_size=beam, **kw)
        b = OD.beam_search_incremental(params, cfg, emb, [3, 150], beam_size=beam, **kw)
        for ha, hb in zip(a, b):
            assert len(ha) == len(hb) == beam
            for x, y in zip(ha, hb):
                assert x.seq.tolist() == y.seq.tolist()
                assert abs(x.score - y.score) <= 1e-5
                assert torch.allclose(x.step_scores, y.step_scores, atol=1e-5)
