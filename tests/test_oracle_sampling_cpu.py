"""CPU: the sampling restatement (fairseq2 TopKSampler / TopPSampler / SamplingSeq2SeqGenerator as the
reference builds them, sonar/inference_pipelines/text.py:315-320) -- known answers of the filters, the
integer draw, and the generator's self-consistency.  PARITY UNPINNED against fairseq2 itself (absent)."""
import numpy as np
import pytest
import torch

from oracle import text_decoder as OD


def test_top_p_keeps_the_sorted_prefix_whose_exclusive_cumsum_stays_within_p():
    probs = torch.tensor([[0.05, 0.5, 0.125, 0.25, 0.075]])       # sorted: .5 .25 .125 .075 .05
    assert OD.top_p_mask(probs, 0.5).tolist() == [[False, True, False, True, False]]    # exclusive 0, .5
    assert OD.top_p_mask(probs, 0.49).tolist() == [[False, True, False, False, False]]
    assert OD.top_p_mask(probs, 0.875).tolist() == [[False, True, True, True, True]]    # .875 <= .875 keeps .075
    assert OD.top_p_mask(probs, 1.0).all()
    assert OD.top_p_mask(probs, 1e-6).sum() == 1                   # the most probable token always survives


def test_top_k_and_ties_go_to_the_lower_token_id():
    probs = torch.tensor([[0.1, 0.3, 0.3, 0.2, 0.1]])
    assert OD.top_k_mask(probs, 1).tolist() == [[False, True, False, False, False]]
    assert OD.top_k_mask(probs, 2).tolist() == [[False, True, True, False, False]]
    assert OD.top_k_mask(probs, 4).tolist() == [[True, True, True, True, False]]
    assert OD.top_k_mask(probs, 99).all()


def test_filter_never_keeps_masked_tokens():
    lg = torch.zeros(16)
    lg[0] = 9.0    # pad would be the most probable token
    lg[3] = 8.0    # then EOS
    keep = OD.sample_filter(lg, ("top_k", 3), pad_idx=0, eos_idx=3, block_eos=True)
    assert keep.sum() == 3 and not keep[0] and not keep[3]
    keep = OD.sample_filter(lg, ("top_k", 3), pad_idx=0, eos_idx=3, block_eos=False)
    assert keep[3] and not keep[0]
    # the probabilities are not renormalised after masking: pad keeps its 0.73, EOS has 0.27
    keep = OD.sample_filter(lg, ("top_p", 0.2), pad_idx=0, eos_idx=3, block_eos=False)
    assert keep.tolist() == [False, False, False, True] + [False] * 12
    keep = OD.sample_filter(lg, ("top_p", 0.5), pad_idx=0, eos_idx=3, block_eos=False)
    assert keep.sum() == 15 and not keep[0]                      # the rest never adds up to p


def test_random_word_and_draw_known_answers():
    assert OD.splitmix_word(1, 0, 0) == 0x910A2DEC89025CC1
    assert OD.splitmix_word(1, 0, 1) != OD.splitmix_word(1, 1, 0)
    masses = np.array([4, 0, 6, 10, 0, 20], dtype=np.uint64)
    keep = np.array([True, False, True, True, False, True])
    # V = 6 -> two 4-token groups owned by threads 0 and 1: the walk is in id order here
    for z, want in ((0, 0), ((4 << 64) // 40, 0), ((4 << 64) // 40 + 1, 2), ((10 << 64) // 40 + 1, 3), ((1 << 64) - 1, 5)):
        assert OD.sample_draw(masses, keep, z)[0] == want
    # a token's share of the draws is its share of the kept mass
    hits = np.zeros(6)
    for i in range(4000):
        hits[OD.sample_draw(masses, keep, OD.splitmix_word(7, i, 0))[0]] += 1
    assert np.abs(hits / 4000 - masses * keep / 40).max() < 0.03


def test_draw_order_is_thread_major_over_four_token_groups():
    v = 4 * OD.SAMPLE_THREADS * 2 + 8
    masses = np.ones(v, dtype=np.uint64)
    keep = np.ones(v, dtype=bool)
    # thread 0 owns groups 0, 1024, 2048: the 5th unit of mass is the first token of group 1024
    assert OD.sample_draw(masses, keep, (4 << 64) // v + 1)[0] == 4 * OD.SAMPLE_THREADS
    assert OD.sample_draw(masses, keep, 0)[0] == 0


@pytest.fixture(scope="module")
def toy():
    cfg = OD.OracleTextDecoderConfig(model_dim=64, num_layers=2, num_heads=4, ffn_inner_dim=128, vocab_size=120,
                                     max_seq_len=64)
    params = OD.make_synthetic_params(cfg, seed=5, std=0.25)
    emb = torch.randn(3, 64, generator=torch.Generator().manual_seed(1))
    return cfg, params, emb


def test_top_k_1_sampling_is_greedy_decoding(toy):
    cfg, params, emb = toy
    got = OD.sampling_generate(params, cfg, emb, [3, 57], ("top_k", 1), seed=9, max_gen_len=(0, 9))
    greedy = OD.greedy_decode(params, cfg, emb, [3, 57], max_new=9)
    for (seq, score, steps), g in zip(got, greedy):
        m = len(seq) - 1                       # the last sampled token may be the forced EOS
        assert seq[:m] == g[:m] and seq[-1] == 3


def test_sampling_generate_scores_lengths_and_reproducibility(toy):
    cfg, params, emb = toy
    a = OD.sampling_generate(params, cfg, emb, [3, 57], ("top_p", 0.9), seed=11, min_gen_len=3, max_gen_len=(0, 7))
    b = OD.sampling_generate(params, cfg, emb, [3, 57], ("top_p", 0.9), seed=11, min_gen_len=3, max_gen_len=(0, 7))
    c = OD.sampling_generate(params, cfg, emb, [3, 57], ("top_p", 0.9), seed=12, min_gen_len=3, max_gen_len=(0, 7))
    assert [x[0] for x in a] == [x[0] for x in b] and [x[0] for x in a] != [x[0] for x in c]
    for e, (seq, score, steps) in zip(emb, a):
        assert 4 <= len(seq) <= 7 and seq[-1] == 3 and 3 not in seq[:-1] and 0 not in seq
        # score = (prompt log-prob + step log-probs) / (len incl. prompt - 1)
        full = torch.tensor([[3, 57] + seq])
        lp = torch.log_softmax(OD.decoder_logits(params, cfg, e.unsqueeze(0), full[:, :-1]), dim=-1)
        ref = lp[0, torch.arange(full.shape[1] - 1), full[0, 1:]]
        assert steps == pytest.approx(ref[1:].tolist(), abs=1e-4)
        assert score == pytest.approx(float(ref.sum()) / (full.shape[1] - 1), abs=1e-4)
    # a sentence's stream does not depend on its position in the batch
    d = OD.sampling_generate(params, cfg, emb[1:], [3, 57], ("top_p", 0.9), seed=11, min_gen_len=3, max_gen_len=(0, 7),
                             row_offset=1)
    assert [x[0] for x in d] == [x[0] for x in a[1:]]
