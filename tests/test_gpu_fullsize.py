"""GPU: the text encoder at the REAL widths (d = 1024, F = 8192) and at BASELINE's full configuration.

* 2 full-width layers on ~9.6 k tokens against the CPU oracle: every GEMM of the layer is large enough to
  run on the 256x256 tile engine with tile-major operands (the toy-width tests mostly exercise the
  128x128 engine).
* BASELINE configs[1] (24 layers, 1024 sentences x 128 tokens, random-init `basic` weights): the oracle
  would need minutes of CPU, so parity is checked through size-independent properties the reference
  itself asserts (tests/integration_tests/test_text_sonar.py:120-161): embeddings do not depend on how
  sentences are batched or ordered; plus run-to-run determinism.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cos_err(a, b):
    return (1 - F.cosine_similarity(a.float().cpu(), b.float().cpu(), dim=-1)).abs().max().item()


@pytest.mark.parametrize("fp16_residual", [False, True])
def test_encoder_full_width_vs_oracle(fp16_residual):
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import (PaddingMask, SequenceBatch, SonarTextEncoderConfig,
                                        SonarTextTransformerEncoderModel, VocabularyInfo)

    ocfg = O.OracleTextEncoderConfig(model_dim=1024, num_layers=2, num_heads=16, ffn_inner_dim=8192, vocab_size=5000)
    cfg = SonarTextEncoderConfig(model_dim=1024, num_encoder_layers=2, num_encoder_attn_heads=16, ffn_inner_dim=8192,
                                 vocab_info=VocabularyInfo(size=5000), _from_fairseq=True)
    params = O.make_synthetic_params(ocfg, seed=99, std=0.03)
    ids, lens = O.synthetic_batch(100, 64, 128, ocfg.vocab_size, seed=11)
    assert int(lens.sum()) >= 9216   # >= 36 row tiles: all four projections take the 256x256 engine
    torch.set_num_threads(min(32, torch.get_num_threads()))
    _, ref = O.text_encoder_forward(params, ocfg, ids, lens)
    model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32,
                                             fp16_residual=fp16_residual)
    emb = model(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))).sentence_embeddings
    assert torch.isfinite(emb).all()
    rel = (emb.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"fp16_residual={fp16_residual}: max (1 - cos) vs oracle = {_cos_err(emb, ref):.2e}, max |diff| / max |ref| = {rel:.2e}")
    assert _cos_err(emb, ref) <= 1e-3          # north_star tolerance
    # measured (r02): 1 - cos <= 3e-5; hold the stack to ~10x that and to an elementwise bound, so that a
    # regression inside the north_star tolerance is still caught
    assert _cos_err(emb, ref) <= 3e-4 and rel <= 3e-2
    # a small slice of the same sentences goes through the 128x128 engine: same vectors
    sub = model(SequenceBatch(ids[:3].cuda(), PaddingMask(lens[:3], ids.shape[1]))).sentence_embeddings
    assert _cos_err(sub, emb[:3]) <= 2e-5


@pytest.fixture(scope="module")
def basic_model():
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, get_text_encoder_config
    from tools.synth import text_encoder_state_dict

    dev = torch.device("cuda:0")
    sd = text_encoder_state_dict(dev)
    model = SonarTextTransformerEncoderModel(get_text_encoder_config("basic"), sd, device=dev, dtype=torch.float16)
    del sd
    torch.cuda.empty_cache()
    return model


def test_baseline_config_properties(basic_model):
    from sonar_amd.text_encoder import PaddingMask, SequenceBatch

    g = torch.Generator(device="cuda").manual_seed(5)
    n, s = 1024, 128
    ids = torch.randint(4, 256001, (n, s), device="cuda", generator=g)
    ids[:, 0] = 256047
    ids[:, -1] = 3
    full = basic_model(SequenceBatch(ids, None)).sentence_embeddings
    assert full.shape == (n, 1024) and torch.isfinite(full).all()
    # determinism: the same batch twice is bit-identical
    again = basic_model(SequenceBatch(ids, None)).sentence_embeddings
    assert torch.equal(full, again)
    # batching invariance: 8 of the sentences alone (small grids, 128x128 engine) and 256 of them
    for sl in (slice(40, 48), slice(512, 768)):
        part = basic_model(SequenceBatch(ids[sl].contiguous(), None)).sentence_embeddings
        assert _cos_err(part, full[sl]) <= 1e-3
    # order invariance: a permuted batch gives the permuted embeddings
    perm = torch.randperm(n, device="cuda", generator=g)
    shuffled = basic_model(SequenceBatch(ids[perm].contiguous(), None)).sentence_embeddings
    assert _cos_err(shuffled, full[perm]) <= 1e-3
    # padding invariance: the same sentences inside a ragged batch (right-padded with 0)
    lens = torch.randint(16, s + 1, (n,), generator=torch.Generator().manual_seed(6))
    lens[:4] = s
    ragged = ids.clone()
    ragged[:, -1] = torch.randint(4, 256001, (n,), device="cuda", generator=g)
    for i in range(4, n):
        L = int(lens[i])
        ragged[i, L - 1] = 3
        ragged[i, L:] = 0
    ragged[:4] = ids[:4]
    out = basic_model(SequenceBatch(ragged, PaddingMask(lens, s))).sentence_embeddings
    assert _cos_err(out[:4], full[:4]) <= 1e-3
    alone = basic_model(SequenceBatch(ragged[100:101, : int(lens[100])].contiguous(), None)).sentence_embeddings
    assert _cos_err(alone, out[100:101]) <= 1e-3


def test_xsim_large_known_neighbours():
    """xsim at a size the CPU oracle cannot score (16 k x 262 k x 1024): the ground-truth nearest
    neighbour is known by construction (X = normalise(Y[perm] + 0.3 noise)), k = 1 and
    k = 4 must agree with it and with each other, and scores must be the cosines of the pairs."""
    from sonar_amd import xsim

    g = torch.Generator(device="cuda").manual_seed(2)
    ny, nx, d = 262144, 16384, 1024
    y = F.normalize(torch.randn(ny, d, device="cuda", generator=g), dim=-1).half()
    perm = torch.randint(0, ny, (nx,), device="cuda", generator=g)
    x = F.normalize(y[perm].float() + 0.3 * torch.randn(nx, d, device="cuda", generator=g) / d ** 0.5, dim=-1).half()
    s1, i1 = xsim.topk(x, y, 1)
    s4, i4 = xsim.topk(x, y, 4)
    torch.cuda.synchronize()
    assert (i1[:, 0].long() == perm).float().mean().item() >= 0.999
    assert torch.equal(i1[:, 0], i4[:, 0]) and torch.equal(s1[:, 0], s4[:, 0])
    assert (s4[:, :-1] >= s4[:, 1:]).all()
    cos = (F.normalize(x.float(), dim=-1) * F.normalize(y[i1[:, 0].long()].float(), dim=-1)).sum(-1)
    assert (s1[:, 0] - cos).abs().max().item() <= 3e-3


def test_basic_decoder_greedy_is_teacher_forced_argmax():
    """Full-size `basic` decoder (24 layers, V = 256206, random-init weights): beam_size = 1 must emit, at
    every step, the arg-max of the teacher-forced next-token distribution of the tokens emitted so far
    (the reference's greedy / logits consistency, test_text_sonar.py:61-118), its score must be the
    length-normalised sum of those log-probs, and beam 5's best hypothesis can only score higher."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sonar_amd.text_decoder import TextDecoderEngine, get_text_decoder_config
    from tools.synth import text_decoder_state_dict

    dev = torch.device("cuda:0")
    eng = TextDecoderEngine(get_text_decoder_config("basic"), text_decoder_state_dict(dev), device=dev)
    torch.cuda.empty_cache()
    g = torch.Generator(device="cuda").manual_seed(21)
    n, steps = 6, 9
    emb = F.normalize(torch.randn(n, 1024, device="cuda", generator=g), dim=-1).half() * 0.2
    prompt = [3, 256047]
    toks, lens, scores = eng.generate(emb, prompt, beam_size=1, min_gen_len=steps, max_gen_len=(0, steps))
    toks5, lens5, scores5 = eng.generate(emb, prompt, beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
    torch.cuda.synchronize()
    toks, lens, scores = toks.cpu(), lens.cpu(), scores.cpu()
    for i in range(n):
        L = int(lens[i, 0])
        seq = toks[i, 0, :L].tolist()
        assert seq[-1] == 3 and L == steps and 3 not in seq[:-1]   # EOS blocked below min_gen_len, forced at the cap
        nfree = L - 1
        full = torch.tensor([prompt + seq], device="cuda")
        lp = torch.log_softmax(eng.logits(emb[i:i + 1], full[:, :-1]), dim=-1)[0]     # [len(prompt) + L - 1, V]
        # the hypothesis score also carries the forced prompt tokens after the first one (fairseq2 scores
        # every step it feeds)
        total = sum(lp[j][prompt[j + 1]].item() for j in range(len(prompt) - 1))
        for t, tok in enumerate(seq):
            row = lp[len(prompt) - 1 + t].clone()
            row[0] = float("-inf")                    # PAD is never generated
            if t < nfree:
                row[3] = float("-inf")                # EOS blocked below min_gen_len
                best = row.max().item()
                assert row[tok].item() >= best - 2e-3, (i, t, tok, row[tok].item(), best)
            total += row[tok].item() if t < nfree else lp[len(prompt) - 1 + t][3].item()
        norm = total / (len(prompt) + L - 1)
        assert abs(norm - scores[i, 0].item()) <= 2e-2
        assert scores5[i, 0].item() >= scores[i, 0].item() - 2e-2


def test_speech_encoder_full_size_properties():
    """sonar_speech_encoder_eng at full size (24 conformer blocks, d = 1024, random-init): batching
    invariance of SpeechToEmbedding (reference: tests/integration_tests/test_sonar_speech_pipeline_models.py
    batch-vs-single checks) and determinism, on clips of different lengths (1.3 - 6 s)."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sonar_amd.speech_encoder import SpeechEncoderEngine, get_speech_encoder_config, waveform_to_fbank
    from tools.synth import speech_encoder_state_dict

    dev = torch.device("cuda:0")
    eng = SpeechEncoderEngine(get_speech_encoder_config("english"), speech_encoder_state_dict(dev), device=dev)
    g = torch.Generator(device="cuda").manual_seed(9)
    nsamp = [96000, 21000, 64000, 33333, 80640]
    feats = [waveform_to_fbank(torch.rand(n, device=dev, generator=g) * 2 - 1) for n in nsamp]
    tmax = max(f.shape[0] for f in feats)
    tmax += tmax % 2
    batch = torch.zeros(len(feats), tmax, 80, device=dev)
    lens = torch.tensor([f.shape[0] for f in feats])
    for i, f in enumerate(feats):
        batch[i, : f.shape[0]] = f
    out = eng.forward(batch, lens, torch.float32)
    assert out.shape == (5, 1024) and torch.isfinite(out).all()
    assert torch.equal(out, eng.forward(batch, lens, torch.float32))
    for i, f in enumerate(feats):
        t = f.shape[0] + f.shape[0] % 2
        one = torch.zeros(1, t, 80, device=dev)
        one[0, : f.shape[0]] = f
        alone = eng.forward(one, torch.tensor([f.shape[0]]), torch.float32)
        assert _cos_err(alone, out[i:i + 1]) <= 1e-3


@pytest.mark.parametrize("fp16_residual", [False, True])
def test_encoder_full_depth_vs_oracle(fp16_residual):
    """All 24 layers at the real widths (d = 1024, F = 8192, 16 heads; small vocabulary) against the fp32
    CPU oracle on a small ragged batch: the accumulated error of the whole stack, for both residual-stream
    precisions, against the north_star bound."""
    from oracle import text_encoder as O
    from sonar_amd.text_encoder import (PaddingMask, SequenceBatch, SonarTextEncoderConfig,
                                        SonarTextTransformerEncoderModel, VocabularyInfo)

    ocfg = O.OracleTextEncoderConfig(model_dim=1024, num_layers=24, num_heads=16, ffn_inner_dim=8192, vocab_size=3000)
    cfg = SonarTextEncoderConfig(model_dim=1024, num_encoder_layers=24, num_encoder_attn_heads=16, ffn_inner_dim=8192,
                                 vocab_info=VocabularyInfo(size=3000), _from_fairseq=True)
    params = O.make_synthetic_params(ocfg, seed=2024, std=0.02)
    ids, lens = O.synthetic_batch(6, 12, 64, ocfg.vocab_size, seed=4)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    _, ref = O.text_encoder_forward(params, ocfg, ids, lens)
    for dt in (torch.float32, torch.float16):
        model = SonarTextTransformerEncoderModel(cfg, params, device="cuda:0", dtype=dt, fp16_residual=fp16_residual)
        emb = model(SequenceBatch(ids.cuda(), PaddingMask(lens, ids.shape[1]))).sentence_embeddings
        err = _cos_err(emb, ref)
        rel = (emb.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
        print(f"24 layers, fp16_residual={fp16_residual}, out {dt}: max (1 - cos) vs fp32 oracle = {err:.2e}, "
              f"max |diff| / max |ref| = {rel:.2e}")
        assert torch.isfinite(emb).all() and err <= 1e-3
        assert err <= 3e-4 and rel <= 3e-2      # ~10x the measured error (2e-7 ... 3e-5), elementwise too
        del model


# --------------------------------------------------------------------------------------------------
# BASELINE configs[4] / configs[3] at FULL size against the fp32 CPU oracle (round 2: VERDICT item 1b)
def _cpu_fp32(sd):
    return {k: v.detach().float().cpu() for k, v in sd.items()}


@pytest.fixture(scope="module")
def basic_decoder():
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import text_decoder as OD
    from sonar_amd.text_decoder import TextDecoderEngine, get_text_decoder_config
    from tools.synth import text_decoder_state_dict

    dev = torch.device("cuda:0")
    sd = text_decoder_state_dict(dev)
    eng = TextDecoderEngine(get_text_decoder_config("basic"), sd, device=dev)
    params = _cpu_fp32(sd)          # the same (fp16-representable) values, fp32 arithmetic on the host
    del sd
    torch.cuda.empty_cache()
    ocfg = OD.OracleTextDecoderConfig(model_dim=1024, num_layers=24, num_heads=16, ffn_inner_dim=8192,
                                      vocab_size=256206, max_seq_len=512)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    return OD, ocfg, params, eng


# observed on MI355X (r06h): 1.0000 on 36 / 140 / 24 (row, position) pairs.  One flip between two candidates the fp16 arithmetic
# cannot separate is box-to-box noise (1 / 36 = 2.8 %); anything beyond that is a defect.
ARGMAX_AGREE_MIN = 0.97


def _argmax_agreement(got, ref):
    """Share of (row, position) pairs whose arg-max token equals the oracle's; where it differs the oracle's top-2 gap must
    be within the logit error bound of the test (a flip between two candidates the arithmetic cannot separate)."""
    ga, ra = got.argmax(-1), ref.argmax(-1)
    agree = (ga == ra).float().mean().item()
    bad = (ga != ra).nonzero()
    worst = 0.0
    for idx in bad.tolist():
        row = ref[tuple(idx)]
        worst = max(worst, (row.max() - row[ga[tuple(idx)]]).item())
    print(f"arg-max agreement with the oracle: {agree:.4f} ({bad.shape[0]} of {ga.numel()} differ; largest oracle gap at a "
          f"differing position {worst:.3e})")
    assert agree >= ARGMAX_AGREE_MIN, agree
    assert worst <= 2 * 5e-3 * ref.abs().max().item(), worst
    return agree


def test_basic_decoder_logits_vs_oracle_full_size(basic_decoder):
    """text_sonar_basic_decoder (24 layers, d 1024, F 8192, V 256206): teacher-forced logits of 4 sentences x 9
    positions against the fp32 oracle (the reference's own logits check has this shape, test_text_sonar.py:61-105)."""
    OD, ocfg, params, eng = basic_decoder
    g = torch.Generator().manual_seed(41)
    emb = F.normalize(torch.randn(4, 1024, generator=g), dim=-1) * 0.2
    prev = torch.randint(4, 256000, (4, 9), generator=g)
    prev[:, 0] = 3
    prev[:, 1] = 256047
    ref = OD.decoder_logits(params, ocfg, emb, prev)
    got = eng.logits(emb.cuda(), prev.cuda()).cpu()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    print(f"basic decoder logits: max |diff| {err:.3e} on scale {scale:.3f} ({err / scale:.2e} relative)")
    assert got.shape == ref.shape == (4, 9, 256206)
    assert err <= 5e-3 * scale
    _argmax_agreement(got, ref)


@pytest.mark.parametrize("beam", [1, 5])
def test_basic_decoder_tokens_vs_oracle_full_size(basic_decoder, beam):
    """Exact token ids at full size: every best hypothesis equals the fp32 oracle's unless the ORACLE's own decision margin
    is below 1e-3 of the logit range (tests/neartie.py); the engine's reported margin must agree with the oracle's."""
    from tests.neartie import check_engine_margin, oracle_excuses

    OD, ocfg, params, eng = basic_decoder
    g = torch.Generator().manual_seed(43 + beam)
    n, steps = 4, 9
    emb = F.normalize(torch.randn(n, 1024, generator=g), dim=-1) * 0.2
    prompt = [3, 256047]
    kw = dict(beam_size=beam, max_gen_len=(0, steps))
    om = []
    ref = OD.beam_search_incremental(params, ocfg, emb, prompt, margins_out=om, **kw)
    toks, lens, scores = eng.generate(emb.cuda(), prompt, **kw)
    margins = eng.last_margins(n).cpu()
    toks, lens, scores = toks.cpu(), lens.cpu(), scores.cpu()
    lg = OD.decoder_logits(params, ocfg, emb, torch.tensor([prompt] * n))
    eps = 1e-3 * (lg.max() - lg.min()).item()
    excused = 0
    for i in range(n):
        seq = toks[i, 0, : int(lens[i, 0])].tolist()
        want = ref[i][0].seq.tolist()
        check_engine_margin(margins[i], om[i], eps, f"beam {beam}, sentence {i}")
        assert abs(scores[i, 0].item() - ref[i][0].score) <= 5e-3 or seq != want
        if seq != want:
            assert oracle_excuses(om[i], eps), (i, seq, want, om[i], margins[i].tolist(), eps)
            excused += 1
    print(f"basic decoder beam {beam}: {n - excused}/{n} token-identical to the oracle, eps {eps:.2e}, "
          f"oracle margins {om}, engine margins {margins.tolist()}")
    assert excused <= 1


def test_basic_decoder_long_prefix_vs_oracle_full_size(basic_decoder):
    """Round 3: the full-size decoder beyond toy lengths.  (1) teacher-forced logits of 2 sentences x 70 positions
    (single-query attention with NG = 1..8 and two online-softmax passes; the C5 bench leg times 65 such steps);
    (2) greedy decoding with 70 forced steps, every decision checked against the oracle's arg-max on the engine's
    own prefix; (3) beam 5 with 70 forced steps: the best hypothesis re-scored by the oracle (its cumulative score
    went through the ancestry table and the KV cache for 70 steps)."""
    OD, ocfg, params, eng = basic_decoder
    g = torch.Generator().manual_seed(61)
    n, t = 2, 70
    emb = F.normalize(torch.randn(n, 1024, generator=g), dim=-1) * 0.2
    prev = torch.randint(4, 256000, (n, t), generator=g)
    prev[:, 0] = 3
    prev[:, 1] = 256047
    ref = OD.decoder_logits(params, ocfg, emb, prev)
    got = eng.logits(emb.cuda(), prev.cuda()).cpu()
    scale = ref.abs().max().item()
    err = (got - ref).abs().amax(dim=(0, 2))
    print(f"basic decoder logits over {t} positions: max |diff| / scale {err.max().item() / scale:.2e} "
          f"(worst position {int(err.argmax())}), last position {err[-1].item() / scale:.2e}")
    assert err.max().item() <= 5e-3 * scale
    _argmax_agreement(got, ref)
    del got, ref

    prompt = [3, 256047]
    forced = 70
    kw = dict(min_gen_len=forced, max_gen_len=(0, forced + 6))
    max_len, min_len = len(prompt) + forced + 6, len(prompt) + forced
    lg = OD.decoder_logits(params, ocfg, emb, torch.tensor([prompt] * n))
    eps = 1e-3 * (lg.max() - lg.min()).item()
    for beam in (1, 5):
        toks, lens, scores = eng.generate(emb.cuda(), prompt, beam_size=beam, **kw)
        toks, lens, scores = toks.cpu(), lens.cpu(), scores.cpu()
        ties = 0
        for i in range(n):
            L = int(lens[i, 0])
            seq = toks[i, 0, :L].tolist()
            assert forced + 1 <= L <= forced + 6 and seq[-1] == 3 and 3 not in seq[:-1] and 0 not in seq
            full = torch.tensor([prompt + seq])
            lp = torch.log_softmax(OD.decoder_logits(params, ocfg, emb[i:i + 1], full[:, :-1]), dim=-1, dtype=torch.float32)[0]
            total = lp[torch.arange(full.shape[1] - 1), full[0, 1:]].sum().item()
            assert abs(total / (len(prompt) + L - 1) - scores[i, 0].item()) <= 5e-3, (beam, i, total, scores[i, 0].item())
            if beam != 1:
                continue
            for tt in range(len(prompt) - 1, full.shape[1] - 1):
                step_nr, chosen = tt + 1, int(full[0, tt + 1])
                if step_nr == max_len - 1:
                    continue
                row = lp[tt].clone()
                row[0] = -torch.inf
                if step_nr < min_len:
                    row[3] = -torch.inf
                best = int(row.argmax())
                if chosen != best:
                    assert (row[best] - row[chosen]).item() < eps, (i, step_nr, chosen, best)
                    ties += 1
        print(f"basic decoder beam {beam}, {forced} forced steps: scores equal the oracle's re-scoring"
              + (f"; greedy decisions within eps {eps:.2e} of the arg-max: {ties}" if beam == 1 else ""))
        assert ties <= 2


def test_basic_decoder_chains_bit_identical_full_size(basic_decoder):
    """Independent decode chains at full size (24 layers, V 256 206): 520 sentences x beam 5 = 2 600 hypothesis rows (2 816 padded) run as
    three chains by default.  With the per-launch tile choices pinned the hypotheses, lengths, scores and
    margins equal the single chain's bit for bit, sentence for sentence (the toy-width twin of this test covers 3 chains and
    uneven groups, tests/test_gpu_decoder.py)."""
    OD, ocfg, params, eng = basic_decoder
    g = torch.Generator(device="cuda").manual_seed(61)
    n = 520
    emb = F.normalize(torch.randn(n, 1024, device="cuda", generator=g), dim=-1).half() * 0.2
    kw = dict(beam_size=5, min_gen_len=5, max_gen_len=(0, 6))
    from sonar_amd import _lib

    # DEC_KS_OUT / DEC_FFN1_ENGINE: the decode step's own per-launch choices.
    # G2_SPLITK_MIN: the split-K FFN output projection: 2 816 rows are 352 units of the 256x256 engine (more than one round: the
    # 128x128 family takes it), a chain's 1 024 rows are 128 units (the 256x256 engine takes it) -- two MFMA shapes, two fp32
    # summation orders.  Pin the family for both.
    # G2_AUTO_MIN: ... and the fused QKV projection: 2 816 rows x 3 072 columns are 132 tiles of the 256x256 engine (its automatic
    # choice from 128 tiles up), a chain's 1 024 rows are 48 (the 128x128 family)
    # DEC_M160: the lone 128 / 160 / 192-row units are chosen from the row count (round 6): off, like the other row-count choices
    with _lib.tuning(DEC_KS_OUT=2, DEC_FFN1_ENGINE=2, G2_SPLITK_MIN=1000000, G2_AUTO_MIN=1000000, DEC_M160=0):
        try:
            eng.set_chains(1)
            one = [t.cpu() for t in eng.generate(emb, [3, 256047], **kw)]
            m_one = eng.last_margins(n).cpu()
            eng.set_chains(0)                       # the engine's own policy: min(3, ceil(2816 / 1280)) = 3 chains
            two = [t.cpu() for t in eng.generate(emb, [3, 256047], **kw)]
            m_two = eng.last_margins(n).cpu()
        finally:
            eng.set_chains(0)
    for a, b in zip(one, two):
        assert torch.equal(a, b)
    assert torch.equal(m_one, m_two)


C5_SAMPLE = [0, 51, 52, 101, 127, 128, 204, 255]     # rows 0, 255|256|260, 505, 635, 640, 1 020|1 024, 1 275..1 279


def _check_hyps_vs_oracle(tag, ref, omargins, toks, lens, scores, margins, where, eps, max_excused):
    """Best hypothesis of sentence where[j] of the GPU call == the oracle's j-th; a mismatch is excused ONLY by the ORACLE's own
    near-tie measurement (tests/neartie.py), and the engine's reported decision margin must agree with the oracle's."""
    from tests.neartie import check_engine_margin, oracle_excuses

    excused = 0
    for j, i in enumerate(where):
        seq = toks[i, 0, : int(lens[i, 0])].tolist()
        want = ref[j][0].seq.tolist()
        check_engine_margin(margins[i], omargins[j], eps, f"{tag}, sentence {i}")
        if seq == want:
            assert abs(scores[i, 0].item() - ref[j][0].score) <= 5e-3, (tag, i, scores[i, 0].item(), ref[j][0].score)
        else:
            assert oracle_excuses(omargins[j], eps), (tag, i, seq, want, omargins[j], margins[i].tolist(), eps)
            excused += 1
    print(f"basic decoder C5 shape [{tag}]: {len(where) - excused}/{len(where)} sampled best hypotheses token-identical to the oracle "
          f"(eps {eps:.2e}; oracle decision margins {[round(m[0], 4) for m in omargins]}, engine's {[round(m, 4) for m in margins[where, 0].tolist()]})")
    assert excused <= max_excused
    return excused


def test_basic_decoder_c5_shape_vs_oracle_full_size(basic_decoder):
    """VERDICT r4 "weak" 1: the decoder at the row count that is BENCHMARKED (BASELINE configs[4]: 256 sentences x beam 5 = 1 280
    hypothesis rows: 256x256 engine for the FFN pair, 8-part split-K with fp16 slabs, tile-major fp16 logits / the fused
    selection epilogue) against the fp32 oracle -- not against itself.  Sentences are independent, so the oracle scores 8 of the
    256, spread over the row tiles.  (1) beam search, default fp16 storage and fp32 storage; (2) teacher-forced logits of all
    1 280 rows (fp32 storage path of the same engines), sampled rows against the oracle's; (3) 512 sentences (2 560 rows), the
    engine's default two chains, the same 8 embeddings interleaved over both chains.
    Reference checks of this shape: tests/integration_tests/test_text_sonar.py:61-118."""
    OD, ocfg, params, eng = basic_decoder
    g = torch.Generator().manual_seed(77)
    n, steps = 256, 12
    emb = F.normalize(torch.randn(n, 1024, generator=g), dim=-1) * 0.2
    emb = emb.half().float()                 # what the fp16 pipeline hands over; the oracle sees the same values
    prompt = [3, 256047]
    kw = dict(beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
    om = []
    ref = OD.beam_search_incremental(params, ocfg, emb[C5_SAMPLE], prompt, margins_out=om, **kw)
    lg = OD.decoder_logits(params, ocfg, emb[:1], torch.tensor([prompt]))
    eps = 1e-3 * (lg.max() - lg.min()).item()
    try:
        for dt in (torch.float16, torch.float32):
            eng.set_beam_logits_dtype(dt)
            eng.set_slab_dtype(dt)
            toks, lens, scores = [t.cpu() for t in eng.generate(emb.cuda().half(), prompt, **kw)]
            margins = eng.last_margins(n).cpu()
            assert (lens[:, 0] == steps).all()           # min_gen_len == max_gen_len: every hypothesis ends at the cap
            _check_hyps_vs_oracle(f"256 sentences, {dt} storage", ref, om, toks, lens, scores, margins, C5_SAMPLE, eps, 1)
    finally:
        eng.set_beam_logits_dtype(torch.float16)
        eng.set_slab_dtype(torch.float16)

    # (3) twice the batch: the sampled embeddings sit at the even positions, so sentences 0..254 / 256..510 fall into both chains
    g2 = torch.Generator().manual_seed(78)
    emb2 = torch.empty(2 * n, 1024)
    emb2[0::2] = emb
    emb2[1::2] = (F.normalize(torch.randn(n, 1024, generator=g2), dim=-1) * 0.2).half().float()
    toks, lens, scores = [t.cpu() for t in eng.generate(emb2.cuda().half(), prompt, **kw)]
    margins = eng.last_margins(2 * n).cpu()
    _check_hyps_vs_oracle("512 sentences, default chains", ref, om, toks, lens, scores, margins, [2 * i for i in C5_SAMPLE], eps, 1)
    del toks, lens, scores

    # (2) teacher-forced logits, 1 280 rows x 3 positions: row r carries embedding r // 5 (the beam layout of the C5 call)
    t = 3
    rows = 5 * n
    prev = torch.randint(4, 256000, (rows, t), generator=g)
    prev[:, 0] = 3
    prev[:, 1] = 256047
    emb_rows = emb.repeat_interleave(5, dim=0)
    sample_rows = [0, 255, 256, 260, 635, 1020, 1024, 1279]
    want = OD.decoder_logits(params, ocfg, emb_rows[sample_rows], prev[sample_rows])
    got = eng.logits(emb_rows.cuda(), prev.cuda())[sample_rows].cpu()
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    print(f"basic decoder logits at 1280 rows: max |diff| {err:.3e} on scale {scale:.3f} ({err / scale:.2e} relative)")
    assert err <= 5e-3 * scale
    _argmax_agreement(got, want)


def test_speech_encoder_english_10s_clip_vs_oracle_full_size():
    """BASELINE configs[3] at its own shape: ONE 10 s clip (998 filterbank frames -> 499 conformer frames, relative
    positions out to +-498, eight 64-key tiles in the relative-position attention) through the full
    sonar_speech_encoder_eng, waveform -> embedding, against the fp32 oracle; fp16 residual stream as benchmarked."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import speech_encoder as OS
    from sonar_amd.speech_encoder import SpeechEncoderEngine, get_speech_encoder_config, waveforms_to_fbank_batch
    from tools.synth import speech_encoder_state_dict

    dev = torch.device("cuda:0")
    sd = speech_encoder_state_dict(dev)
    params = _cpu_fp32(sd)
    so = OS.OracleSpeechEncoderConfig(model_dim=1024, num_layers=24, num_heads=16, ffn_inner_dim=4096, conv_kernel=31,
                                      pooler_layers=3, pooler_heads=16, pooler_ffn_dim=4096, pooler_vocab=1024)
    g = torch.Generator().manual_seed(23)
    ns = 160000
    wav = (torch.rand(ns, generator=g) * 2 - 1) * 0.5 + 0.2 * torch.sin(torch.arange(ns) * 0.03)
    f = OS.kaldi_fbank(wav)
    assert f.shape[0] == 998
    torch.set_num_threads(min(32, torch.get_num_threads()))
    _, ref = OS.speech_encoder_forward(params, so, f.unsqueeze(0), torch.tensor([998]))
    eng = SpeechEncoderEngine(get_speech_encoder_config("english"), sd, device=dev, fp16_residual=True)
    gfb, glens = waveforms_to_fbank_batch([wav.to(dev)])
    assert glens == [998]
    out = eng.forward(gfb, glens, torch.float32).cpu()
    err = _cos_err(out, ref)
    rel = (out - ref).abs().max().item() / ref.abs().max().item()
    print(f"english speech encoder, one 10 s clip (499 frames): max (1 - cos) {err:.2e}, max |diff| / max |ref| {rel:.2e}")
    assert err <= 1e-3 and rel <= 5e-2
    # the same clip inside a batch of shorter ones gives the same vector
    wavs = [wav.to(dev), wav[:52000].to(dev), wav[:160000 - 321].to(dev)]
    gfb, glens = waveforms_to_fbank_batch(wavs)
    outb = eng.forward(gfb, glens, torch.float32).cpu()
    assert _cos_err(outb[:1], out) <= 1e-5


def test_speech_encoder_streams_agree_at_benchmark_rows():
    """The tile-major residual stream with the LayerNorm fold (round 4, the default for fp16 models) against the
    row-major stream with LayerNorm launches, on a batch whose GEMMs give every persistent workgroup SEVERAL tiles
    (26 clips x 5 s = 6 474 frames = 26 row tiles: 104-416 tiles per GEMM on 256 CUs) -- per-tile state that only a
    second tile of a workgroup can clobber (the staged fold epilogue's LDS constants, found the hard way) is invisible
    to the short-clip oracle tests above.  The two engines differ by fp16 roundings of h only."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sonar_amd.speech_encoder import SpeechEncoderEngine, get_speech_encoder_config, waveforms_to_fbank_batch
    from tools.synth import speech_encoder_state_dict

    dev = torch.device("cuda:0")
    sd = speech_encoder_state_dict(dev)
    cfg = get_speech_encoder_config("english")
    g = torch.Generator(device=dev).manual_seed(31)
    wavs = [torch.rand(80000 - 37 * i, device=dev, generator=g) * 2 - 1 for i in range(26)]
    fb, lens = waveforms_to_fbank_batch(wavs)
    from sonar_amd import _lib

    with _lib.tuning(SPEECH_X_TM=0):                       # read when the engine is created
        rowmajor = SpeechEncoderEngine(cfg, sd, device=dev, fp16_residual=True)
    want = rowmajor.forward(fb, lens, torch.float32).cpu()
    del rowmajor
    eng = SpeechEncoderEngine(cfg, sd, device=dev, fp16_residual=True)
    got = eng.forward(fb, lens, torch.float32).cpu()
    again = eng.forward(fb, lens, torch.float32).cpu()
    assert torch.isfinite(got).all() and torch.equal(got, again)
    err = _cos_err(got, want)
    rel = (got - want).abs().max().item() / want.abs().max().item()
    print(f"english speech encoder, 26 x 5 s: tile-major stream + LN fold vs row-major stream: max (1 - cos) {err:.2e}, "
          f"max |diff| / max |ref| {rel:.2e}")
    assert err <= 1e-5 and rel <= 2e-2


def test_speech_encoder_english_vs_oracle_full_size():
    """sonar_speech_encoder_eng (24 conformer blocks, d 1024, 3-layer pooler) on 3 clips of 1.0 / 1.6 / 2.0 s,
    waveform -> embedding, against the fp32 oracle (filterbank included), both residual precisions."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import speech_encoder as OS
    from sonar_amd.speech_encoder import SpeechEncoderEngine, get_speech_encoder_config, waveforms_to_fbank_batch
    from tools.synth import speech_encoder_state_dict

    dev = torch.device("cuda:0")
    sd = speech_encoder_state_dict(dev)
    params = _cpu_fp32(sd)
    so = OS.OracleSpeechEncoderConfig(model_dim=1024, num_layers=24, num_heads=16, ffn_inner_dim=4096, conv_kernel=31,
                                      pooler_layers=3, pooler_heads=16, pooler_ffn_dim=4096, pooler_vocab=1024)
    g = torch.Generator().manual_seed(17)
    wavs = [(torch.rand(n, generator=g) * 2 - 1) * 0.5 + 0.2 * torch.sin(torch.arange(n) * 0.03) for n in (16000, 25600, 32000)]
    feats = [OS.kaldi_fbank(w) for w in wavs]
    lens = torch.tensor([f.shape[0] for f in feats])
    t = int(lens.max()) + int(lens.max()) % 2
    fb = torch.zeros(3, t, 80)
    for i, f in enumerate(feats):
        fb[i, : f.shape[0]] = f
    torch.set_num_threads(min(32, torch.get_num_threads()))
    _, ref = OS.speech_encoder_forward(params, so, fb, lens)
    for fp16_residual in (False, True):
        eng = SpeechEncoderEngine(get_speech_encoder_config("english"), sd, device=dev, fp16_residual=fp16_residual)
        gfb, glens = waveforms_to_fbank_batch([w.to(dev) for w in wavs])
        assert glens == lens.tolist()
        out = eng.forward(gfb, glens, torch.float32).cpu()
        err = _cos_err(out, ref)
        rel = (out - ref).abs().max().item() / ref.abs().max().item()
        print(f"english speech encoder, fp16_residual={fp16_residual}: max (1 - cos) {err:.2e}, max |diff| / max |ref| {rel:.2e}")
        assert err <= 1e-3 and rel <= 5e-2
        del eng
