#!/usr/bin/env python
"""Headline benchmark: sentences/s embedded by text_sonar_basic_encoder
(fp16, batch 1024, seq_len 128 -- BASELINE.json configs[1]) on N MI355X, plus
xsim pairs/s at BASELINE configs[2] scale (1M x 1M x 1024), the roofline of the
dominant kernel and the CPU oracle baseline timed on the same box in the same run.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (smi_text_encoder_forward: embed -> 24 layers ->
LayerNorm -> mean-pool) over one synthetic 1024 x 128 batch per GPU, inputs resident in
HBM; for N > 1 each rank encodes its own batch (weak scaling, sentences are independent:
SURVEY 8(e)) and the step ends with the RCCL all-gather that assembles the embedding
matrix.  Rank 0 prints ONE JSON line on stdout; everything else goes to stderr.

Extra objects in the same line (N = 1): `varlen` (C2(b): lengths randint(16,129)), `c1`
(BASELINE configs[0]: 32 sentences <= 64 tokens -- GPU time, CPU-oracle time and the
1 - cos between the two on identical weights and inputs), `xsim` (+ its own
`cpu_baseline`), `decoder` (C5), `speech` (C4).

SONAR_BENCH_DRYRUN=1 swaps the engine for a CPU stub and RCCL for gloo: it exists so
that the N > 1 control flow (process group, the all-gathers inside the timed step, the
max-over-ranks timing, the sharded xsim leg) is executed by tests/test_bench_dryrun_cpu.py
without GPUs.  A dry run prints "data": "dry-run stub" and is never a measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.synth import BATCH, SEQ, D, F, L, H, V, text_encoder_state_dict  # noqa: E402
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic flops: per token per layer 8 d^2 + 4 d F + 4 S d
FLOPS_PER_SENTENCE = SEQ * L * (8 * D * D + 4 * D * F + 4 * SEQ * D)
FFN1_FLOPS_PER_LAUNCH = 2.0 * BATCH * SEQ * F * D  # one launch = the whole 131072-token batch
DRYRUN = os.environ.get("SONAR_BENCH_DRYRUN") == "1"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def flops_of_lengths(lens) -> float:
    """sum_i L_i * 24 * (8 d^2 + 4 d F + 4 L_i d)."""
    return float(sum(int(n) * L * (8 * D * D + 4 * D * F + 4 * int(n) * D) for n in lens))


def cpu_threads() -> int:
    import torch

    # measured on the MI355X box's host (2 x EPYC 9575F, 256 hw threads, shared): the torch CPU
    # oracle peaks at 16 threads (6.3 sent/s) and degrades with more (1.4 sent/s at 128)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, avail)))
    return torch.get_num_threads()


def best_of(fn, reps=3):
    """warm-up 1 + best of `reps`; returns (best seconds, all timed seconds)."""
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), ts


def cpu_baseline(params, c2_ids, c1_ids, c1_lens):
    """Reference-equivalent op sequence (the oracle, kind 'port') on the host cores, on the SAME weights
    the GPU engine holds: the first rows of the ACTUAL C2 batch the GPU step encodes (c2_ids, [n_sent, 128])
    and the whole C1 batch.  Returns (record, C2-slice embeddings, C1 embeddings) for the parity fields."""
    from oracle import text_encoder as O

    cores = cpu_threads()
    cfg = O.OracleTextEncoderConfig()
    n_sent = c2_ids.shape[0]
    c2_out = {}

    def run_c2():
        c2_out["emb"] = O.text_encoder_forward(params, cfg, c2_ids, None)[1]

    best, ts = best_of(run_c2)
    c1_out = {}

    def run_c1():
        c1_out["emb"] = O.text_encoder_forward(params, cfg, c1_ids, c1_lens)[1]

    c1_best, c1_ts = best_of(run_c1)
    return ({"value": n_sent / best, "unit": "sentences/s", "cores": cores, "kind": "port",
             "sample": f"the first {n_sent} sentences x {SEQ} tokens of the C2 batch the GPU step encodes (same ids, same "
                       f"weights), full 24-layer fp32 model, torch CPU oracle, warm-up 1 + best of 3 "
                       f"({', '.join(f'{t:.1f}' for t in ts)} s)",
             "c1": {"value": c1_ids.shape[0] / c1_best, "unit": "sentences/s", "seconds": c1_best,
                    "sample": f"BASELINE configs[0]: {c1_ids.shape[0]} sentences, lengths randint(8,65) seed 0 "
                              f"({int(c1_lens.sum())} tokens), fp32, warm-up 1 + best of 3"}},
            c2_out["emb"], c1_out["emb"])


def xsim_cpu_baseline(nx=8192, ny=32768):
    """Blocked fp32 normalise + matmul + top-1 on the host (the oracle's cosine_topk), extrapolated per pair."""
    import torch

    from oracle import xsim as OX

    cores = cpu_threads()
    g = torch.Generator().manual_seed(2)
    y = torch.randn(ny, D, generator=g)
    x = y[torch.randint(0, ny, (nx,), generator=g)] + 0.3 * torch.randn(nx, D, generator=g)
    best, ts = best_of(lambda: OX.cosine_topk(x, y, 1), reps=2)
    return {"value": nx * ny / best, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{nx} x {ny} x {D} fp32 slice, torch CPU blocked matmul + top-1 (oracle/xsim.py), warm-up 1 + "
                      f"best of 2 ({', '.join(f'{t:.1f}' for t in ts)} s), extrapolated per pair"}


def xsim_margin_cpu_baseline(n=8192):
    """LASER's margin xsim (the oracle's `laser_xsim`: forward + backward k-NN, k = 4, ratio margin) on the host."""
    import torch

    from oracle import xsim as OX

    g = torch.Generator().manual_seed(2)
    y = torch.randn(n, D, generator=g)
    x = y + 0.3 * torch.randn(n, D, generator=g)
    res = {}

    def run():
        res["err"] = OX.laser_xsim(x, y, "ratio", 4)[0]

    best, ts = best_of(run, reps=2)
    return {"value": n * n / best, "unit": "pairs/s", "cores": cpu_threads(), "kind": "port", "errors": res["err"],
            "sample": f"{n} x {n} x {D} fp32 aligned slice, oracle/xsim.py laser_xsim(ratio, k = 4), warm-up 1 + best of 2 "
                      f"({', '.join(f'{t:.1f}' for t in ts)} s), extrapolated per pair"}


def decoder_leg(dev, n=256, steps=64, cpu=True):
    """BASELINE configs[4]: text_sonar_basic_decoder, beam 5, fp16, batch 256, `steps` forced steps; CPU baseline:
    the oracle's incremental beam search (the reference's evaluation order) on a bounded sample, same weights."""
    import torch

    from sonar_amd.text_decoder import TextDecoderEngine, get_text_decoder_config
    from tools.synth import text_decoder_state_dict

    sd = text_decoder_state_dict(dev)
    eng = TextDecoderEngine(get_text_decoder_config("basic"), sd, device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    emb = torch.nn.functional.normalize(torch.randn(n, D, device=dev, generator=g), dim=-1).half() * 0.2
    # untimed warm-up with the same shapes: the first call allocates the KV cache / logits workspace
    eng.generate(emb, [3, 256047], beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.generate(emb, [3, 256047], beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flop_step = n * 5 * (L * (16 * D * D + 4 * D * F) + 2 * D * V)
    out = {"workload": f"text_sonar_basic_decoder beam 5 fp16, batch {n}, {steps + 1} decode steps (EOS blocked), eng_Latn prompt",
           "ms": dt * 1e3, "ms_per_step": dt * 1e3 / (steps + 1), "sentences_per_s": n / dt,
           "tokens_per_s": n * (steps + 1) / dt,
           "frac_of_mfma_peak": flop_step * (steps + 1) / dt / 1e12 / MFMA_PEAK_TFLOPS}
    out["chains"] = 1   # decode_chains(): one chain up to 2048 hypothesis rows
    # what "fp16" means in this leg since round 4: fp16 operands into fp32 accumulators (as before), an fp32 residual stream,
    # and -- because the model is an fp16 model -- fp16 STORAGE of the logits and of the split-K partial sums of the two
    # N = model_dim projections (the reference's fp16 model rounds its logits and every sublayer output to fp16 too;
    # smi_text_decoder_set_beam_logits_dtype).  SMI_DEC_LOGITS_F16=0 SMI_DEC_SLAB_F16=0 restore fp32 storage.
    out["storage"] = {"logits": "f16 (tile-major, softmax statistics of the rounded values)", "split_k_partials": "f16",
                      "residual_stream": "f32", "accumulation": "f32"}
    # the same decoder on a bucket twice as large: 2560 rows need a second round of FFN tiles as ONE chain, and run as two
    # independent chains by default (round 4); both timed, same engine, same embeddings
    try:
        emb2 = torch.nn.functional.normalize(torch.randn(2 * n, D, device=dev, generator=g), dim=-1).half() * 0.2
        res = {}
        for chains in (1, 0):
            eng.set_chains(chains)
            eng.generate(emb2, [3, 256047], beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.generate(emb2, [3, 256047], beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
            torch.cuda.synchronize()
            res[chains] = time.perf_counter() - t0
        eng.set_chains(0)
        out["batch_x2"] = {"workload": f"the same decoder, batch {2 * n} ({2 * n * 5} hypothesis rows)",
                           "one_chain_ms_per_step": res[1] * 1e3 / (steps + 1),
                           "default_chains_ms_per_step": res[0] * 1e3 / (steps + 1), "default_chains": 2,
                           "sentences_per_s": 2 * n / res[0]}
        del emb2
    except Exception as e:   # an extra, never the reason the C5 number is lost
        out["batch_x2"] = {"error": repr(e)}
    if cpu:
        from oracle import text_decoder as OD

        cores = cpu_threads()
        ocfg = OD.OracleTextDecoderConfig()
        params = {k: v.float().cpu() for k, v in sd.items()}
        del sd
        # Parity of the BENCHMARKED shape: one more untimed call on ALL n embeddings -- the same 1 280
        # hypothesis rows, hence the same engines, split-K part counts and fp16 storage as the timed call -- and sentences
        # spread over its row tiles (rows 0, 255|256|260, 635, 1 020|1 024, 1 275..1 279) against the oracle's incremental
        # beam search on exactly those embeddings.  The same oracle run is the CPU baseline's timed sample.
        steps_cpu = 12
        picks = sorted({0, n // 5, n // 5 + 1, n // 2 - 1, (4 * n) // 5, n - 1})
        n_cpu = len(picks)
        e_cpu = emb[picks].float().cpu()
        kw = dict(beam_size=5, min_gen_len=steps_cpu, max_gen_len=(0, steps_cpu))
        t0 = time.perf_counter()
        om = []   # the ORACLE's own decision margins: the only thing that may excuse a token mismatch
        ref = OD.beam_search_incremental(params, ocfg, e_cpu, [3, 256047], margins_out=om, **kw)
        ct = time.perf_counter() - t0
        toks, lens, scores = eng.generate(emb, [3, 256047], **kw)
        margins = eng.last_margins(n).cpu()
        toks, lens, scores = toks.cpu(), lens.cpu(), scores.cpu()
        lg = OD.decoder_logits(params, ocfg, e_cpu[:1], torch.tensor([[3, 256047]]))
        eps = 1e-3 * float(lg.max() - lg.min())
        same, mism, dscore, dmargin = 0, [], 0.0, 0.0
        for j, i in enumerate(picks):
            seq = toks[i, 0, : int(lens[i, 0])].tolist()
            want = ref[j][0].seq.tolist()
            if om[j][2] < 1e30 and float(margins[i, 0]) < 1e30:   # [2]: without the forced-EOS step at the cap (engine: final margin)
                dmargin = max(dmargin, abs(float(margins[i, 0]) - om[j][2]))
            if seq == want:
                same += 1
                dscore = max(dscore, abs(float(scores[i, 0]) - ref[j][0].score))
            else:
                mism.append({"sentence": i, "oracle_margins": [float(m) for m in om[j][:2]], "engine_margins": [float(m) for m in margins[i]],
                             "first_diff_at": next((t for t, (a, b) in enumerate(zip(seq, want)) if a != b), min(len(seq), len(want)))})
        out["parity"] = {"call": f"generate() on all {n} embeddings ({n * 5} hypothesis rows: the timed call's engines and storage), "
                                 f"beam 5, {steps_cpu + 1} decode steps",
                         "sentences_checked": picks, "best_hypotheses_token_identical_to_oracle": f"{same}/{n_cpu}",
                         "max_abs_score_diff_of_identical": dscore, "mismatches": mism,
                         "near_tie_eps": eps,
                         "max_abs_diff_engine_vs_oracle_decision_margin": dmargin,
                         "all_mismatches_are_near_ties_by_the_oracle": all(min(m["oracle_margins"]) < eps for m in mism)}
        out["cpu_baseline"] = {"value": n_cpu * (steps_cpu + 1) / ct, "unit": "tokens/s", "cores": cores, "kind": "port",
                               "sample": f"{n_cpu} sentences of the batch x beam 5 x {steps_cpu + 1} decode steps, oracle incremental beam "
                                         f"search (K/V cache, fp32, same weights), one run of {ct:.1f} s",
                               "best_hypotheses_token_identical_to_gpu": f"{same}/{n_cpu}"}
        out["speedup_vs_cpu_tokens_per_s"] = out["tokens_per_s"] / out["cpu_baseline"]["value"]
    return out


def e2e_leg(model, dev, n=16384, batch_size=1024):
    """SURVEY 8 row f1 in the driver line: `TextToEmbeddingModelPipeline.predict()` on `n` synthetic strings -- SentencePiece
    tokenisation, the native host input path (length sort, bucketing, collation, pinned staging), H2D copies and the encoder,
    i.e. what the reference's predict() does per call (sonar/inference_pipelines/text.py:221-268).  There is no real NLLB
    SentencePiece model on the box (no network), so a synthetic 32 k-piece unigram model is trained in-process on the same
    synthetic corpus (tools/bench_e2e.py does the same); `$SONAR_CHECKPOINT_DIR/sentencepiece.source.256000.model` is used when
    it exists.  Kernel-only headline and this number share the model object."""
    import random
    import tempfile

    import sentencepiece as spm
    import torch

    from sonar_amd.inference_pipelines.text import TextToEmbeddingModelPipeline
    from sonar_amd.tokenizer import NllbTokenizer

    rnd = random.Random(0)
    syll = ["ka", "lo", "mi", "ten", "sur", "pa", "ri", "vo", "da", "ne", "shi", "bu", "tra", "el", "on", "qu", "ix", "za"]
    words = ["".join(rnd.choice(syll) for _ in range(rnd.randint(1, 4))) for _ in range(30000)]
    cum, acc = [], 0.0
    for i in range(len(words)):   # Zipf-like word frequencies
        acc += 1.0 / (i + 1) ** 0.9
        cum.append(acc)
    texts = [" ".join(rnd.choices(words, cum_weights=cum, k=rnd.randint(20, 120))) for _ in range(n)]
    real = os.path.join(os.environ.get("SONAR_CHECKPOINT_DIR", ""), "sentencepiece.source.256000.model")
    with tempfile.TemporaryDirectory() as tmp:
        if os.path.exists(real):
            spm_path, spm_kind = real, "the released NLLB SentencePiece model"
        else:
            with open(os.path.join(tmp, "corpus.txt"), "w") as fh:
                fh.write("\n".join(texts[:12000]))
            spm.SentencePieceTrainer.train(input=os.path.join(tmp, "corpus.txt"), model_prefix=os.path.join(tmp, "e2e"),
                                           vocab_size=32000, model_type="unigram", hard_vocab_limit=False, minloglevel=2)
            spm_path, spm_kind = os.path.join(tmp, "e2e.model"), "synthetic 32k-piece unigram SentencePiece model trained in-process"
        tok = NllbTokenizer(spm_path)
        pipe = TextToEmbeddingModelPipeline(model, tok, device=dev)
        enc = tok.create_encoder(lang="eng_Latn")
        lens = [min(len(t), SEQ * 4) for t in enc.encode_batch(texts)]       # for the statistics only (untimed)
        pipe.predict(texts[:2 * batch_size], source_lang="eng_Latn", batch_size=batch_size)   # warm-up
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            out = pipe.predict(texts, source_lang="eng_Latn", batch_size=batch_size)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        enc.encode_batch(texts)
        tok_t = time.perf_counter() - t0
    dt = min(ts)
    # predict() sorts by CHARACTER length and cuts buckets of `batch_size`; the padded [b, max_len] id matrix is what crosses PCIe
    # (the engine packs the tokens on the device: padding costs H2D bytes, not MFMA time)
    order = sorted(range(n), key=lambda i: len(texts[i]))
    padded = 0
    for b in range(0, n, batch_size):
        chunk = [lens[i] for i in order[b:b + batch_size]]
        padded += max(chunk) * len(chunk)
    ntok = sum(lens)
    return {"workload": f"TextToEmbeddingModelPipeline.predict() on {n} synthetic strings (20-120 Zipf words each), batch_size "
                        f"{batch_size}, eng_Latn: tokenisation + native host input path + H2D + encoder + un-sort, strings in, "
                        f"[n, 1024] fp16 embeddings on the device out",
            "tokenizer": spm_kind, "sentences_per_s": n / dt, "tokens_per_s": ntok / dt, "ms": dt * 1e3,
            "runs_s": [round(t, 3) for t in ts], "tokens_per_sentence": ntok / n,
            "pad_waste_pct_of_h2d_ids": 100.0 * (1 - ntok / padded),
            "tokenise_only_sentences_per_s": n / tok_t,
            "finite": bool(torch.isfinite(out).all()), "out_shape": list(out.shape)}


def speech_leg(dev, n=64, cpu=True):
    """BASELINE configs[3]: sonar_speech_encoder_eng, 64 clips x 10 s @ 16 kHz (fbank + conformer + pooler); CPU baseline:
    the oracle (Kaldi fbank + 24 conformer blocks + pooler, fp32, same weights) on 2 of the clips."""
    import torch

    from sonar_amd.speech_encoder import SpeechEncoderEngine, get_speech_encoder_config, waveforms_to_fbank_batch
    from tools.synth import speech_encoder_state_dict

    sd = speech_encoder_state_dict(dev)
    eng = SpeechEncoderEngine(get_speech_encoder_config("english"), sd, device=dev)
    g = torch.Generator(device=dev).manual_seed(4)
    wavs = torch.rand(n, 160000, device=dev, generator=g) * 2 - 1

    def run():
        feats, _ = waveforms_to_fbank_batch(list(wavs))     # one launch for the batch, as predict() does
        return eng.forward(feats, None, torch.float16)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        emb = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # flops of one forward (matrix products only; T = 499 conformer frames per 10 s clip, d = 1024, F = 4096, 24 blocks):
    # per frame and block 2 x 16 d F / 2 (two macaron FFNs) + 8 d^2 (q|k|v|out) + 2 d^2 (relative-position projection) +
    # 6 d^2 (pointwise conv 1, GLU-doubled, and 2) + attention 4 T d (+ 4 T d relative scores); frame stacking GEMM and the
    # 3-layer pooler are < 1 %
    frames = 499
    d_, f_ = 1024, 4096
    flop_frame_block = 2 * (4 * d_ * f_) + 2 * (4 * d_ * d_) + 2 * d_ * d_ + 2 * (3 * d_ * d_) + 8 * frames * d_
    flops = n * frames * 24 * flop_frame_block
    out = {"workload": f"sonar_speech_encoder_eng fp16, {n} clips x 10 s @ 16 kHz, GPU fbank + 24 conformer blocks + pooler",
           "ms": dt * 1e3, "clips_per_s": n / dt, "audio_seconds_per_s": n * 10 / dt,
           "tflop_per_forward": flops / 1e12, "frac_of_mfma_peak": flops / dt / 1e12 / MFMA_PEAK_TFLOPS}
    if cpu:
        from oracle import speech_encoder as OS

        cores = cpu_threads()
        so = OS.OracleSpeechEncoderConfig(model_dim=1024, num_layers=24, num_heads=16, ffn_inner_dim=4096, conv_kernel=31,
                                          pooler_layers=3, pooler_heads=16, pooler_ffn_dim=4096, pooler_vocab=1024)
        params = {k: v.float().cpu() for k, v in sd.items()}
        del sd
        n_cpu = 2
        w_cpu = wavs[:n_cpu].cpu()

        def cpu_run():
            fb = torch.stack([OS.kaldi_fbank(w) for w in w_cpu])
            return OS.speech_encoder_forward(params, so, fb, torch.tensor([fb.shape[1]] * n_cpu))[1]

        t0 = time.perf_counter()
        ref = cpu_run()
        ct = time.perf_counter() - t0
        cos = torch.nn.functional.cosine_similarity(emb[:n_cpu].float().cpu(), ref, dim=-1)
        out["cpu_baseline"] = {"value": n_cpu / ct, "unit": "clips/s", "cores": cores, "kind": "port",
                               "sample": f"{n_cpu} of the clips (10 s each), oracle Kaldi fbank + 24 conformer blocks + pooler, fp32, "
                                         f"same weights, one run of {ct:.1f} s",
                               "max_1_minus_cos_vs_gpu": float((1 - cos).max())}
        out["speedup_vs_cpu"] = out["clips_per_s"] / out["cpu_baseline"]["value"]
    return out


# ---------------------------------------------------------------------- dry-run stubs (CPU, gloo)
class _StubEngine:
    device_bytes = 0

    def set_profiling(self, on):
        pass

    def read_profile(self):
        return {"gemm_ffn1": {"ms": 1.0, "launches": 1}}


class _StubModel:
    """Stands in for SonarTextTransformerEncoderModel in a dry run: right shapes, no arithmetic."""

    def __init__(self, dev):
        self.engine = _StubEngine()
        self.dev = dev

    def __call__(self, batch):
        import torch

        from sonar_amd.text_encoder import SonarEncoderOutput

        n = batch.seqs.shape[0]
        return SonarEncoderOutput(None, torch.ones((n, D), dtype=torch.float16, device=self.dev), batch.padding_mask)


class _StubXsim:
    """normalize_rows / topk_normalized of sonar_amd.xsim with torch on the CPU (dry run only)."""

    @staticmethod
    def padded(n):
        return (n + 255) // 256 * 256

    def normalize_rows(self, t):
        import torch

        out = torch.zeros((self.padded(t.shape[0]), t.shape[1]), dtype=torch.float16)
        out[: t.shape[0]] = torch.nn.functional.normalize(t.float(), dim=-1).half()
        return out

    def topk_normalized(self, xn, nx, yn, ny, k, y_index_offset=0):
        s = xn[:nx].float() @ yn[:ny].float().T
        v, i = s.topk(k, dim=1)
        return v, (i + y_index_offset).int()


def _with_rccl_log(collective: dict, path) -> dict:
    """RCCL's own init lines (NCCL_DEBUG=INFO, subsystem INIT, written to a file): the `nranks` of the communicator."""
    if path and not os.path.exists(path):
        collective["rccl_init_lines"] = "NCCL_DEBUG_FILE was not written by this RCCL build (the per-rank fact sheet stands alone)"
    if path and os.path.exists(path):
        try:
            with open(path, "r", errors="replace") as fh:
                lines = [ln.strip() for ln in fh if "nranks" in ln or "NCCL version" in ln or "RCCL version" in ln]
            collective["rccl_init_lines"] = [ln[-220:] for ln in lines[:4]]
        except OSError:
            pass
    return collective


class _StubBackend:
    """The five-method xsim backend of sonar_amd.distributed on the CPU stub (dry run of --xsim-ring)."""

    def __init__(self, xs):
        self.xs = xs

    def normalize(self, t):
        return self.xs.normalize_rows(t)

    def pad_rows(self, tn, n):
        import torch

        pad = self.xs.padded(n) - tn.shape[0]
        return torch.cat([tn, tn.new_zeros((pad, tn.shape[1]))]) if pad > 0 else tn

    def topk(self, xn, nx, yn, ny, k, y_index_offset=0):
        return self.xs.topk_normalized(xn, nx, yn, ny, k, y_index_offset)

    def merge_topk(self, part_scores, part_idx=None):
        p, n, k = part_scores.shape
        flat = part_scores.permute(1, 0, 2).reshape(n, p * k)
        v, o = flat.topk(k, dim=1)
        return v, (part_idx.permute(1, 0, 2).reshape(n, p * k).gather(1, o) if part_idx is not None else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-xsim", action="store_true")
    ap.add_argument("--xsim-ring", action="store_true",
                    help="N > 1: rotate the Y shards around the ranks under the mining (sonar_amd.distributed, ring=True) "
                         "instead of all-gathering Y first")
    ap.add_argument("--no-extras", action="store_true", help="skip the varlen / C1 / decoder (C5) / speech (C4) legs")
    ap.add_argument("--xsim-n", type=int, default=1 << 20,
                    help="rows of X and of Y IN TOTAL (BASELINE configs[2]: 1M x 1M); both are sharded over the ranks")
    ap.add_argument("--cpu-sentences", type=int, default=32)
    ap.add_argument("--fp32-residual", action="store_true",
                    help="keep the encoder's residual stream in fp32 (default: fp16, as the reference's fp16 model)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from sonar_amd.text_encoder import PaddingMask, SequenceBatch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run
        # (rendezvous on 127.0.0.1, a free port); rank 0's JSON line is the only thing on stdout either way
        import socket
        import subprocess

        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log(f"[launcher] WORLD_SIZE unset and --gpus {args.gpus}: re-launching as {' '.join(cmd)}")
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # SONAR_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, barriers, all-gathers, all-reduces) with whatever
    # world size the launcher gave -- with one rank it runs every RCCL call of the multi-GPU path on a 1-GPU box
    # (tests/test_gpu_rccl.py); the numbers of such a run are not the N = 1 bench line (no CPU baselines, no extra legs).
    use_dist = world > 1 or os.environ.get("SONAR_BENCH_FORCE_DIST") == "1"
    if use_dist and "MASTER_ADDR" not in os.environ:
        import socket

        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        sk.close()
    batch_n, seq = (BATCH, SEQ) if not DRYRUN else (8, 16)
    rccl_log = None
    if DRYRUN:
        dev = torch.device("cpu")
        sync = lambda: None
        if use_dist:
            dist.init_process_group("gloo")
        xs_mod = _StubXsim()
        padded_rows = xs_mod.padded
    else:
        from sonar_amd import xsim as xs_mod

        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible "
                             f"(--gpus {args.gpus} needs that many devices on this node)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
        if use_dist:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if "NCCL_DEBUG" not in os.environ:  # RCCL's own account of the communicator ("... nranks N ... Init COMPLETE")
                rccl_log = f"/tmp/sonar_bench_rccl_{os.getpid()}.log"
                os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE=rccl_log)
            dist.init_process_group("nccl", device_id=dev)
        padded_rows = lambda n: int(xs_mod._lib.load().smi_xsim_padded_rows(n))
    # what the process group actually is (the first SCALE record must prove N ranks over RCCL)
    collective = {"world_size": dist.get_world_size() if use_dist else 1,
                  "backend": (dist.get_backend() if use_dist else None)}
    if not DRYRUN:
        try:
            collective["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            collective["rccl_version"] = None

    if use_dist:  # one fact sheet per rank, gathered on every rank, printed by rank 0
        import socket as _socket

        me = {"rank": rank, "local_rank": local_rank, "host": _socket.gethostname(), "pid": os.getpid()}
        if not DRYRUN:
            props = torch.cuda.get_device_properties(dev)
            me.update(device=str(dev), name=props.name, cus=props.multi_processor_count,
                      hbm_gb=round(props.total_memory / 2**30, 1),
                      pci=getattr(props, "pci_bus_id", None), uuid=str(getattr(props, "uuid", "")))
        facts = [None] * (dist.get_world_size())
        dist.all_gather_object(facts, me)
        collective["ranks"] = facts

    def barrier():
        if use_dist:
            dist.barrier()

    def max_over_ranks(seconds: float) -> float:
        if not use_dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(fn, steps, warmup):
        """`warmup` untimed calls, then exactly `steps` calls between barrier + synchronize pairs; max over ranks."""
        for _ in range(warmup):
            fn()
        sync()
        barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        barrier()
        sync()
        return max_over_ranks(time.perf_counter() - t0)

    t0 = time.time()
    if DRYRUN:
        model, sd_cpu = _StubModel(dev), None
    else:
        from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, get_text_encoder_config

        sd = text_encoder_state_dict(dev)
        model = SonarTextTransformerEncoderModel(get_text_encoder_config("basic"), sd, device=dev, dtype=torch.float16,
                                                 max_tokens_hint=BATCH * SEQ, fp16_residual=not args.fp32_residual)
        want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
        # the CPU baseline runs the SAME weights (fp16-representable values, fp32 arithmetic)
        sd_cpu = {k: v.float().cpu() for k, v in sd.items()} if want_cpu else None
        del sd
        torch.cuda.empty_cache()
    log(f"[rank {rank}] engine ready in {time.time() - t0:.1f}s, {model.engine.device_bytes / 1e9:.2f} GB in HBM")

    # like predict(): batches are queued, the out-of-vocabulary check (one stream synchronisation) runs once, below
    if hasattr(model, "deferred_check"):
        model.deferred_check = True
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    ids = torch.randint(4, 256001, (batch_n, seq), device=dev, generator=g)
    ids[:, 0] = 256047  # __eng_Latn__
    ids[:, -1] = 3      # </s>
    batch = SequenceBatch(ids, None)
    gathered = torch.empty((world * batch_n, D), dtype=torch.float16, device=dev) if use_dist else None

    def step():
        emb = model(batch).sentence_embeddings
        if use_dist:
            dist.all_gather_into_tensor(gathered, emb)  # assemble the embedding matrix over RCCL / xGMI
        return emb

    for _ in range(args.warmup):
        step()
    # Two timed passes of exactly K steps: the HEADLINE pass runs without the engine's per-launch
    # HIP-event profiling (~170 event pairs per step); the second pass, with it, supplies the per-kernel table and the
    # roofline's live launch durations, and is reported beside the headline as `ms_per_step_profiled`.
    elapsed = timed(step, args.steps, 0)
    model.engine.set_profiling(True)
    elapsed_prof = timed(step, args.steps, 0)
    model.engine.set_profiling(False)
    prof = model.engine.read_profile()

    c2_gpu = None
    if not DRYRUN and rank == 0 and world == 1 and not args.no_cpu_baseline:
        c2_gpu = step()[: args.cpu_sentences].float().cpu()   # the rows the CPU oracle will encode (untimed extra call)
        c2_ids_cpu = ids[: args.cpu_sentences].cpu()
    ms_per_step = elapsed / args.steps * 1e3
    ms_per_step_profiled = elapsed_prof / args.steps * 1e3
    value = world * batch_n * args.steps / elapsed
    ffn1 = prof["gemm_ffn1"]
    ffn1_ms = ffn1["ms"] / max(ffn1["launches"], 1)
    achieved = FFN1_FLOPS_PER_LAUNCH / (ffn1_ms * 1e-3) / 1e12 if ffn1_ms > 0 else None
    traffic, traffic_src = None, None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            traffic = tj.get("gemm_ffn1_hbm_bytes_per_launch")
            traffic_src = "profiles/traffic.json (STATIC: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of " + \
                          str(tj.get("state", "r01m")) + ", not measured in this run)"
        except Exception:
            traffic = None
    roofline = {"bound": "mfma", "kernel": "gemm_v2_kernel<EPI_RELU_F16, LayerNorm-fold consumer> (FFN inner projection, M=131072 N=8192 K=1024; "
                                           "4-wave 256x256 engine, csrc/gemm_v2.hip; SMI_G2V2=0: gemm_tn256_kernel<EPI_RELU_F16>)",
                "timed_region": "the second (profiled) K-step pass: HIP events around every launch on the engine's stream",
                "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_PEAK_TFLOPS if achieved else None, "traffic": traffic,
                "traffic_source": traffic_src, "avg_launch_ms": ffn1_ms, "launches": ffn1["launches"],
                "flops_per_launch": FFN1_FLOPS_PER_LAUNCH,
                "note": "since round 3 this kernel also applies the LayerNorm that precedes it (LayerNorm folded into the GEMMs, "
                        "DESIGN.md 3.1: +2.6 % of its time, measured same-box, for 48 LayerNorm launches = 4.5 ms less per step; "
                        "SMI_ENC_LNFOLD=0 restores the separate launches); `achieved` counts the GEMM's flops only"}
    kernels = {k: {"ms_per_step": v["ms"] / args.steps, "launches_per_step": v["launches"] / args.steps} for k, v in prof.items()}
    # HBM-bound kernels of the step: algorithmic bytes per launch (fp16 residual stream and operands,
    # T = 131072 tokens, d = 1024) / mean HIP-event time of the launches inside the timed region
    HBM_PEAK_GBS = 8000.0
    tok = batch_n * seq
    hbm_bytes = {"layernorm": tok * D * 2 * 2,                 # read x, write h
                 "attention": tok * D * 2 * 4,                 # read q|k|v, write ctx
                 "ln_pool": tok * D * 2 + batch_n * D * 2,     # read x, write one vector per sentence (the pooling kernel)
                 "embed": tok * (8 + D * 2 + D * 2)}           # ids + gathered table row + x row (the fp32 position rows are L2 hits)
    hbm_kernels = {}
    ln_prof = prof.get("layernorm")
    if ln_prof and ln_prof["launches"] and ln_prof["launches"] / args.steps < 2:
        # LayerNorm folded into the GEMMs: the only launch left in this slot is the row-statistics pass
        # over the embedding output (reads x once, writes 8 B per row)
        hbm_bytes["layernorm"] = tok * D * 2 + tok * 8
        hbm_kernels["layernorm_note"] = "LayerNorm is folded into the QKV / FFN-inner GEMMs; this slot times the one row-statistics pass"
    for name, nbytes in hbm_bytes.items():
        p_ = prof.get(name)
        if p_ and p_["launches"] and p_["ms"] > 0:
            us = p_["ms"] / p_["launches"] * 1e3
            hbm_kernels[name] = {"bytes_per_launch": nbytes, "avg_launch_us": us, "achieved_GBs": nbytes / us / 1e3,
                                 "frac_of_hbm_peak": nbytes / us / 1e3 / HBM_PEAK_GBS,
                                 "launches_per_step": p_["launches"] / args.steps}
    hbm_kernels["note"] = ("single launches per step (embed, ln_pool, the row-statistics pass) are dominated by ramp-up and "
                           "tail; attention (24 launches per step) is the steady-state figure")

    # ------------------------------------------------------------ C2(b) varlen + C1 (N = 1 only)
    extra = {}
    c1_gpu = None
    if world == 1 and not args.no_extras and not DRYRUN:
        lens = torch.randint(16, SEQ + 1, (BATCH,), generator=torch.Generator().manual_seed(0))
        vids = ids.clone()
        for i, n in enumerate(lens.tolist()):
            vids[i, n - 1] = 3
            vids[i, n:] = 0
        vbatch = SequenceBatch(vids, PaddingMask(lens.to(torch.int32), SEQ))
        vt = timed(lambda: model(vbatch).sentence_embeddings, args.steps, 1)
        vflop = flops_of_lengths(lens.tolist())
        extra["varlen"] = {"workload": "C2(b): 1024 sentences, lengths randint(16,129) seed 0, right-padded to 128 (packed on device)",
                           "tokens": int(lens.sum()), "ms_per_step": vt / args.steps * 1e3,
                           "sentences_per_s": BATCH * args.steps / vt, "tokens_per_s": int(lens.sum()) * args.steps / vt,
                           "frac_of_mfma_peak": vflop * args.steps / vt / 1e12 / MFMA_PEAK_TFLOPS}
        from oracle import text_encoder as O

        c1_ids, c1_lens = O.synthetic_batch(32, 8, 64, V, seed=0)
        c1_batch = SequenceBatch(c1_ids.to(dev), PaddingMask(c1_lens, c1_ids.shape[1]))
        c1_t = timed(lambda: model(c1_batch).sentence_embeddings, 20, 3)
        c1_gpu = model(c1_batch).sentence_embeddings.float().cpu()
        extra["c1"] = {"workload": "BASELINE configs[0] on the GPU engine: 32 sentences, lengths randint(8,65) seed 0, fp16",
                       "tokens": int(c1_lens.sum()), "ms": c1_t / 20 * 1e3, "sentences_per_s": 32 * 20 / c1_t,
                       "storage": "fp16 residual stream (the fp16 model's), fp16 split-K partial sums summed in fp32 (round 4; "
                                  "SMI_ENC_SLAB_F16=0: fp32 partials)"}

        # the reference's DEFAULT call: predict(..., batch_size=5) (sonar/inference_pipelines/text.py:178)
        b5_ids, b5_lens = O.synthetic_batch(5, 8, 64, V, seed=1)
        b5_batch = SequenceBatch(b5_ids.to(dev), PaddingMask(b5_lens, b5_ids.shape[1]))
        b5_t = timed(lambda: model(b5_batch).sentence_embeddings, 50, 5)
        extra["batch5"] = {"workload": "one predict() bucket at the reference's default batch_size=5: 5 sentences, lengths "
                                       "randint(8,65) seed 1, fp16 (text.py:178)",
                           "tokens": int(b5_lens.sum()), "ms": b5_t / 50 * 1e3, "sentences_per_s": 5 * 50 / b5_t}

    if world == 1 and not args.no_extras and not DRYRUN:
        try:
            extra["e2e"] = e2e_leg(model, dev)
        except Exception as e:  # an extra, never the reason the headline is lost
            extra["e2e"] = {"error": repr(e)}

    if hasattr(model, "deferred_check"):
        model.engine.check()  # IndexError if any of the batches above held ids outside the embedding table

    # ------------------------------------------------------------ xsim leg (BASELINE configs[2])
    xs = None
    if not args.no_xsim:
        n_total = args.xsim_n if not DRYRUN else 512 * world
        if n_total % world:
            raise SystemExit(f"--xsim-n {n_total} must divide over {world} ranks")
        nloc = n_total // world
        gx = torch.Generator(device=dev).manual_seed(2 + rank)
        y_local = torch.randn(nloc, D, device=dev, generator=gx, dtype=torch.float32).half()
        # x_i = y_src(i) + noise: the nearest neighbour of every x row is known by construction
        src_local = torch.randint(0, nloc, (nloc,), device=dev, generator=gx)
        x_local = (y_local[src_local].float() + 0.3 * torch.randn(nloc, D, device=dev, generator=gx)).half()
        dense = nloc == padded_rows(nloc)   # equal shards of whole 256-row tiles gather into a dense layout
        yn_all = torch.empty((world * nloc, D), dtype=torch.float16, device=dev) if use_dist and dense else None

        def mine():
            xn = xs_mod.normalize_rows(x_local)
            yn = xs_mod.normalize_rows(y_local)
            if not use_dist:
                return xs_mod.topk_normalized(xn, nloc, yn, nloc, 1)
            if args.xsim_ring:  # Y shards rotate around the ranks under the mining; per-shard lists merged at the end
                from sonar_amd import distributed as sdist

                sdist.force_collectives(world == 1)
                return sdist.sharded_xsim_topk(x_local, y_local, 1, backend=_StubBackend(xs_mod) if DRYRUN else None,
                                               ring=True)
            if dense:
                dist.all_gather_into_tensor(yn_all, yn)  # assemble Y over RCCL / xGMI (2 KB per row)
                return xs_mod.topk_normalized(xn, nloc, yn_all, n_total, 1)
            from sonar_amd import distributed as sdist
            from sonar_amd.distributed import all_gather_rows

            sdist.force_collectives(world == 1)  # a forced one-rank run issues the collectives too
            ya, _ = all_gather_rows(yn[:nloc])
            pad = padded_rows(n_total) - n_total
            if pad:
                ya = torch.cat([ya, ya.new_zeros((pad, D))])
            return xs_mod.topk_normalized(xn, nloc, ya.contiguous(), n_total, 1)

        reps = 3
        xt = timed(mine, reps, 1) / reps
        # correctness of the timed configuration itself: top-1 index == the constructed source row (global index)
        _, top_i = mine()
        hit = (top_i[:nloc, 0].long() == src_local + rank * nloc).sum().to(torch.float64)
        if use_dist:
            dist.all_reduce(hit)
        top1_agree = float(hit.item()) / n_total
        # the HBM-bound half of the leg on its own: L2 normalisation of one side (read fp16, write fp16)
        nt_ = timed(lambda: xs_mod.normalize_rows(y_local), 3, 1) / 3
        pairs = float(n_total) * n_total
        xs = {"workload": f"xsim cosine mining, {n_total} x {n_total} x {D} fp16 (BASELINE configs[2]), top-1, X and Y "
                          f"row-sharded over {world} rank(s), Y all-gathered",
              "pairs_per_s": pairs / xt, "ms": xt * 1e3, "nx_per_gpu": nloc, "nx_total": n_total, "ny_total": n_total,
              "d": D, "k": 1, "tflops": pairs * 2 * D / xt / 1e12,
              "frac_of_mfma_peak": pairs * 2 * D / xt / 1e12 / (MFMA_PEAK_TFLOPS * world),
              "includes": "row normalisation of X and Y, Y all-gather (N>1), top-1 mining + chunk merge",
              "y_exchange": ("ring rotation under the mining (--xsim-ring)" if use_dist and args.xsim_ring else
                             ("all-gather" if use_dist else None)),
              "top1_agreement_with_constructed_neighbours": top1_agree,
              "normalise": {"ms": nt_ * 1e3, "bytes": nloc * D * 4, "achieved_GBs": nloc * D * 4 / nt_ / 1e9,
                            "frac_of_hbm_peak": nloc * D * 4 / nt_ / 1e9 / 8000.0},
              "scaling": "strong (the 1M x 1M problem is fixed, X rows are split over the ranks)"}
        # xsim as LASER defines it (source/xsim.py, margin "ratio", k = 4): forward top-4 (x -> y), backward top-4
        # (y -> x), margin re-scoring of the forward candidates; N = 1 only.  The error is counted against the
        # constructed neighbours (x_i was built from y_src(i)), so it must be 0.
        if world == 1 and not args.no_extras and not DRYRUN:
            def margin_mine():
                xn = xs_mod.normalize_rows(x_local)
                yn = xs_mod.normalize_rows(y_local)
                fs, fi = xs_mod.topk_normalized(xn, nloc, yn, nloc, 4)
                bs, _ = xs_mod.topk_normalized(yn, nloc, xn, nloc, 4)
                return xs_mod.margin_select(fs, fi, bs, "ratio")

            mreps = 2
            mt = timed(margin_mine, mreps, 1) / mreps
            mpred, _ = margin_mine()
            merr = float((mpred.long() != src_local).sum().item()) / nloc
            t4 = timed(lambda: xs_mod.topk_normalized(xs_mod.normalize_rows(x_local), nloc,
                                                      xs_mod.normalize_rows(y_local), nloc, 4), 1, 0)
            xs["margin"] = {"workload": f"xsim, ratio margin, k = 4 (LASER xsim.py): {n_total} x {n_total} x {D} fp16 on one "
                                        "GPU -- forward top-4 + backward top-4 + margin select, normalisation included",
                            "ms": mt * 1e3, "pairs_per_s": pairs / mt, "tflops": 2 * pairs * 2 * D / mt / 1e12,
                            "frac_of_mfma_peak": 2 * pairs * 2 * D / mt / 1e12 / MFMA_PEAK_TFLOPS,
                            "error_rate_vs_constructed_neighbours": merr,
                            "top4_one_direction_ms": t4 * 1e3,
                            "top4_frac_of_mfma_peak": pairs * 2 * D / t4 / 1e12 / MFMA_PEAK_TFLOPS,
                            "note": "pairs_per_s counts each (x, y) pair once; the margin needs the score matrix in both "
                                    "directions, so tflops = 2 x 2 d flop per pair"}
        del x_local, y_local, yn_all, src_local
        if rank == 0 and world == 1 and not args.no_cpu_baseline and not DRYRUN:
            xs["cpu_baseline"] = xsim_cpu_baseline()
            if "margin" in xs:
                xs["margin"]["cpu_baseline"] = xsim_margin_cpu_baseline()

    # ------------------------------------------------- secondary configs (BASELINE C4 / C5), N = 1 only
    if world == 1 and not args.no_extras and not DRYRUN:
        del model
        torch.cuda.empty_cache()
        try:
            extra["decoder"] = decoder_leg(dev, cpu=not args.no_cpu_baseline)
        except Exception as e:  # the headline line must survive a failure here
            extra["decoder"] = {"error": repr(e)}
        try:
            extra["speech"] = speech_leg(dev, cpu=not args.no_cpu_baseline)
        except Exception as e:
            extra["speech"] = {"error": repr(e)}
        model = None

    cb = None
    if sd_cpu is not None:
        model = None
        torch.cuda.empty_cache()
        from oracle import text_encoder as O

        if c1_gpu is None:
            c1_ids, c1_lens = O.synthetic_batch(32, 8, 64, V, seed=0)
        cb, c2_cpu, c1_cpu = cpu_baseline(sd_cpu, c2_ids_cpu, c1_ids, c1_lens)
        if c2_gpu is not None:
            cos2 = torch.nn.functional.cosine_similarity(c2_gpu, c2_cpu, dim=-1)
            cb["c2_max_1_minus_cos_vs_gpu"] = float((1 - cos2).max())
            cb["c2_rows_compared"] = int(c2_gpu.shape[0])
        if c1_gpu is not None:
            cos = torch.nn.functional.cosine_similarity(c1_gpu, c1_cpu, dim=-1)
            extra["c1"]["max_1_minus_cos_vs_cpu_oracle"] = float((1 - cos).max())
            extra["c1"]["cpu_sentences_per_s"] = cb["c1"]["value"]
            extra["c1"]["speedup_vs_cpu"] = extra["c1"]["sentences_per_s"] / cb["c1"]["value"]

    if rank == 0:
        out = {
            "metric": "sentences/sec embedded (seq128 b1024)", "value": value, "unit": "sentences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_step_profiled": ms_per_step_profiled,   # the second K-step pass, per-launch HIP-event profiling on (kernels / roofline)
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic" if not DRYRUN else "dry-run stub",
            "config": {"workload": "text_sonar_basic_encoder fp16, batch 1024 x seq_len 128 per GPU, eng_Latn (BASELINE configs[1])",
                       "global_batch": world * batch_n, "seq_len": seq, "parallelism": f"dp{world}",
                       "weights": "random-init basic arch (24L, d=1024, F=8192, V=256206)"},
            "roofline": roofline, "cpu_baseline": cb,
            "parity": {"tolerance_1_minus_cos": 1e-3,
                       "c2_max_1_minus_cos_vs_cpu_oracle": cb.get("c2_max_1_minus_cos_vs_gpu") if cb else None,
                       "c1_max_1_minus_cos_vs_cpu_oracle": extra.get("c1", {}).get("max_1_minus_cos_vs_cpu_oracle"),
                       "xsim_top1_agreement": xs.get("top1_agreement_with_constructed_neighbours") if xs else None},
            "collective": _with_rccl_log(collective, rccl_log),
            "encoder_tflops": value * FLOPS_PER_SENTENCE / 1e12,
            "encoder_frac_of_mfma_peak": value * FLOPS_PER_SENTENCE / 1e12 / (MFMA_PEAK_TFLOPS * world),
            "kernels": kernels, "hbm_kernels": hbm_kernels, "xsim": xs, **extra,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
