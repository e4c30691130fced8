#!/usr/bin/env python
"""Headline benchmark: sentences/s embedded by text_sonar_basic_encoder
(fp16, batch 1024, seq_len 128 -- BASELINE.json configs[1]) on N MI355X, plus
xsim pairs/s, the roofline of the dominant kernel and the CPU oracle baseline.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (smi_text_encoder_forward: embed -> 24 layers ->
LayerNorm -> mean-pool) over one synthetic 1024 x 128 batch per GPU, inputs resident in
HBM; for N > 1 each rank encodes its own batch (weak scaling, sentences are independent:
SURVEY 8(e)) and the step ends with the RCCL all-gather that assembles the embedding
matrix.  Rank 0 prints ONE JSON line on stdout; everything else goes to stderr.
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


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(n_sent=96):
    """Reference-equivalent op sequence (the oracle, kind 'port') on the host cores."""
    import torch

    from oracle import text_encoder as O

    # measured on the MI355X box's host (2 x EPYC 9575F, 256 hw threads, shared): the torch CPU
    # oracle peaks at 16 threads (6.3 sent/s) and degrades with more (1.4 sent/s at 128)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, avail)))
    cfg = O.OracleTextEncoderConfig()
    t0 = time.time()
    params = O.make_synthetic_params(cfg, seed=1234)
    log(f"[cpu_baseline] built fp32 oracle weights in {time.time() - t0:.1f}s")
    ids, _ = O.synthetic_batch(n_sent, SEQ, SEQ, cfg.vocab_size, seed=0)
    O.text_encoder_forward(params, cfg, ids[:2], None)  # warm-up
    t0 = time.time()
    O.text_encoder_forward(params, cfg, ids, None)
    dt = time.time() - t0
    return {"value": n_sent / dt, "unit": "sentences/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_sent} sentences x {SEQ} tokens, full 24-layer fp32 model, torch CPU oracle, 1 timed pass ({dt:.1f} s)"}


def decoder_leg(dev, n=256, steps=64):
    """BASELINE configs[4]: text_sonar_basic_decoder, beam 5, fp16, batch 256, `steps` forced steps."""
    import torch

    from sonar_amd.text_decoder import TextDecoderEngine, get_text_decoder_config
    from tools.synth import text_decoder_state_dict

    eng = TextDecoderEngine(get_text_decoder_config("basic"), text_decoder_state_dict(dev), device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    emb = torch.nn.functional.normalize(torch.randn(n, D, device=dev, generator=g), dim=-1).half() * 0.2
    # untimed warm-up with the same shapes: the first call allocates the KV cache / logits workspace
    eng.generate(emb, [3, 256047], beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.generate(emb, [3, 256047], beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": f"text_sonar_basic_decoder beam 5 fp16, batch {n}, {steps + 1} decode steps (EOS blocked), eng_Latn prompt",
            "ms": dt * 1e3, "ms_per_step": dt * 1e3 / (steps + 1), "sentences_per_s": n / dt,
            "tokens_per_s": n * (steps + 1) / dt}


def speech_leg(dev, n=64):
    """BASELINE configs[3]: sonar_speech_encoder_eng, 64 clips x 10 s @ 16 kHz (fbank + conformer + pooler)."""
    import torch

    from sonar_amd.speech_encoder import SpeechEncoderEngine, get_speech_encoder_config, waveforms_to_fbank_batch
    from tools.synth import speech_encoder_state_dict

    eng = SpeechEncoderEngine(get_speech_encoder_config("english"), speech_encoder_state_dict(dev), device=dev)
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
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"workload": f"sonar_speech_encoder_eng fp16, {n} clips x 10 s @ 16 kHz, GPU fbank + 24 conformer blocks + pooler",
            "ms": dt * 1e3, "clips_per_s": n / dt, "audio_seconds_per_s": n * 10 / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-xsim", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the decoder (C5) and speech (C4) legs")
    ap.add_argument("--xsim-nx", type=int, default=65536, help="X rows per rank")
    ap.add_argument("--xsim-ny", type=int, default=1 << 20, help="total Y rows (sharded over ranks)")
    ap.add_argument("--cpu-sentences", type=int, default=96)
    ap.add_argument("--fp32-residual", action="store_true",
                    help="keep the encoder's residual stream in fp32 (default: fp16, as the reference's fp16 model)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from sonar_amd import xsim
    from sonar_amd.text_encoder import SequenceBatch, SonarTextTransformerEncoderModel, get_text_encoder_config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    cfg = get_text_encoder_config("basic")
    t0 = time.time()
    sd = text_encoder_state_dict(dev)
    model = SonarTextTransformerEncoderModel(cfg, sd, device=dev, dtype=torch.float16, max_tokens_hint=BATCH * SEQ,
                                             fp16_residual=not args.fp32_residual)
    del sd
    torch.cuda.empty_cache()
    log(f"[rank {rank}] engine ready in {time.time() - t0:.1f}s, {model.engine.device_bytes / 1e9:.2f} GB in HBM")

    g = torch.Generator(device=dev).manual_seed(100 + rank)
    ids = torch.randint(4, 256001, (BATCH, SEQ), device=dev, generator=g)
    ids[:, 0] = 256047  # __eng_Latn__
    ids[:, -1] = 3      # </s>
    batch = SequenceBatch(ids, None)
    gathered = torch.empty((world * BATCH, D), dtype=torch.float16, device=dev) if world > 1 else None

    def step():
        emb = model(batch).sentence_embeddings
        if world > 1:
            dist.all_gather_into_tensor(gathered, emb)
        return emb

    for _ in range(args.warmup):
        step()
    model.engine.set_profiling(True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model.engine.set_profiling(False)
    prof = model.engine.read_profile()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = world * BATCH * args.steps / elapsed
    ffn1 = prof["gemm_ffn1"]
    ffn1_ms = ffn1["ms"] / max(ffn1["launches"], 1)
    achieved = FFN1_FLOPS_PER_LAUNCH / (ffn1_ms * 1e-3) / 1e12 if ffn1_ms > 0 else None
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get("gemm_ffn1_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "mfma", "kernel": "gemm_tn256_kernel<EPI_RELU_F16> (FFN inner projection, M=131072 N=8192 K=1024)",
                "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_PEAK_TFLOPS if achieved else None, "traffic": traffic,
                "avg_launch_ms": ffn1_ms, "launches": ffn1["launches"],
                "flops_per_launch": FFN1_FLOPS_PER_LAUNCH}
    kernels = {k: {"ms_per_step": v["ms"] / args.steps, "launches_per_step": v["launches"] / args.steps} for k, v in prof.items()}

    # ------------------------------------------------------------ xsim leg
    xs = None
    if not args.no_xsim:
        nx = args.xsim_nx
        ny_local = args.xsim_ny // world
        gx = torch.Generator(device=dev).manual_seed(2 + rank)
        y_local = torch.randn(ny_local, D, device=dev, generator=gx, dtype=torch.float32).half()
        x_local = (y_local[torch.randint(0, ny_local, (nx,), device=dev, generator=gx)].float()
                   + 0.3 * torch.randn(nx, D, device=dev, generator=gx)).half()
        yn_all = torch.empty((world * int(xsim._lib.load().smi_xsim_padded_rows(ny_local)), D), dtype=torch.float16, device=dev) if world > 1 else None

        def mine():
            xn = xsim.normalize_rows(x_local)
            yn = xsim.normalize_rows(y_local)
            if world > 1:
                dist.all_gather_into_tensor(yn_all, yn)  # assemble Y over RCCL/xGMI
                # shards are padded to 128 rows each; equal sizes keep the gathered layout dense
                return xsim.topk_normalized(xn, nx, yn_all, yn_all.shape[0], 1)
            return xsim.topk_normalized(xn, nx, yn, ny_local, 1)

        mine()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            mine()
        torch.cuda.synchronize()
        barrier()
        xt = (time.perf_counter() - t1) / reps
        if world > 1:
            t = torch.tensor([xt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            xt = float(t.item())
        ny_total = ny_local * world
        pairs = float(world) * nx * ny_total
        xs = {"pairs_per_s": pairs / xt, "ms": xt * 1e3, "nx_per_gpu": nx, "ny_total": ny_total, "d": D, "k": 1,
              "tflops": pairs * 2 * D / xt / 1e12,
              "frac_of_mfma_peak": pairs * 2 * D / xt / 1e12 / (MFMA_PEAK_TFLOPS * world),
              "includes": "row normalisation, Y all-gather (N>1), top-1 mining"}
        del x_local, y_local, yn_all

    # ------------------------------------------------- secondary configs (BASELINE C4 / C5), N = 1 only
    extra = {}
    if world == 1 and not args.no_extras:
        del model
        torch.cuda.empty_cache()
        try:
            extra["decoder"] = decoder_leg(dev)
        except Exception as e:  # the headline line must survive a failure here
            extra["decoder"] = {"error": repr(e)}
        try:
            extra["speech"] = speech_leg(dev)
        except Exception as e:
            extra["speech"] = {"error": repr(e)}
        model = None

    cb = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        model = None
        torch.cuda.empty_cache()
        cb = cpu_baseline(args.cpu_sentences)

    if rank == 0:
        out = {
            "metric": "sentences/sec embedded (seq128 b1024)", "value": value, "unit": "sentences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "text_sonar_basic_encoder fp16, batch 1024 x seq_len 128 per GPU, eng_Latn (BASELINE configs[1])",
                       "global_batch": world * BATCH, "seq_len": SEQ, "parallelism": f"dp{world}",
                       "weights": "random-init basic arch (24L, d=1024, F=8192, V=256206)"},
            "roofline": roofline, "cpu_baseline": cb,
            "encoder_tflops": value * FLOPS_PER_SENTENCE / 1e12,
            "encoder_frac_of_mfma_peak": value * FLOPS_PER_SENTENCE / 1e12 / (MFMA_PEAK_TFLOPS * world),
            "kernels": kernels, "xsim": xs, **extra,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
