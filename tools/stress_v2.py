"""Repeat-launch stress of the round-6 GEMM kernels (development aid): every kernel variant is launched ITER times on the same
operands while a second stream keeps the chip busy with an unrelated GEMM (so that DMA / store latencies vary from launch to
launch), and every output must be bit-identical to the first launch's.  The counted vmcnt waits, the cross-tile operand stream and
the deferred row-sum / statistics stores are where a latent race would show as a rare differing launch.
usage: python tools/stress_v2.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    main_stream = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    st = lambda s=None: int((s or main_stream).cuda_stream)
    tm = _lib.SMI_GEMM_IN_TM | _lib.SMI_GEMM_OUT_TM

    def to_tile_major(a):
        dst = torch.empty(a.numel(), device="cuda", dtype=torch.float16)
        _lib.check(lib.smi_pack_tile_major(a.contiguous().data_ptr(), dst.data_ptr(), a.shape[0], a.shape[1], 0, st()))
        torch.cuda.synchronize()
        return dst

    g = torch.Generator(device="cuda").manual_seed(7)
    # the disturbance: a mid-size GEMM on the other stream, re-launched whenever it has finished
    dx = to_tile_major((torch.randn(4096, 1024, device="cuda", generator=g) * 0.5).half())
    dw = to_tile_major((torch.randn(4096, 1024, device="cuda", generator=g) * 0.05).half())
    dout = torch.empty(4096 * 4096, device="cuda", dtype=torch.float16)
    devt = torch.cuda.Event()

    def disturb():
        if devt.query():
            with torch.cuda.stream(side):
                with _lib.tuning(G2V2=0):
                    _lib.check(lib.smi_gemm_tn(1 | tm, dx.data_ptr(), dw.data_ptr(), None, dout.data_ptr(), 4096, 4096, 1024, 4096, st(side)))
                devt.record(side)

    bad = 0
    cases = [("v2 relu", 1, 8192, 8192, 1024, {}), ("v2 silu", 5, 6144, 4096, 1024, {}), ("v2 bias K=256", 0, 4096, 3072, 256, {}),
             ("v2 resid", 8, 8192, 1024, 4096, {}), ("v2 resid half", 9, 6144, 1024, 1024, {}),
             ("lone relu 160", 1, 1280, 8192, 1024, {}), ("lone bias 192", 0, 1536, 3072, 1024, {}), ("lone relu 128", 1, 1024, 8192, 1024, {})]
    for name, epi, m, n, k, tune in cases:
        x = to_tile_major((torch.randn(m, k, device="cuda", generator=g) * 0.5).half())
        w = to_tile_major((torch.randn(n, k, device="cuda", generator=g) * 0.05).half())
        b = torch.randn(n, device="cuda", generator=g)
        base = (torch.randn(m * n, device="cuda", generator=g)).half() if epi in (8, 9) else None
        first, diff = None, 0
        for it in range(iters):
            disturb()
            out = base.clone() if base is not None else torch.full((m * n,), float("nan"), device="cuda", dtype=torch.float16)
            with _lib.tuning(G2V2=1, G2V2_MIN=1, **tune):
                _lib.check(lib.smi_gemm_tn(epi | tm, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, n, st()))
            if first is None:
                first = out
            elif not torch.equal(out, first):
                diff += 1
        torch.cuda.synchronize()
        print(f"{name:16s} M={m} N={n} K={k}: {diff} of {iters - 1} launches differ", flush=True)
        bad += diff
    # split-K slabs on lone units and the logits projection with tile statistics
    for name, m, n, k in [("lone slabs 160", 1280, 1024, 8192)]:
        x = to_tile_major((torch.randn(m, k, device="cuda", generator=g) * 0.5).half())
        w = to_tile_major((torch.randn(n, k, device="cuda", generator=g) * 0.05).half())
        b = torch.randn(n, device="cuda", generator=g)
        first, diff = None, 0
        for it in range(iters):
            disturb()
            parts = torch.full((8, m, n), float("nan"), device="cuda", dtype=torch.float16)
            _lib.check(lib.smi_gemm_tn_splitk(x.data_ptr(), w.data_ptr(), b.data_ptr(), parts.data_ptr(), m, n, k, 8, 1, _lib.SMI_F16, st()))
            if first is None:
                first = parts
            elif not torch.equal(parts, first):
                diff += 1
        torch.cuda.synchronize()
        print(f"{name:16s} M={m} N={n} K={k}: {diff} of {iters - 1} launches differ", flush=True)
        bad += diff
    m, n, k, valid = 1280, 65536, 1024, 65536 - 50
    x = to_tile_major((torch.randn(m, k, device="cuda", generator=g) * 0.5).half())
    w = to_tile_major((torch.randn(n, k, device="cuda", generator=g) * 0.1).half())
    first, diff = None, 0
    for it in range(iters):
        disturb()
        out = torch.full((m * n,), float("nan"), device="cuda", dtype=torch.float16)
        tmax = torch.full((n // 256, m), float("nan"), device="cuda")
        tsum = torch.full((n // 256, m), float("nan"), device="cuda")
        with _lib.tuning(G2V2=1, G2V2_MIN=1):
            _lib.check(lib.smi_gemm_tn_tile_stats(x.data_ptr(), w.data_ptr(), out.data_ptr(), m, n, k, 0.7, valid, tmax.data_ptr(),
                                                  tsum.data_ptr(), st()))
        cur = (out, tmax, tsum)
        if first is None:
            first = cur
        elif not all(torch.equal(a, c) for a, c in zip(cur, first)):
            diff += 1
    torch.cuda.synchronize()
    print(f"{'v2 tile stats':16s} M={m} N={n} K={k}: {diff} of {iters - 1} launches differ", flush=True)
    bad += diff
    # the conformer's relative-position attention (LDS ring + fp16 pad, inline-asm LDS reads): the benchmark shape (64 clips of 499
    # frames, 16 heads = 4096 workgroups, three per CU) and a ragged batch, row-major and tile-major q | k | v
    for lens, heads in (([499] * 64, 16), ([37, 499, 128, 300, 1, 260] * 8, 16)):
        d = heads * 64
        t = sum(lens)
        pad = (t + 255) // 256 * 256
        tmx = max(lens)
        qkv = (torch.randn(pad, 3 * d, device="cuda", generator=g) * 1.2).half()
        qkv_t = to_tile_major(qkv)
        rp_rows = (2 * tmx - 1 + 127) // 128 * 128
        rp = (torch.randn(rp_rows, d, device="cuda", generator=g) * 0.8).half()
        ub = torch.randn(d, device="cuda", generator=g) * 0.3
        vb = torch.randn(d, device="cuda", generator=g) * 0.3
        cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
        for tmode in (1, 3):
            first, diff = None, 0
            for it in range(iters):
                disturb()
                ctx = torch.zeros((pad, d), device="cuda", dtype=torch.float16)
                _lib.check(lib.smi_relpos_attention((qkv_t if tmode & 2 else qkv).data_ptr(), cu.data_ptr(), rp.data_ptr(), tmx - 1, rp_rows,
                                                    ub.data_ptr(), vb.data_ptr(), ctx.data_ptr(), len(lens), tmx, d, heads, tmode, st()))
                if first is None:
                    first = ctx
                elif not torch.equal(ctx, first):
                    diff += 1
            torch.cuda.synchronize()
            ok = bool(torch.isfinite(first.float()).all())
            print(f"{'relpos attention':16s} {len(lens)} clips, {t} frames, tile_major={tmode}: {diff} of {iters - 1} launches differ, finite={ok}", flush=True)
            bad += diff + (0 if ok else 1)
    del x, w, out, tmax, tsum, first, cur, dx, dw, dout
    torch.cuda.empty_cache()
    # the full text encoder (LayerNorm-fold consumer / producer kernels, residual stream, row sums): repeated forwards of
    # one batch must be bit-identical
    from tools.synth import text_encoder_state_dict
    from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, get_text_encoder_config
    from sonar_amd.text_encoder import SequenceBatch

    dev = torch.device("cuda:0")
    model = SonarTextTransformerEncoderModel(get_text_encoder_config("basic"), text_encoder_state_dict(dev), device=dev,
                                             dtype=torch.float16, max_tokens_hint=512 * 128, fp16_residual=True)
    ids = torch.randint(4, 256001, (512, 128), device=dev, generator=g)
    ids[:, 0] = 256047
    ids[:, -1] = 3
    batch = SequenceBatch(ids, None)
    first, diff = None, 0
    n_fwd = max(iters // 10, 5)
    for it in range(n_fwd):
        emb = model(batch).sentence_embeddings.clone()
        if first is None:
            first = emb
        elif not torch.equal(emb, first):
            diff += 1
    torch.cuda.synchronize()
    print(f"{'text encoder':16s} 512 x 128 tokens: {diff} of {n_fwd - 1} forwards differ", flush=True)
    bad += diff
    print("STRESS", "FAILED" if bad else "OK", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
