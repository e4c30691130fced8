"""Per-kernel timing probe on one MI355X (development aid, not the judged bench)."""
import json
import sys
import time

import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib  # noqa: E402


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    if os.environ.get("SMI_LIB"):  # a variant build of the library
        from pathlib import Path
        _lib.LIB_PATH = Path(os.environ["SMI_LIB"]).resolve()
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    st = lambda: int(torch.cuda.current_stream().cuda_stream)
    res = {}
    M = 131072
    tmf, otm = _lib.SMI_GEMM_IN_TM, _lib.SMI_GEMM_IN_TM | _lib.SMI_GEMM_OUT_TM
    # the layouts only change addresses, so the same random buffers serve as row-major or tile-major operands
    for (n, k, epi) in [(3072, 1024, 0), (1024, 1024, 2), (8192, 1024, 1), (1024, 8192, 2),
                        (3072, 1024, 0 | otm), (1024, 1024, 2 | tmf), (8192, 1024, 1 | otm), (1024, 8192, 2 | tmf)]:
        x = (torch.randn(M, k, device="cuda") * 0.5).half()
        w = (torch.randn(n, k, device="cuda") * 0.03).half()
        b = torch.randn(n, device="cuda")
        out = torch.zeros(M, n, device="cuda", dtype=torch.float32 if (epi & 0xff) == 2 else torch.float16)
        ms = timeit(lambda: _lib.check(lib.smi_gemm_tn(epi, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, n, k, n, st())))
        tf = 2.0 * M * n * k / ms / 1e9
        tag = "tm" if epi & tmf else "rm"
        res[f"gemm_{n}x{k}_epi{epi & 0xff}_{tag}"] = {"ms": ms, "TF": tf}
        print(f"gemm {tag} M={M} N={n} K={k} epi={epi & 0xff}: {ms:.3f} ms  {tf:.0f} TF/s", flush=True)
        del x, w, b, out
    if len(sys.argv) > 1 and sys.argv[1] == "gemm":
        return
    # layernorm
    x = torch.randn(M, 1024, device="cuda")
    w = torch.randn(1024, device="cuda"); b = torch.randn(1024, device="cuda")
    h = torch.empty(M, 1024, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: _lib.check(lib.smi_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), 1e-5, h.data_ptr(), M, 1024, 0, st())))
    res["layernorm"] = {"ms": ms, "GBs": M * 1024 * 6 / ms / 1e6}
    print(f"layernorm: {ms:.3f} ms {res['layernorm']['GBs']:.0f} GB/s", flush=True)
    # attention
    qkv = (torch.randn(M, 3072, device="cuda")).half()
    cu = torch.arange(0, M + 1, 128, dtype=torch.int32, device="cuda")
    ctx = torch.empty(M, 1024, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: _lib.check(lib.smi_attention(qkv.data_ptr(), cu.data_ptr(), ctx.data_ptr(), 1024, 128, 1024, 16, 0, st())))
    res["attention"] = {"ms": ms, "GBs": M * 4096 * 2 / ms / 1e6, "TF": 4.0 * 128 * 1024 * M / ms / 1e9}
    print(f"attention: {ms:.3f} ms {res['attention']['GBs']:.0f} GB/s {res['attention']['TF']:.0f} TF/s", flush=True)
    del qkv, ctx, x, h
    # full encoder
    from sonar_amd.text_encoder import get_text_encoder_config, SonarTextTransformerEncoderModel, SequenceBatch
    cfg = get_text_encoder_config("basic")
    sd = {}
    d, f = 1024, 8192
    def rnd(*shape, std=0.02):
        return (torch.randn(*shape, device="cuda") * std).half()
    t0 = time.time()
    sd["encoder_frontend.embed.weight"] = rnd(cfg.vocab_info.size, d)
    sd["layer_norm.weight"] = torch.ones(d, device="cuda"); sd["layer_norm.bias"] = torch.zeros(d, device="cuda")
    for i in range(24):
        p = f"encoder.layers.{i}."
        for nme, shp in [("self_attn.q_proj", (d, d)), ("self_attn.k_proj", (d, d)), ("self_attn.v_proj", (d, d)),
                         ("self_attn.output_proj", (d, d)), ("ffn.inner_proj", (f, d)), ("ffn.output_proj", (d, f))]:
            sd[p + nme + ".weight"] = rnd(*shp); sd[p + nme + ".bias"] = rnd(shp[0]).float()
        for nme in ("self_attn_layer_norm", "ffn_layer_norm"):
            sd[p + nme + ".weight"] = torch.ones(d, device="cuda"); sd[p + nme + ".bias"] = torch.zeros(d, device="cuda")
    model = SonarTextTransformerEncoderModel(cfg, sd, device="cuda:0", dtype=torch.float16, max_tokens_hint=M)
    del sd
    torch.cuda.synchronize()
    print(f"model create {time.time()-t0:.1f}s, device bytes {model.engine.device_bytes/1e9:.2f} GB", flush=True)
    ids = torch.randint(4, 256000, (1024, 128), device="cuda")
    ms = timeit(lambda: model(SequenceBatch(ids, None)), iters=5, warmup=2)
    res["encoder_1024x128"] = {"ms": ms, "sent_per_s": 1024 / ms * 1e3, "TF": 133.59e3 / ms}
    print(f"encoder 1024x128: {ms:.2f} ms  {1024/ms*1e3:.0f} sent/s  {133.59e3/ms:.0f} TF/s", flush=True)
    json.dump(res, open("gpurun_out/probe_perf.json", "w"), indent=1)


if __name__ == "__main__":
    main()
