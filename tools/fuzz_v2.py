"""Random-shape comparison of the round-6 GEMM kernels with the 8-wave / 128x128 engines (development aid).  Every case draws a
shape the 4-wave engine or the lone units accept (M, N multiples of 256; K a multiple of 128 from 256 up), an epilogue, a raster
and whether a bias is given, runs it with the round-6 kernels on and off and requires
  * without a bias: bit-identical outputs (same MFMA shape, same K order, one rounding) -- against the 8-wave 256x256 engine,
    which the comparison leg forces: the 128x128 family's k-sliced units sum K in another fp32 association,
  * with a bias: agreement within one rounding of the fp16 result (the engines add the bias at different ends of the K sum),
  * for the residual epilogues the same on the read-modify-written stream, for split-K the same per slab,
  * for the logits GEMM with tile statistics: bit-identical logits, statistics within fp32 rounding.
usage: python tools/fuzz_v2.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd = random.Random(seed)
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    st = lambda: int(torch.cuda.current_stream().cuda_stream)
    tm = _lib.SMI_GEMM_IN_TM | _lib.SMI_GEMM_OUT_TM
    g = torch.Generator(device="cuda").manual_seed(seed)

    def pack(a):
        dst = torch.empty(a.numel(), device="cuda", dtype=torch.float16)
        _lib.check(lib.smi_pack_tile_major(a.contiguous().data_ptr(), dst.data_ptr(), a.shape[0], a.shape[1], 0, st()))
        return dst

    bad = 0
    kinds = {"gemm": 0, "splitk": 0, "stats": 0}
    for case in range(cases):
        kind = rnd.choices(["gemm", "splitk", "stats"], [6, 2, 2])[0]
        m = 256 * rnd.randint(1, 12)
        k = 128 * rnd.randint(2, 24)
        n = 256 * rnd.randint(1, 16)
        raster = rnd.choice([0, 2])
        x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
        xt, wt = pack(x), pack(w)
        bias = torch.randn(n, device="cuda", generator=g) if rnd.random() < 0.6 else None
        bp = bias.data_ptr() if bias is not None else None
        desc = f"{kind} m={m} n={n} k={k} raster={raster} bias={bias is not None}"
        outs = {}
        if kind == "gemm":
            epi = rnd.choice([0, 1, 5, 8, 9])
            desc += f" epi={epi}"
            base = torch.randn(m * n, device="cuda", generator=g).half() if epi in (8, 9) else None
            for on in (0, 1):
                # off: the 8-wave 256x256 engine, forced (left to itself a launch of few tiles takes the 128x128 family, whose
                # k-sliced units sum K in another fp32 association: ~0.1 % of the fp16 outputs one ulp from ANY 256x256 engine)
                with _lib.tuning(G2V2=on, G2V2_MIN=1, DEC_M160=2 if on else 0, G2_RASTER=raster):
                    out = base.clone() if base is not None else torch.full((m * n,), float("nan"), device="cuda", dtype=torch.float16)
                    _lib.check(lib.smi_gemm_tn(epi | tm | (0 if on else 2 << 8), xt.data_ptr(), wt.data_ptr(), bp, out.data_ptr(), m, n, k, n, st()))
                    outs[on] = (out,)
        elif kind == "splitk":
            ks = rnd.choice([2, 4, 8])
            if (k // 32) % ks or k // ks < 256:
                continue
            desc += f" ks={ks}"
            for on in (0, 1):
                with _lib.tuning(G2V2=on, DEC_M160=2 if on else 0):
                    parts = torch.full((ks, m, n), float("nan"), device="cuda", dtype=torch.float16)
                    rc = lib.smi_gemm_tn_splitk(xt.data_ptr(), wt.data_ptr(), bp, parts.data_ptr(), m, n, k, ks, 1, _lib.SMI_F16, st())
                    if rc != 0:
                        break
                    outs[on] = (parts,)
            if len(outs) < 2:
                continue
        else:
            n = 256 * rnd.randint(8, 96)
            w = (torch.randn(n, k, device="cuda", generator=g) * 0.1).half()
            wt = pack(w)
            valid = n - rnd.choice([0, 1, 50, 255])
            scale = rnd.choice([1.0, 0.7, 2.5])
            desc = f"stats m={m} n={n} k={k} valid={valid} scale={scale}"
            for on in (0, 1):
                with _lib.tuning(G2V2=on, G2V2_MIN=1):
                    out = torch.full((m * n,), float("nan"), device="cuda", dtype=torch.float16)
                    tmax = torch.full((n // 256, m), float("nan"), device="cuda")
                    tsum = torch.full((n // 256, m), float("nan"), device="cuda")
                    _lib.check(lib.smi_gemm_tn_tile_stats(xt.data_ptr(), wt.data_ptr(), out.data_ptr(), m, n, k, scale, valid,
                                                          tmax.data_ptr(), tsum.data_ptr(), st()))
                    outs[on] = (out, tmax, tsum)
        torch.cuda.synchronize()
        kinds[kind] += 1
        a, b = outs[0], outs[1]
        ok = all(torch.isfinite(t.float()).all().item() for t in b)
        if kind == "stats":
            ok = ok and torch.equal(a[0], b[0])
            ok = ok and (a[1] - b[1]).abs().max().item() <= 1e-5 * max(a[1].abs().max().item(), 1.0)
            ok = ok and ((a[2] - b[2]).abs() / a[2]).max().item() <= 2e-5
        elif bias is None and kind == "gemm":
            ok = ok and torch.equal(a[0], b[0])   # the 4-wave engine and the lone units against the 8-wave 256x256 engine: bit for bit
        else:
            fa, fb = a[0].float(), b[0].float()
            tol = 2e-3 * max(fa.abs().max().item(), 1.0)
            ok = ok and (fa - fb).abs().max().item() <= tol and (fa != fb).float().mean().item() <= 0.03
        if not ok:
            bad += 1
            print("MISMATCH", desc, flush=True)
        elif case % 20 == 0:
            print("ok", desc, flush=True)
    print(f"cases by kind: {kinds}; mismatches: {bad}")
    print("FUZZ", "FAILED" if bad else "OK", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
