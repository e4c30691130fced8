"""Lone-tile GEMM probe: the projections of a small encoder batch (M = 256: predict(batch_size=5)) and of a decode step
(M = 1280) through smi_gemm_tn with the 128x128-family engines -- round 3's ring (SMI_LONE=0) against the lone-tile engine
(gemm_lone.hpp, 64x64 units while they are all resident).  Each timing is a chain of `reps` dependent-stream launches
between two HIP events (launch gaps included: that is what a forward pays).
usage: python tools/bench_lone.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    st = int(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [  # (label, M, N, K, epi, tile-major in, tile-major out)
        ("enc256 qkv", 256, 3072, 1024, 0, 1, 1), ("enc256 out(slab)", 256, 1024, 1024, 3, 1, 0),
        ("enc256 ffn1", 256, 8192, 1024, 1, 1, 1), ("enc256 ffn2 K/8(slab)", 256, 1024, 1024, 3, 1, 0),
        ("enc512 qkv", 512, 3072, 1024, 0, 1, 1), ("enc512 ffn1", 512, 8192, 1024, 1, 1, 1),
        ("dec1280 qkv", 1280, 3072, 1024, 0, 0, 0), ("dec1280 out(slab)", 1280, 1024, 1024, 3, 0, 0),
        ("dec256 qkv", 256, 3072, 1024, 0, 0, 0),
    ]
    variants = [("ring(r3)", {"SMI_LONE": "0"}), ("lone 64x64 (where it fits)", {"SMI_LONE": "1"})]
    for label, m, n, k, epi, tm, otm in shapes:
        x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
        bias = torch.randn(n, device="cuda", generator=g)
        out = torch.empty(m, n, device="cuda", dtype=torch.float32 if epi == 3 else torch.float16)
        flags = (_lib.SMI_GEMM_IN_TM if tm else 0) | (_lib.SMI_GEMM_OUT_TM if otm else 0)
        line = []
        for name, env in variants:
            for key in ("SMI_LONE", "SMI_LONE_SHAPE"):
                os.environ.pop(key, None)
            os.environ.update(env)
            call = lambda: _lib.check(lib.smi_gemm_tn(epi | (1 << 8) | flags, x.data_ptr(), w.data_ptr(), bias.data_ptr(),
                                                      out.data_ptr(), m, n, k, n, st))
            for _ in range(10):
                call()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps * 1e3)
            line.append(f"{name} {best:6.2f}")
        print(f"{label:24s} M={m:5d} N={n:5d} K={k:5d} us/launch: " + " | ".join(line), flush=True)
    for key in ("SMI_LONE", "SMI_LONE_SHAPE"):
        os.environ.pop(key, None)


if __name__ == "__main__":
    main()
