"""Development aid: where a lone-tile GEMM launch (gemm_lone.hpp, 64x64 units) spends its time at the small-batch shapes.
Needs a -DSMI_GEMM_TRACE build (SMI_LIB=<variant .so>).  Phase times of thread 0 of workgroups 0..15 (100 MHz wall clock):
entry -> fill issued -> stage 0 landed -> K loop done -> stores issued, next to the wall time of the launch (HIP events around
ONE launch, cache flushed before it or not).  usage: SMI_LIB=gpurun_variants/libtrace.so python tools/lone_trace.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    raw = C.CDLL(os.environ["SMI_LIB"])
    st = int(torch.cuda.current_stream().cuda_stream)
    flush = torch.empty(1 << 27, device="cuda", dtype=torch.float32)
    g = torch.Generator(device="cuda").manual_seed(0)

    def trace():
        buf = np.zeros(16 * 64 * 8, dtype=np.uint64)
        assert raw.smi_debug_gemm_trace(buf.ctypes.data_as(C.c_void_p)) == 0
        t = buf.reshape(16, 64, 8).astype(np.int64)[:, 0]
        ph = {"entry->set up": t[:, 3] - t[:, 5], "fill issue": t[:, 0] - t[:, 3], "stage 0 landed": t[:, 1] - t[:, 0], "k-loop": t[:, 2] - t[:, 1],
              "epilogue": t[:, 4] - t[:, 2], "in-kernel": t[:, 4] - t[:, 5]}
        cyc = buf.reshape(16, 64, 8).astype(np.int64)[:, 1, :4].mean(axis=0)   # shader-clock cycles, summed over the K loop
        return (", ".join(f"{k} {v.mean() / 100:.2f} (max {v.max() / 100:.2f})" for k, v in ph.items()) +
                f"; loop cycles of thread 0: issue reads {cyc[0]:.0f}, issue DMA {cyc[1]:.0f}, issue MFMAs {cyc[2]:.0f}, waits + barrier {cyc[3]:.0f}")

    for label, m, n, k, epi in [("qkv", 256, 3072, 1024, 0), ("out", 256, 1024, 1024, 3), ("ffn1", 256, 8192, 1024, 1)]:
        x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
        bias = torch.randn(n, device="cuda", generator=g)
        out = torch.empty(m, n, device="cuda", dtype=torch.float32 if epi == 3 else torch.float16)
        flags = _lib.SMI_GEMM_IN_TM | (_lib.SMI_GEMM_OUT_TM if epi != 3 else 0)
        for cold in (0, 1):
            ts = []
            for rep in range(7):
                if cold:
                    flush.fill_(float(rep))
                x.add_(0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.smi_gemm_tn(epi | (1 << 8) | flags, x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                           m, n, k, n, st))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            print(f"{label} M={m} N={n} {'cold' if cold else 'hot '}: event pair {ts[len(ts) // 2]:.1f} us; us: {trace()}", flush=True)


if __name__ == "__main__":
    main()
