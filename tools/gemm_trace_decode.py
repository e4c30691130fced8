"""Development aid: where a LONE 256x256 unit of the decoder's M = 1280 projections spends its time.
Needs a -DSMI_GEMM_TRACE build (SMI_LIB=<variant .so>).  Prints, per shape, the phase times of thread 0 of workgroups
0..15 (100 MHz wall clock): entry -> fill issued -> fill landed (g2_begin) -> K loop -> epilogue, and the wall time of
the whole launch (HIP events) for comparison: launch wall - in-kernel span = dispatch / ramp / drain outside the trace."""
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
    raw = C.CDLL(str(_lib.LIB_PATH if not os.environ.get("SMI_LIB") else os.environ["SMI_LIB"]))
    st = lambda: int(torch.cuda.current_stream().cuda_stream)
    M = 1280
    tm, otm = _lib.SMI_GEMM_IN_TM, _lib.SMI_GEMM_OUT_TM

    def trace():
        buf = np.zeros(16 * 64 * 8, dtype=np.uint64)
        assert raw.smi_debug_gemm_trace(buf.ctypes.data_as(C.c_void_p)) == 0
        t = buf.reshape(16, 64, 8).astype(np.int64)[:, 0]      # the first (only) unit of workgroups 0..15
        ph = {"entry->fill issued": t[:, 0] - t[:, 5], "fill landed": t[:, 1] - t[:, 0], "k-loop": t[:, 2] - t[:, 1],
              "epilogue": t[:, 4] - t[:, 3], "in-kernel": t[:, 4] - t[:, 5]}
        return ", ".join(f"{k} {v.mean() / 100:.2f}" for k, v in ph.items())

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    # FFN inner: 160 lone tiles, K = 1024, relu, tile-major in/out
    for (n, k, name) in [(8192, 1024, "ffn1 <relu f16, tm in/out> 160 tiles")]:
        x = (torch.randn(M, k, device="cuda") * 0.5).half()
        w = (torch.randn(n, k, device="cuda") * 0.03).half()
        b = torch.randn(n, device="cuda")
        out = torch.zeros(M, n, device="cuda", dtype=torch.float16)
        epi = 1 | (2 << 8) | tm | otm
        f = lambda: _lib.check(lib.smi_gemm_tn(epi, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, n, k, n, st()))
        us = timed(f)
        print(f"{name}: launch wall {us:.1f} us; us: {trace()}", flush=True)
    # FFN output: split-K into fp32 slabs, K = 8192
    k, n = 8192, 1024
    x = (torch.randn(M, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * 0.03).half()
    b = torch.randn(n, device="cuda")
    parts = torch.zeros(16, M, n, device="cuda", dtype=torch.float32)
    for ks in (4, 8, 10, 12):
        f = lambda: raw.smi_debug_gemm_splitk(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                                              C.c_void_p(parts.data_ptr()), M, n, k, ks, 1, C.c_void_p(st()))
        us = timed(f)
        print(f"ffn2 split-K {ks} ({20 * ks} units x {256 // ks} slices, fp32 slabs): launch wall {us:.1f} us; us: {trace()}", flush=True)
    # empty-ish launch for reference: same kernel, K = 32 (one slice)
    x1 = (torch.randn(M, 64, device="cuda") * 0.5).half()
    w1 = (torch.randn(8192, 64, device="cuda") * 0.03).half()
    out = torch.zeros(M, 8192, device="cuda", dtype=torch.float16)
    f = lambda: _lib.check(lib.smi_gemm_tn(1 | (2 << 8) | tm | otm, x1.data_ptr(), w1.data_ptr(), 0, out.data_ptr(), M, 8192, 64, 8192, st()))
    try:
        us = timed(f)
        print(f"two-slice launch (K = 64, 160 tiles): launch wall {us:.1f} us; us: {trace()}", flush=True)
    except Exception as e:
        print("one-slice launch refused:", e)


if __name__ == "__main__":
    main()
