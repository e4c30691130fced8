"""Development aid: per-tile phase breakdown of the persistent 256x256 GEMM kernel.
Build with SMI_HIPCC_FLAGS=-DSMI_GEMM_TRACE python -m sonar_amd.build --force, run on the GPU."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib  # noqa: E402


def main():
    if os.environ.get("SMI_LIB"):   # a variant build of the library
        from pathlib import Path
        _lib.LIB_PATH = Path(os.environ["SMI_LIB"]).resolve()
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    st = lambda: int(torch.cuda.current_stream().cuda_stream)
    M = 131072
    tm = _lib.SMI_GEMM_IN_TM
    for (n, k, epi, name) in [(8192, 1024, 1 | tm | _lib.SMI_GEMM_OUT_TM, "ffn1"), (3072, 1024, 0 | tm, "qkv"),
]:
        x = (torch.randn(M, k, device="cuda") * 0.5).half()
        w = (torch.randn(n, k, device="cuda") * 0.03).half()
        b = torch.randn(n, device="cuda")
        out = torch.zeros(M, n, device="cuda", dtype=torch.float32 if (epi & 0xff) == 2 else torch.float16)
        for _ in range(3):
            _lib.check(lib.smi_gemm_tn(epi, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, n, k, n, st()))
        torch.cuda.synchronize()
        buf = np.zeros(16 * 64 * 8, dtype=np.uint64)
        assert lib.smi_debug_gemm_trace(buf.ctypes.data_as(C.c_void_p)) == 0
        t = buf.reshape(16, 64, 8).astype(np.int64)
        ntile = min(64, (M // 256) * (n // 256) // 256)
        t = t[:, 1:ntile - 1]  # steady-state tiles
        ph = {"begin(drain+barrier)": t[..., 1] - t[..., 0], "k-loop": t[..., 2] - t[..., 1],
              "prefetch-issue": t[..., 3] - t[..., 2], "epilogue": t[..., 4] - t[..., 3],
              "tile": t[..., 4] - t[..., 0]}
        print(f"{name}: M={M} N={n} K={k}: " + ", ".join(f"{k_} {v.mean() / 100:.2f} us" for k_, v in ph.items()), flush=True)
        del x, w, b, out


if __name__ == "__main__":
    main()
