"""Same-box alternating timing of the 8-wave and the 4-wave 256x256 engines on the encoder's tile-major fp16 GEMM shapes
(development aid).  usage: python tools/probe_v2.py [rounds]   (SMI_LIB=<variant .so> for probe builds)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib  # noqa: E402


def main():
    if os.environ.get("SMI_LIB"):
        from pathlib import Path
        _lib.LIB_PATH = Path(os.environ["SMI_LIB"]).resolve()
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    st = lambda: int(torch.cuda.current_stream().cuda_stream)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    M = 131072
    flags = _lib.SMI_GEMM_IN_TM | _lib.SMI_GEMM_OUT_TM
    for (n, k, epi, name) in [(8192, 1024, 1, "ffn_inner"), (3072, 1024, 0, "qkv"), (4096, 1024, 5, "speech_ffn")]:
        x = (torch.rand(M, k, device="cuda") * 2 - 1).half()
        w = (torch.rand(n, k, device="cuda") * 2 - 1).half()
        if os.environ.get("ZEROS"):   # data-dependent power: zero operands keep the clock at its maximum
            x.zero_()
            w.zero_()
        b = torch.randn(n, device="cuda")
        out = torch.zeros(M, n, device="cuda", dtype=torch.float16)
        res = {0: [], 1: []}
        for r in range(rounds + 1):
            for v2 in (0, 1):
                with _lib.tuning(G2V2=v2):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(10):
                        _lib.check(lib.smi_gemm_tn(epi | flags, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, n, k, n, st()))
                    e.record()
                    torch.cuda.synchronize()
                    if r:
                        res[v2].append(s.elapsed_time(e) / 10)
        m0, m1 = sorted(res[0])[len(res[0]) // 2], sorted(res[1])[len(res[1]) // 2]
        tf = lambda ms: 2.0 * M * n * k / ms / 1e9
        print(f"{name} M={M} N={n} K={k}: 8-wave {m0:.4f} ms ({tf(m0):.0f} TF)  4-wave {m1:.4f} ms ({tf(m1):.0f} TF)  {100 * (m0 / m1 - 1):+.1f} %", flush=True)
        del x, w, b, out


if __name__ == "__main__":
    main()
