"""Same-box alternating timing of the lone 128 / 160 / 192 x 256 units (gemm_v2_lone.hip) against the 256-row tiles on the FFN
shapes of a decode step / a small-batch encoder forward, weights rotating over more matrices than the Infinity Cache holds
(development aid).  usage: python tools/probe_lone.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    st = lambda: int(torch.cuda.current_stream().cuda_stream)
    flags = _lib.SMI_GEMM_IN_TM | _lib.SMI_GEMM_OUT_TM
    NW = 20
    for m in (512, 768, 1024, 1280, 1536):
        for (n, k, ks, name) in [(8192, 1024, 1, "ffn_inner"), (1024, 8192, 8, "ffn_out"), (1024, 1024, 2, "attn_out")]:
            x = (torch.rand(m, k, device="cuda") * 2 - 1).half()
            ws = [(torch.rand(n, k, device="cuda") * 2 - 1).half() for _ in range(NW)]
            b = torch.randn(n, device="cuda")
            out = torch.zeros(max(ks, 1) * m * n, device="cuda", dtype=torch.float16)
            res = {0: [], 1: []}
            for r in range(5):
                for on in (0, 1):
                    with _lib.tuning(DEC_M160=on):
                        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        s.record()
                        for i in range(40):
                            w = ws[i % NW]
                            if ks == 1:
                                _lib.check(lib.smi_gemm_tn(1 | flags, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, n, st()))
                            else:
                                _lib.check(lib.smi_gemm_tn_splitk(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, ks, 1,
                                                                  _lib.SMI_F16, st()))
                        e.record()
                        torch.cuda.synchronize()
                        if r:
                            res[on].append(s.elapsed_time(e) / 40 * 1e3)
            a, c = sorted(res[0])[len(res[0]) // 2], sorted(res[1])[len(res[1]) // 2]
            print(f"M={m:5d} {name:9s} N={n} K={k} ks={ks}: 256-row tiles / other engines {a:6.2f} us   lone units {c:6.2f} us   {100 * (a / c - 1):+.1f} %", flush=True)
            del x, ws, b, out


if __name__ == "__main__":
    main()
