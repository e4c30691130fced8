"""Duo engine (gemm.hip: gemm_duo_kernel) against the 8-wave 256x256 engine on the encoder's K = 1024 shapes:
bit-exact outputs and ms per launch.  Run with SMI_G2_DUO=<n> (0 = both columns are the 8-wave engine)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib  # noqa: E402
from tools.probe_perf import timeit  # noqa: E402


def main():
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    st = lambda: int(torch.cuda.current_stream().cuda_stream)
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    otm = _lib.SMI_GEMM_IN_TM | _lib.SMI_GEMM_OUT_TM
    torch.manual_seed(0)
    for (n, k, epi) in [(3072, 1024, 0), (8192, 1024, 1)]:
        x = (torch.randn(M, k, device="cuda") * 0.5).half()
        w = (torch.randn(n, k, device="cuda") * 0.03).half()
        b = torch.randn(n, device="cuda")
        o_big = torch.zeros(M, n, device="cuda", dtype=torch.float16)
        o_duo = torch.zeros(M, n, device="cuda", dtype=torch.float16)
        big = lambda: _lib.check(lib.smi_gemm_tn(epi | otm | (2 << 8), x.data_ptr(), w.data_ptr(), b.data_ptr(), o_big.data_ptr(), M, n, k, n, st()))
        duo = lambda: _lib.check(lib.smi_gemm_tn(epi | otm, x.data_ptr(), w.data_ptr(), b.data_ptr(), o_duo.data_ptr(), M, n, k, n, st()))
        big(); duo()
        torch.cuda.synchronize()
        same = bool(torch.equal(o_big, o_duo))
        nz = float((o_duo != 0).float().mean())
        ms_b = timeit(big, iters=20)
        ms_d = timeit(duo, iters=20)
        ms_b2 = timeit(big, iters=20)
        ms_d2 = timeit(duo, iters=20)
        tf = lambda ms: 2.0 * M * n * k / ms / 1e9
        print(f"N={n} K={k} epi={epi}: identical={same} nonzero={nz:.3f}  8-wave {ms_b:.4f} / {ms_b2:.4f} ms ({tf(min(ms_b, ms_b2)):.0f} TF/s)   "
              f"auto {ms_d:.4f} / {ms_d2:.4f} ms ({tf(min(ms_d, ms_d2)):.0f} TF/s)", flush=True)
        if not same:
            d = (o_big.float() - o_duo.float()).abs()
            print("   max abs diff", float(d.max()), "mismatching", int((d > 0).sum()))


if __name__ == "__main__":
    main()
