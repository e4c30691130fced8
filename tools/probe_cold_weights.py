"""What do COLD weights cost a lone-tile GEMM?  A small-batch forward streams every layer's weights from HBM exactly once, while
tools/bench_lone.py re-reads the same matrix from L2.  Here every timed launch is preceded by a cache flush (1 GiB written),
and optionally by a read of the weight matrix (weights then sit in the Infinity Cache / partly in the L2s): the difference is
the most a weight prefetch could buy.  usage: python tools/probe_cold_weights.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib


def main():
    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    st = int(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(0)
    flush = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
    shapes = [("qkv", 256, 3072, 1024, 0), ("out", 256, 1024, 1024, 3), ("ffn1", 256, 8192, 1024, 1), ("ffn2 part", 256, 1024, 1024, 3)]
    for label, m, n, k, epi in shapes:
        x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
        bias = torch.randn(n, device="cuda", generator=g)
        out = torch.empty(m, n, device="cuda", dtype=torch.float32 if epi == 3 else torch.float16)
        flags = _lib.SMI_GEMM_IN_TM | (_lib.SMI_GEMM_OUT_TM if epi != 3 else 0)
        call = lambda: _lib.check(lib.smi_gemm_tn(epi | (1 << 8) | flags, x.data_ptr(), w.data_ptr(), bias.data_ptr(),
                                                  out.data_ptr(), m, n, k, n, st))
        res = {}
        for mode in ("hot", "cold", "cold+read"):
            ts = []
            for rep in range(12):
                if mode != "hot":
                    flush.fill_(float(rep))
                if mode == "cold+read":
                    w.view(torch.int32).sum()
                x.add_(0)   # the activations are always fresh from the previous kernel
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                call()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res[mode] = ts[len(ts) // 2]
        print(f"{label:10s} M={m} N={n} K={k}: us per launch (event pair around ONE launch, median of 12): " +
              " | ".join(f"{a} {b:6.2f}" for a, b in res.items()), flush=True)


if __name__ == "__main__":
    main()
