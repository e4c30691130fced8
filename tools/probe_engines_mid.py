import os, sys
import torch
sys.path.insert(0, "/root/repo")
from sonar_amd import _lib
lib = _lib.load(); _lib.check(lib.smi_init(0))
st = int(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(0)
flush = torch.empty(1 << 27, device="cuda", dtype=torch.float32)
for (m, n, k, epi) in [(1536, 3072, 1024, 0), (1536, 8192, 1024, 1), (1536, 1024, 1024, 3), (1024, 3072, 1024, 0), (1024, 8192, 1024, 1), (768, 3072, 1024, 0), (768, 8192, 1024, 1), (2048, 3072, 1024, 0), (2048, 8192, 1024, 1)]:
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(n, device="cuda", generator=g)
    out = torch.empty(m, n, device="cuda", dtype=torch.float32 if epi == 3 else torch.float16)
    flags = _lib.SMI_GEMM_IN_TM | (_lib.SMI_GEMM_OUT_TM if epi != 3 else 0)
    res = []
    for sel in (0, 1, 2):
        for cold in (0, 1):
            ts = []
            for rep in range(9):
                if cold: flush.fill_(float(rep))
                x.add_(0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.smi_gemm_tn(epi | (sel << 8) | flags, x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), m, n, k, n, st))
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort(); res.append(f"sel{sel}{'c' if cold else 'h'} {ts[len(ts)//2]:6.1f}")
    print(f"M={m} N={n} K={k} epi={epi}: " + " | ".join(res), flush=True)
