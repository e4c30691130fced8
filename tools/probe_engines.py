"""Development aid: 128x128 vs 256x256 tile engine on small-M (decode / small-batch) GEMM shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib
from tools.probe_perf import timeit

lib = _lib.load(); _lib.check(lib.smi_init(0))
st = lambda: int(torch.cuda.current_stream().cuda_stream)
for k in (1024,):
    for n in (1024, 3072, 8192):
        for m in (256, 512, 768, 1024, 1280, 2048, 4096):
            x = (torch.randn(m, k, device="cuda") * 0.5).half(); w = (torch.randn(n, k, device="cuda") * 0.03).half()
            b = torch.randn(n, device="cuda"); out = torch.empty(m, n, device="cuda", dtype=torch.float16)
            t = {}
            for sel in (1, 2):
                t[sel] = timeit(lambda: _lib.check(lib.smi_gemm_tn(0 | (sel << 8), x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, n, st())), iters=20, warmup=5)
            tiles = (m // 256) * (n // 256)
            print(f"M={m:5d} N={n:5d} K={k}: 128-engine {t[1]*1e3:7.1f} us, 256-engine {t[2]*1e3:7.1f} us  ({tiles} 256-tiles)  -> {'256' if t[2] < t[1] else '128'}", flush=True)
