"""xsim mining time for one (nx, ny) under the SMI_XSIM_* development knobs (read once per process)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import xsim

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = torch.Generator(device="cuda").manual_seed(2)
y = torch.randn(ny, 1024, device="cuda", generator=g).half()
x = (y[torch.randint(0, ny, (nx,), device="cuda", generator=g)].float() + 0.3 * torch.randn(nx, 1024, device="cuda", generator=g)).half()
xn, yn = xsim.normalize_rows(x), xsim.normalize_rows(y)
xsim.topk_normalized(xn, nx, yn, ny, k)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    s, i = xsim.topk_normalized(xn, nx, yn, ny, k)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
t = min(ts)
print(f"xsim nx={nx} ny={ny} k={k} LL={os.environ.get('SMI_XSIM_LL','1')} TM={os.environ.get('SMI_XSIM_TM','1')}: "
      f"{t*1e3:.1f} ms  {nx*ny/t:.3e} pairs/s  {nx*ny*2048/t/1e12:.0f} TFLOP/s  checksum {int(i.sum())} score-sum {float(s.double().sum()):.6f}", flush=True)
