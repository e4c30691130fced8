"""Development aid: how much of the xsim mining kernel's slice stream is the per-tile fold (needs a -DSMI_XSIM_TRACE build:
SMI_HIPCC_FLAGS=-DSMI_XSIM_TRACE python -m sonar_amd.build --force, or SMI_LIB=<variant .so>)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import _lib, xsim  # noqa: E402


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    ny = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    g = torch.Generator(device="cuda").manual_seed(2)
    y = torch.randn(ny, 1024, device="cuda", generator=g).half()
    x = (y[torch.randint(0, ny, (nx,), device="cuda", generator=g)].float() + 0.3 * torch.randn(nx, 1024, device="cuda", generator=g)).half()
    xn, yn = xsim.normalize_rows(x), xsim.normalize_rows(y)
    for _ in range(2):
        xsim.topk_normalized(xn, nx, yn, ny, k)
    torch.cuda.synchronize()
    raw = C.CDLL(str(_lib.LIB_PATH))
    buf = np.zeros(256 * 2 * 4, dtype=np.uint64)
    assert raw.smi_debug_xsim_trace(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(256, 2, 4).astype(np.float64)
    for grp in (0, 1):
        tot, fold, n, mx = t[:, grp, 0], t[:, grp, 1], t[:, grp, 2], t[:, grp, 3]
        ok = n > 0
        print(f"group {grp}: stream {tot[ok].mean() / 100:.0f} us per workgroup, {n[ok].mean():.0f} folds, fold {fold[ok].sum() / n[ok].sum() / 100:.3f} us each "
              f"(longest {mx[ok].max() / 100:.2f} us) = {100 * fold[ok].sum() / tot[ok].sum():.1f} % of the stream; "
              f"tile period {((tot[ok] - 0) / n[ok]).mean() / 100:.2f} us", flush=True)


if __name__ == "__main__":
    main()
