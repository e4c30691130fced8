import sys, os, torch
sys.path.insert(0, "/root/repo")
from sonar_amd import _lib
from tools.probe_perf import timeit
if os.environ.get("SMI_LIB"):  # a variant build of the library
    from pathlib import Path
    _lib.LIB_PATH = Path(os.environ["SMI_LIB"]).resolve()
lib=_lib.load(); _lib.check(lib.smi_init(0))
st=lambda:int(torch.cuda.current_stream().cuda_stream)
for n in (4096, 8192):
    for tag,flags in (("rm",0),("tm",_lib.SMI_GEMM_IN_TM|_lib.SMI_GEMM_OUT_TM)):
        x=(torch.rand(n,n,device="cuda")*2-1).half(); w=(torch.rand(n,n,device="cuda")*2-1).half()
        out=torch.empty(n,n,device="cuda",dtype=torch.float16)
        ms=timeit(lambda:_lib.check(lib.smi_gemm_tn(0|(2<<8)|flags,x.data_ptr(),w.data_ptr(),None,out.data_ptr(),n,n,n,n,st())),iters=20,warmup=5)
        print(f"square {n}^3 {tag}: {ms:.3f} ms {2*n**3/ms/1e9:.0f} TF/s (uniform random [-1,1))",flush=True)
        xz=torch.zeros_like(x); 
        ms=timeit(lambda:_lib.check(lib.smi_gemm_tn(0|(2<<8)|flags,xz.data_ptr(),xz.data_ptr(),None,out.data_ptr(),n,n,n,n,st())),iters=20,warmup=5)
        print(f"square {n}^3 {tag}: {ms:.3f} ms {2*n**3/ms/1e9:.0f} TF/s (zeros)",flush=True)
