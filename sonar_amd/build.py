"""Build the gfx950 shared library (hipcc, in-tree) -- `python -m sonar_amd.build`.

The library is the product: nothing in this package computes on the CPU when
it is missing (see `_lib.load`).  Objects are rebuilt only when a source or
header is newer than the object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OUT_DIR = ROOT / "lib"
LIB_NAME = "libsonar_mi355.so"
ARCH = "gfx950"
SOURCES = ["api.hip", "gemm.hip", "gemm_v2.hip", "gemm_v2_lone.hip", "rowops.hip", "attention.hip", "xsim.hip", "decoder.hip", "decoder_api.hip", "speech.hip", "speech_api.hip", "host_input.cpp", "host_audio.cpp", "heads.hip", "sampling.hip", "flex.hip", "flex_encoder.hip"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build the gfx950 kernels)")


def _newest_header() -> float:
    hdrs = list(CSRC.glob("*.hpp")) + list((ROOT.parent / "include").glob("*.h"))
    return max(p.stat().st_mtime for p in hdrs)


def build(force: bool = False, verbose: bool = True) -> Path:
    cc = hipcc()
    obj_dir = OUT_DIR / "obj"
    obj_dir.mkdir(parents=True, exist_ok=True)
    hdr_m = _newest_header()
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    extra = os.environ.get("SMI_HIPCC_FLAGS", "").split()

    def compile_one(src: str) -> Path:
        s = CSRC / src
        o = obj_dir / (s.stem + ".o")
        if force or not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, hdr_m):
            cmd = [cc, *flags, *extra, "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    lib = OUT_DIR / LIB_NAME
    if force or not lib.exists() or lib.stat().st_mtime < max(o.stat().st_mtime for o in objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(lib)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
