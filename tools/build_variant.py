"""Variant build of the library for same-box A/B runs: recompile the named sources with extra flags, link them with the in-tree
objects of everything else into gpurun_variants/lib<name>.so (git-ignored; travels to the GPU box).  Load it with SMI_LIB=...
usage: python tools/build_variant.py <name> <src.hip>[=<other file to compile in its place>][,<src2.hip>] -DFLAG[=v] ..."""
import os
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd import build as B  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[3:]
    alt = dict((x.split("=", 1) + [""])[:2] for x in sys.argv[2].split(","))   # source -> replacement file ("" = itself)
    srcs = list(alt)
    B.build(verbose=False)
    out = Path(__file__).resolve().parent.parent / "gpurun_variants"
    objd = out / f"obj_{name}"
    objd.mkdir(parents=True, exist_ok=True)
    cc = B.hipcc()
    base = [f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    objs = []
    for s in B.SOURCES:
        stem = Path(s).stem
        if s in srcs:
            o = objd / (stem + ".o")
            src = str(B.CSRC / s)
            if alt[s]:  # compile a COPY inside the variant's own directory: quote-includes look next to the including file
                src = str(objd / s)  # first, and a scratch directory such as /tmp may hold stale headers of the same names
                with open(alt[s], "rb") as fi, open(src, "wb") as fo:
                    fo.write(fi.read())
            subprocess.run([cc, *base, *flags, f"-I{B.CSRC}", "-c", src, "-o", str(o)], check=True)
        else:
            o = B.OUT_DIR / "obj" / (stem + ".o")
        objs.append(str(o))
    lib = out / f"lib{name}.so"
    subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", str(lib)], check=True)
    print(lib)


if __name__ == "__main__":
    main()
