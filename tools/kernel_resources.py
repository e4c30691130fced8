"""Per-kernel register / spill / LDS summary of one HIP source, offline (no GPU): hipcc -Rpass-analysis=kernel-resource-usage.
usage: python tools/kernel_resources.py sonar_amd/csrc/xsim.hip [name filter] [-D...]"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for ln in out.splitlines():
        m = re.search(r"remark: .*?(Function Name|Name): (\S+)", ln)
        if m:
            cur = m.group(2)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*?): (\d+)", ln)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for name, r in rows.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = dem.split("(")[0]
        if flt and flt not in short:
            continue
        print(f"{short[:80]:80s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):4d} spill {r.get('VGPRs Spill', -1):4d} "
              f"scratch {r.get('ScratchSize [bytes/lane]', -1):5d} occ {r.get('Occupancy [waves/SIMD]', -1)}")


if __name__ == "__main__":
    main()
