"""profiles/traffic.json (read by bench.py for roofline.traffic) from the PMC summary of tools/gpu_round.sh <tag> pmc.
usage: python tools/make_traffic_json.py <tag>"""
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc = json.load(open(os.path.join(root, "gpurun_out", "pmc_traffic.json")))
# EPI_RELU_F16, tile-major in/out, LayerNorm-fold consumer = FFN inner: the 4-wave engine's kernel since round 6, else the 8-wave one
key = next((k for k in pmc if "gemm_v2_kernelILi1ELb1E" in k), None) or next(k for k in pmc if "gemm_tn256_kernelILi1ELi2E" in k)
e = pmc[key]
out = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 2 --warmup 1 "
              f"--no-cpu-baseline --no-xsim --no-extras` (tools/gpu_round.sh {tag} pmc); FETCH_SIZE doubled (gfx950 note, "
              "MI355X_MICROARCH.md HBM section); KiB->bytes. The counters sit at the L2 -> fabric boundary: Infinity-Cache hits are included.",
    "state": tag,
    "kernel": key,
    "launches": e["launches"],
    "gemm_ffn1_read_bytes_per_launch": e["read_bytes_per_launch"],
    "gemm_ffn1_write_bytes_per_launch": e["write_bytes_per_launch"],
    "gemm_ffn1_hbm_bytes_per_launch": e["read_bytes_per_launch"] + e["write_bytes_per_launch"],
    "algorithmic_bytes_per_launch": 2 * 131072 * 1024 + 2 * 8192 * 1024 + 2 * 131072 * 8192,
}
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
