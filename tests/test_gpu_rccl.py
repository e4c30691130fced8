"""The N > 1 code path over RCCL on the GPU box.  One GPU per box means one rank -- but with the collectives FORCED
(distributed.force_collectives() / SONAR_BENCH_FORCE_DIST) every all-gather, all-reduce and barrier of sonar_amd.distributed and
of bench.py's multi-GPU branch is issued through the "nccl" backend on device tensors, and the results must equal the
single-process ones.  (The multi-rank arithmetic -- uneven shards, merges -- is covered by the gloo tests on CPU.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra):
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return p.stdout


def test_distributed_module_over_rccl_with_one_rank():
    out = _run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py")], {})
    line = [l for l in out.splitlines() if l.startswith("RCCL_WORKER ")][-1]
    res = json.loads(line[len("RCCL_WORKER "):])
    print(res)
    assert res["backend"] == "nccl" and res["world"] == 1 and res["rccl"]
    for k, v in res.items():
        if k.endswith("_ok"):
            assert v is True, (k, res)


def test_bench_multi_gpu_branch_over_rccl_with_one_rank():
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline", "--no-extras", "--xsim-n", "65536"], {"SONAR_BENCH_FORCE_DIST": "1"})
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print({k: rec[k] for k in ("value", "n_gpus", "collective")}, rec["xsim"]["top1_agreement_with_constructed_neighbours"])
    assert rec["collective"]["backend"] == "nccl" and rec["collective"]["world_size"] == 1
    assert rec["collective"]["rccl_version"]
    # the per-rank fact sheet and RCCL's own init lines (NCCL_DEBUG=INFO into a file) ride in rank 0's JSON
    assert rec["collective"]["ranks"][0]["rank"] == 0 and rec["collective"]["ranks"][0]["cus"] > 0
    print("RCCL init:", rec["collective"].get("rccl_init_lines"))
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["xsim"]["top1_agreement_with_constructed_neighbours"] == 1.0


def test_bench_xsim_ring_branch_over_rccl_with_one_rank():
    """bench.py's `--xsim-ring` branch (Y shards rotated around the ranks under the mining) through the nccl
    backend: with one forced rank the shard travels once around the one-rank ring -- isend / irecv to self in one RCCL group
    call (sonar_amd/distributed.py) -- and the copy that arrived is mined; top-1 must still be the constructed neighbours."""
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--xsim-ring",
                "--no-cpu-baseline", "--no-extras", "--xsim-n", "32768"], {"SONAR_BENCH_FORCE_DIST": "1"})
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert rec["collective"]["backend"] == "nccl"
    assert rec["xsim"]["y_exchange"].startswith("ring")
    assert rec["xsim"]["top1_agreement_with_constructed_neighbours"] == 1.0
