"""The N > 1 control flow of bench.py, executed without GPUs: SONAR_BENCH_DRYRUN=1 swaps the engine for
a CPU stub and RCCL for gloo, everything else -- torch.distributed.run launch contract, process-group
set-up from the environment, the all-gathers inside the timed step, barrier + max-over-ranks timing, the
sharded xsim leg, rank 0 printing ONE JSON line -- is the code the driver's 8-GPU run will execute."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,form", [(1, "plain"), (2, "torchrun"), (4, "torchrun"), (2, "plain"), (4, "plain"),
                                        (8, "torchrun"), (8, "plain-ring"), (2, "torchrun-ring")])
def test_bench_control_flow_dry_run(world, form):
    """form "torchrun": the driver's documented N > 1 command; form "plain": `python bench.py --gpus N` with no
    WORLD_SIZE in the environment -- bench.py then launches its own ranks (round-2 verdict: the first SCALE
    record must not die on the launch contract)."""
    env = dict(os.environ, SONAR_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--gpus", str(world), "--steps", "3", "--warmup", "1"]
    ring = form.endswith("-ring")       # --xsim-ring: Y shards rotated around the ranks under the mining
    if ring:
        args.append("--xsim-ring")
        form = form[:-5]
    if form == "plain":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), *args]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 only, one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 3 and out["warmup"] == 1
    assert out["data"] == "dry-run stub" and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["global_batch"] == 8 * world and out["config"]["parallelism"] == f"dp{world}"
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - 8 * world * 3 / (out["ms_per_step"] * 3 / 1e3)) / out["value"] < 1e-6
    assert out["collective"]["world_size"] == world
    assert out["collective"]["backend"] == ("gloo" if world > 1 else None)
    if world > 1:   # one fact sheet per rank (the first real SCALE record must show N distinct ranks)
        assert [f["rank"] for f in out["collective"]["ranks"]] == list(range(world))
        assert len({f["pid"] for f in out["collective"]["ranks"]}) == world
        assert out["xsim"]["y_exchange"].startswith("ring" if ring else "all-gather")
    xs = out["xsim"]
    assert xs["nx_total"] == xs["ny_total"] == 512 * world and xs["nx_per_gpu"] == 512 and xs["pairs_per_s"] > 0
    # the timed xsim configuration checks itself: the constructed neighbour of every row, global indices over ranks
    assert xs["top1_agreement_with_constructed_neighbours"] >= 0.99
    assert out["parity"]["xsim_top1_agreement"] == xs["top1_agreement_with_constructed_neighbours"]
    for key in ("metric", "unit", "roofline", "cpu_baseline", "vs_baseline", "dtype"):
        assert key in out


def test_bench_refuses_a_rank_count_that_contradicts_gpus():
    """Launched under a process group of the wrong size, bench.py stops instead of measuring something else."""
    env = dict(os.environ, SONAR_BENCH_DRYRUN="1", OMP_NUM_THREADS="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
