"""CPU, world_size 2, gloo: the N>1 path (sharding + the embedding all-gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sonar_amd.distributed import deal_by_length, shard_range


def test_shard_range_and_deal():
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    a = deal_by_length([5, 1, 9, 3, 3, 7], 2)
    assert sorted(a[0] + a[1]) == list(range(6))
    loads = [sum([5, 1, 9, 3, 3, 7][i] for i in lst) for lst in a]
    assert abs(loads[0] - loads[1]) <= 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sonar_amd.distributed import all_gather_rows, sharded_encode

        # uneven all-gather
        t = torch.full((rank + 2, 3), float(rank))
        g, counts = all_gather_rows(t)
        assert counts == [2, 3] and g.shape == (5, 3)
        assert torch.equal(g[:2], torch.zeros(2, 3)) and torch.equal(g[2:], torch.ones(3, 3))
        # even all-gather keeps the dense fast path
        g2, c2 = all_gather_rows(torch.full((4, 2), float(rank)))
        assert c2 == [4, 4] and g2.shape == (8, 2)
        # sharded encode restores the input order on every rank
        texts = ["a" * n for n in (5, 1, 9, 3, 3, 7, 2)]
        calls = []

        def encode(batch):
            calls.append(len(batch))
            return torch.tensor([[float(len(s)), float(rank)] for s in batch]).reshape(-1, 2)

        out = sharded_encode(encode, texts)
        assert out[:, 0].tolist() == [5.0, 1.0, 9.0, 3.0, 3.0, 7.0, 2.0]
        assert set(out[:, 1].tolist()) == {0.0, 1.0}  # both ranks contributed
        assert calls and calls[0] < len(texts)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_all_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
