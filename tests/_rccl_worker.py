"""Worker of tests/test_gpu_rccl.py: ONE rank on cuda:0 with the "nccl" (= RCCL) backend and distributed.force_collectives(),
so that every collective of sonar_amd.distributed's N > 1 path is issued through RCCL on the GPU box."""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sk.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from sonar_amd import distributed as D
    from sonar_amd import xsim

    D.force_collectives(True)

    res = {"backend": dist.get_backend(), "world": dist.get_world_size(),
           "rccl": ".".join(str(v) for v in torch.cuda.nccl.version())}
    g = torch.Generator(device=dev).manual_seed(0)
    # all-gather of an embedding shard (device tensors, fp16)
    t = torch.randn(37, 1024, device=dev, generator=g).half()
    got, counts = D.all_gather_rows(t)
    res["gather_ok"] = bool(torch.equal(got, t)) and counts == [37]
    # sharded encode: token-balanced deal + all-gather + restore the input order
    texts = ["x" * n for n in (5, 1, 9, 3, 3, 7, 2)]
    enc = lambda batch: torch.tensor([[float(len(s))] for s in batch], device=dev)
    res["encode_ok"] = D.sharded_encode(enc, texts)[:, 0].tolist() == [5.0, 1.0, 9.0, 3.0, 3.0, 7.0, 2.0]
    # sharded mining against the single-process entry points on the same rows (uneven, not a multiple of 256)
    n = 3000
    y = torch.randn(n, 1024, device=dev, generator=g).half()
    x = (y.float() + 0.4 * torch.randn(n, 1024, device=dev, generator=g)).half()
    s1, i1 = xsim.topk(x, y, 4)
    s2, i2 = D.sharded_xsim_topk(x, y, 4)
    res["topk_ok"] = bool(torch.equal(i1, i2[:n])) and bool(torch.equal(s1, s2[:n]))
    s3, i3 = D.sharded_xsim_topk(x, y, 4, ring=True)   # ring rotation of the Y shards (one rank: its own shard only)
    res["ring_ok"] = bool(torch.equal(i1, i3[:n])) and bool(torch.equal(s1, s3[:n]))
    for margin in ("cosine", "ratio", "distance"):
        e1, p1 = xsim.xsim_error(x, y, margin)
        e2, p2 = D.sharded_xsim_error(x, y, margin)
        res[f"error_{margin}_ok"] = e1 == e2 and bool(torch.equal(p1.int().cpu(), p2.int().cpu()))
        res[f"error_{margin}"] = e2
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORKER " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
