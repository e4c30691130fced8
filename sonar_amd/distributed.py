"""One-process-per-GPU data parallelism for the SONAR hot path (torch.distributed;
backend "nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference has no collectives; its only parallel notion is sharding the input
(`dataset.shard(num_shards=world_size, index=rank)`, huggingface_pipelines/dataset.py:89-90).
Here every rank holds a full engine replica, encodes its share of the sentences with no
communication, and ONE all-gather assembles the embedding matrix.  xsim
shards X by rows and all-gathers the normalised Y once; on 8 fully connected MI355X the
all-gather of 1M x 1024 fp16 (256 MB per rank) is a few ms next to the ~0.3 s of mining.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of n items for `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def deal_by_length(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Token-balanced assignment: walk the items from longest to shortest and give each to the
    currently lightest rank.  Returns per-rank lists of item indices (each kept in input order)."""
    loads = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in sorted(range(len(lengths)), key=lambda j: -lengths[j]):
        r = min(range(world_size), key=lambda q: (loads[q], q))
        out[r].append(i)
        loads[r] += lengths[i]
    for lst in out:
        lst.sort()
    return out


def all_gather_rows(t: torch.Tensor) -> Tuple[torch.Tensor, List[int]]:
    """All-gather a [n_r, d] matrix whose row count differs per rank.
    Returns (concatenation in rank order [sum n_r, d], per-rank row counts)."""
    rank, ws = world()
    if ws == 1:
        return t, [t.shape[0]]
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(counts, n)
    counts_l = [int(c.item()) for c in counts]
    n_max = max(counts_l)
    if n_max == 0:
        return t, counts_l
    pad = t
    if t.shape[0] < n_max:
        pad = torch.cat([t, t.new_zeros((n_max - t.shape[0],) + tuple(t.shape[1:]))])
    buf = t.new_empty((ws * n_max,) + tuple(t.shape[1:]))
    dist.all_gather_into_tensor(buf, pad.contiguous())
    if all(c == n_max for c in counts_l):
        return buf, counts_l
    parts = [buf[r * n_max: r * n_max + c] for r, c in enumerate(counts_l)]
    return torch.cat(parts), counts_l


def sharded_encode(encode_fn: Callable[[List[str]], torch.Tensor], texts: Sequence[str]) -> torch.Tensor:
    """Encode `texts` data-parallel: every rank runs `encode_fn` on its token-balanced share,
    one all-gather assembles the [len(texts), d] matrix in INPUT order on every rank."""
    rank, ws = world()
    if ws == 1:
        return encode_fn(list(texts))
    assignment = deal_by_length([len(t) for t in texts], ws)
    mine = assignment[rank]
    emb = encode_fn([texts[i] for i in mine])
    gathered, counts = all_gather_rows(emb)
    order = torch.tensor([i for lst in assignment for i in lst], dtype=torch.int64, device=gathered.device)
    out = torch.empty_like(gathered)
    out[order] = gathered
    return out


def sharded_xsim_topk(x_local: torch.Tensor, y_local: torch.Tensor, k: int = 1):
    """Rows of X and Y are sharded over ranks (rank order = row order).  Returns, for the
    local X rows, (scores [n_local,k], global Y indices [n_local,k])."""
    from . import xsim

    rank, ws = world()
    xn = xsim.normalize_rows(x_local)
    if ws == 1:
        return xsim.topk_normalized(xn, x_local.shape[0], xsim.normalize_rows(y_local), y_local.shape[0], k)
    # normalise locally (fp16), gather the unpadded rows, then re-pad once
    yn_local = xsim.normalize_rows(y_local)[: y_local.shape[0]]
    yn_all, counts = all_gather_rows(yn_local)
    ny = yn_all.shape[0]
    pad = int(xsim._lib.load().smi_xsim_padded_rows(ny)) - ny
    if pad:
        yn_all = torch.cat([yn_all, yn_all.new_zeros((pad, yn_all.shape[1]))])
    return xsim.topk_normalized(xn, x_local.shape[0], yn_all.contiguous(), ny, k)
