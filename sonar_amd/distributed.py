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

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


_force_collectives = False


def force_collectives(enable: bool = True) -> None:
    """Test hook: make a SINGLE rank issue every collective of the N > 1 path (world size 1) -- how the RCCL calls are
    executed on a 1-GPU box (tests/test_gpu_rccl.py) and under gloo (tests/test_distributed_cpu.py).  A switch the
    test sets on the module, not an environment variable production control flow would depend on."""
    global _force_collectives
    _force_collectives = bool(enable)


def _collectives(ws: int) -> bool:
    """Whether the exchange steps run: always with more than one rank; with one rank only under force_collectives()
    and an initialised process group."""
    if ws > 1:
        return True
    return _force_collectives and dist.is_available() and dist.is_initialized()


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
    if not _collectives(ws):
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
    if not _collectives(ws):
        return encode_fn(list(texts))
    assignment = deal_by_length([len(t) for t in texts], ws)
    mine = assignment[rank]
    emb = encode_fn([texts[i] for i in mine])
    gathered, counts = all_gather_rows(emb)
    order = torch.tensor([i for lst in assignment for i in lst], dtype=torch.int64, device=gathered.device)
    out = torch.empty_like(gathered)
    out[order] = gathered
    return out


class EngineXsimBackend:
    """The xsim primitives the sharded mining is written against -- here the MI355X engine
    (sonar_amd.xsim).  The CPU multi-process tests substitute a torch stand-in with the same five
    methods so that the collective plumbing (uneven shards, global index offsets, merge, error
    reduction) runs under gloo without a GPU."""

    def normalize(self, t: torch.Tensor) -> torch.Tensor:
        """L2-normalised rows of t (engine layout may pad rows; only the first len(t) are meaningful)."""
        from . import xsim

        return xsim.normalize_rows(t)

    def pad_rows(self, tn: torch.Tensor, n: int) -> torch.Tensor:
        """Re-pad a gathered [n, d] matrix of normalised rows to the engine's row multiple."""
        from . import xsim

        pad = int(xsim._lib.load().smi_xsim_padded_rows(n)) - tn.shape[0]
        if pad > 0:
            tn = torch.cat([tn, tn.new_zeros((pad, tn.shape[1]))])
        return tn.contiguous()

    def topk(self, xn, nx: int, yn, ny: int, k: int, y_index_offset: int = 0):
        from . import xsim

        return xsim.topk_normalized(xn, nx, yn, ny, k, y_index_offset)

    def merge_topk(self, part_scores, part_idx=None):
        from . import xsim

        return xsim.merge_topk(part_scores, part_idx)

    def margin_select(self, fs, fi, bs, margin: str, x_index_offset: int, err_count):
        from . import xsim

        return xsim.margin_select(fs, fi, bs, margin, x_index_offset, err_count)


def _empty_normalized(be, like: torch.Tensor) -> torch.Tensor:
    """[0, d] matrix of the backend's normalised-row dtype (an empty shard in a collective)."""
    return be.normalize(like.new_zeros((1, like.shape[1])))[:0]


def _row_offsets(counts: Sequence[int]) -> List[int]:
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + int(c))
    return offs


def _no_candidates(nx_local: int, k: int, dev):
    """The result of mining `nx_local` rows against NO candidates: [nx_local, k] lists of (-inf, -1), the padding a list with
    fewer than k candidates carries (header of sharded_xsim_topk); [0, k] only for a rank without X rows."""
    return (torch.full((nx_local, k), float("-inf"), dtype=torch.float32, device=dev),
            torch.full((nx_local, k), -1, dtype=torch.int32, device=dev))


def _ring_xsim_topk(be, xn, nx_local: int, y_local: torch.Tensor, k: int):
    """Forward mining with the Y shards ROTATED around the ranks instead of all-gathered: in step s a rank mines its X
    rows against the shard of rank (rank - s) mod ws while that shard travels on to rank + 1 (one isend + one irecv
    per step, posted BEFORE the mining call, so on RCCL the transfer runs on the communicator's stream under the
    mining kernel); the per-shard top-k lists are k-way merged at the end (same total order: score desc, index asc).
    A rank holds two shards instead of all of Y, and no step waits for more than one neighbour."""
    rank, ws = world()
    dev = y_local.device
    d = y_local.shape[1]
    cnt = torch.tensor([y_local.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(ws)]
    dist.all_gather(cnts, cnt)
    counts = [int(c.item()) for c in cnts]
    offs = _row_offsets(counts)
    ny = offs[-1]
    # one fixed-size, zero-padded buffer per shard (the engine's row multiple): what arrives can be mined as it is
    yn_local = be.normalize(y_local)[: y_local.shape[0]] if y_local.shape[0] else _empty_normalized(be, y_local)[:0]
    n_pad = be.pad_rows(yn_local.new_zeros((max(max(counts), 1), d)), max(max(counts), 1)).shape[0]
    cur = yn_local.new_zeros((n_pad, d))
    cur[: yn_local.shape[0]] = yn_local
    nxt = torch.empty_like(cur)
    part_s, part_i = [], []
    if ws == 1 and _force_collectives and dist.get_backend() == "nccl":
        # a forced one-rank run over RCCL (tests/test_gpu_rccl.py): the loop below never posts a transfer, so send the shard
        # once around the one-rank "ring" -- isend + irecv to self in one group call (gloo has no self-send) -- and mine the
        # copy that ARRIVED
        for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, cur, 0), dist.P2POp(dist.irecv, nxt, 0)]):
            r.wait()
        cur, nxt = nxt, cur
    for s in range(ws):
        owner = (rank - s) % ws
        reqs = []
        if s + 1 < ws:  # pass the shard on while it is mined
            ops = [dist.P2POp(dist.isend, cur, (rank + 1) % ws), dist.P2POp(dist.irecv, nxt, (rank - 1) % ws)]
            reqs = dist.batch_isend_irecv(ops)
        if nx_local and counts[owner]:
            ps, pi = be.topk(xn, nx_local, cur, counts[owner], min(k, counts[owner]), offs[owner])
            if ps.shape[1] < k:  # a shard smaller than k: pad its list, the merge drops the padding
                ps = torch.cat([ps, ps.new_full((nx_local, k - ps.shape[1]), float("-inf"))], dim=1)
                pi = torch.cat([pi, pi.new_full((nx_local, k - pi.shape[1]), -1)], dim=1)
            part_s.append(ps)
            part_i.append(pi)
        for r in reqs:
            r.wait()
        cur, nxt = nxt, cur
    if not nx_local or not ny:
        return _no_candidates(nx_local, k, dev)
    if len(part_s) == 1:
        return part_s[0], part_i[0]
    return be.merge_topk(torch.stack(part_s), torch.stack(part_i))


def sharded_xsim_topk(x_local: torch.Tensor, y_local: torch.Tensor, k: int = 1, backend=None, ring: bool = False):
    """Rows of X and Y are sharded over ranks (rank order = row order).  Returns, for the
    local X rows, (scores [n_local,k], GLOBAL Y indices [n_local,k]; -inf / -1 where Y has fewer than k rows).  `ring`: rotate the Y shards around the
    ranks under the mining (see _ring_xsim_topk) instead of all-gathering Y first."""
    be = backend or EngineXsimBackend()
    rank, ws = world()
    nx_local, d = x_local.shape
    xn = be.normalize(x_local) if nx_local else None
    empty = _no_candidates(nx_local, k, x_local.device)   # [n_local, k] of (-inf, -1): the contract holds for an empty Y too
    if ring and _collectives(ws):
        return _ring_xsim_topk(be, xn, nx_local, y_local, k)
    if not _collectives(ws):
        if not nx_local or not y_local.shape[0]:
            return empty
        return be.topk(xn, nx_local, be.normalize(y_local), y_local.shape[0], k)
    # normalise locally (fp16), gather the unpadded rows, then re-pad once
    if y_local.shape[0]:
        yn_local = be.normalize(y_local)[: y_local.shape[0]]
    else:  # an empty shard still takes part in the all-gather
        yn_local = _empty_normalized(be, y_local)
    yn_all, counts = all_gather_rows(yn_local)
    ny = sum(counts)
    if not nx_local or not ny:
        return empty
    return be.topk(xn, nx_local, be.pad_rows(yn_all, ny), ny, k)


def sharded_xsim_error(x_local: torch.Tensor, y_local: torch.Tensor, margin: str = "ratio", k: int = 4,
                       backend=None):
    """xsim error rate of aligned pairs x[i] <-> y[i] whose rows are sharded over ranks in rank order
    (SURVEY 8(e)); every rank returns (global error rate, predicted GLOBAL y index of its local x rows).

    Exchange steps (everything else is local mining):
      1. all-gather of the normalised Y shards (2 KB per row)              -> every rank holds Y;
      2. margin only: each rank mines, for EVERY y, its k best neighbours among the LOCAL x rows; the
         [Ny, k] partial score lists are all-gathered (Ny * k * 4 B per rank, 16 MB at 1 M x k = 4) and
         k-way merged, which gives mean_kNN(y_j) over all of X on every rank;
      3. one scalar all-reduce of the per-rank error counts."""
    be = backend or EngineXsimBackend()
    rank, ws = world()
    nx_local, ny_local = x_local.shape[0], y_local.shape[0]
    dev = x_local.device
    d = x_local.shape[1]
    # an empty shard is legal (fewer rows than ranks): it takes part in every collective with zero rows
    xn = be.normalize(x_local) if nx_local else _empty_normalized(be, x_local)
    yn_local = be.normalize(y_local) if ny_local else _empty_normalized(be, y_local)
    coll = _collectives(ws)
    if coll:
        yn_all, y_counts = all_gather_rows(yn_local[:ny_local])
        ny = sum(y_counts)
        yn = be.pad_rows(yn_all, ny)
        xc = torch.tensor([nx_local], dtype=torch.int64, device=dev)
        xcs = [torch.zeros_like(xc) for _ in range(ws)]
        dist.all_gather(xcs, xc)
        x_counts = [int(c.item()) for c in xcs]
    else:
        yn, ny, x_counts = yn_local, ny_local, [nx_local]
    nx = sum(x_counts)
    if nx != ny:
        raise ValueError(f"xsim expects aligned x and y ({nx} vs {ny} rows in total)")
    if nx == 0:  # nothing to align on any rank: no error rate to speak of (every collective above has been joined)
        return float("nan"), torch.zeros(0, dtype=torch.int32, device=dev)
    x_off = _row_offsets(x_counts)[rank]
    errs = torch.zeros(1, dtype=torch.int32, device=dev)
    pred = torch.zeros(0, dtype=torch.int32, device=dev)
    if margin == "cosine":
        if nx_local:
            fs, fi = be.topk(xn, nx_local, yn, ny, 1)
            pred, _ = be.margin_select(fs, fi, None, "cosine", x_off, errs)
    else:
        # the neighbourhood size is a property of the PROBLEM (LASER: k = 4), not of the sharding: a rank
        # with fewer than kk local rows contributes the candidates it has, padded with -inf, so the merged
        # lists -- and the margin means -- equal the single-process result for every shard layout
        kk = min(k, nx)
        kloc = min(kk, nx_local)
        if kloc:
            bs_part, _ = be.topk(yn, ny, xn, nx_local, kloc, x_off)
            bs_part = bs_part[:ny]
            if kloc < kk:
                bs_part = torch.cat([bs_part, bs_part.new_full((ny, kk - kloc), float("-inf"))], dim=1)
        else:
            bs_part = torch.full((ny, kk), float("-inf"), dtype=torch.float32, device=dev)
        if coll:
            parts = bs_part.new_empty((ws * bs_part.shape[0], bs_part.shape[1]))
            dist.all_gather_into_tensor(parts, bs_part.contiguous())
            bs, _ = be.merge_topk(parts.view(ws, bs_part.shape[0], bs_part.shape[1]), None)
        else:
            bs = bs_part
        if nx_local:
            fs, fi = be.topk(xn, nx_local, yn, ny, kk)
            pred, _ = be.margin_select(fs, fi, bs, margin, x_off, errs)
    total = errs.to(torch.int64)
    if coll:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    return int(total.item()) / nx, pred
