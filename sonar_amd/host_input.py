"""Native host input path of TextToEmbeddingModelPipeline.predict.

The reference builds a fairseq2n (C++) DataPipeline: tokenize -> truncate ->
dynamic_bucket -> Collater(pad) -> to-device -> prefetch(2)
(sonar/inference_pipelines/text.py:226-247).  At ~7.5 k sentences/s per GPU a
per-sentence Python loop cannot feed the engine, so the same stages run here as

  * SentencePiece's own multi-threaded batch encode over chunks of sentences,
  * `smi_host_token_lengths` / `smi_host_dynamic_bucket` / `smi_host_collate_nllb`
    (C++ host threads, include/sonar_mi355.h) for id assembly, truncation, bucketing and
    right-padded collation straight into a ring of pinned staging buffers,
  * an asynchronous H2D copy per batch.

Bucket composition, truncation and padding are identical to the per-sentence
path in inference_pipelines/text.py (tests/test_pipeline_cpu.py checks that).
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .text_encoder import PaddingMask, SequenceBatch

_I64P = C.POINTER(C.c_int64)


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class _PinnedRing:
    """A few pinned int64 staging buffers reused round-robin; a slot is reused only after the
    H2D copy that read it has completed (event per slot)."""

    def __init__(self, device: torch.device, slots: int = 4):
        self.device = device
        self.pinned = device.type == "cuda" and torch.cuda.is_available()
        self.bufs: List[Optional[torch.Tensor]] = [None] * slots
        self.events: List[Optional[torch.cuda.Event]] = [None] * slots
        self.next = 0

    def take(self, numel: int) -> Tuple[torch.Tensor, int]:
        i = self.next
        self.next = (i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
        b = self.bufs[i]
        if b is None or b.numel() < numel:
            b = torch.empty(max(numel, 1 << 16), dtype=torch.int64, pin_memory=self.pinned)
            self.bufs[i] = b
        return b[:numel], i

    def copied(self, slot: int) -> None:
        if self.pinned:
            ev = self.events[slot] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.events[slot] = ev


def iter_text_batches(texts: Sequence[str], order: Sequence[int], encoder, *, max_seq_len: Optional[int],
                      batch_size: Optional[int], batch_max_tokens: Optional[int], pad_idx: int,
                      device: torch.device, chunk: int = 8192, num_threads: Optional[int] = None,
                      stats: Optional[dict] = None) -> Iterator[SequenceBatch]:
    """Yield the model-ready batches of `predict()` for the sentences `texts[i], i in order`.

    encoder: an `NllbEncoder` (provides `.tok.sp`, `.prefix`, `.suffix`).
    stats["n_truncated"] is incremented by the number of truncated sequences.
    """
    lib = _lib.load()
    threads = num_threads or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8)
    prefix = np.asarray(encoder.prefix, dtype=np.int64)
    suffix = np.asarray(encoder.suffix, dtype=np.int64)
    threshold = batch_max_tokens or 2**31
    max_num = batch_size or 20_000
    limit = int(max_seq_len) if max_seq_len is not None else 0
    ring = _PinnedRing(device)
    sp = encoder.tok.sp

    # carried-over (not yet bucketed) sequences
    c_pieces = np.empty(0, dtype=np.int32)
    c_off = np.zeros(1, dtype=np.int64)
    c_lens = np.empty(0, dtype=np.int32)

    def emit(pieces, off, lens, b0, b1) -> SequenceBatch:
        nb = b1 - b0
        s_max = int(lens[b0:b1].max())
        stage, slot = ring.take(nb * s_max)
        _lib.check(lib.smi_host_collate_nllb(_ptr(pieces), _ptr(off), _ptr(lens), b0, nb, _ptr(prefix), len(prefix),
                                             _ptr(suffix), len(suffix), 1, pad_idx, C.c_void_p(stage.data_ptr()),
                                             s_max, threads))
        seqs = stage.view(nb, s_max).to(device, non_blocking=True)
        ready = None
        if device.type != "cuda":
            seqs = seqs.clone()  # the staging slot is reused
        else:
            # this runs on the prefetch thread (its own current stream); the consumer's stream waits for
            # the copy through this event before the model reads `seqs`
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(device))
        ring.copied(slot)
        bl = torch.from_numpy(lens[b0:b1].copy())
        if bool((bl != s_max).any()):
            return SequenceBatch(seqs, PaddingMask(bl, s_max), ready)
        return SequenceBatch(seqs, None, ready)

    order = list(order)
    for c0 in range(0, len(order), chunk):
        idx = order[c0:c0 + chunk]
        enc = sp.encode([texts[i] for i in idx], num_threads=threads)
        n_new = len(enc)
        plen = np.fromiter(map(len, enc), dtype=np.int64, count=n_new)
        off_new = np.zeros(n_new + 1, dtype=np.int64)
        np.cumsum(plen, out=off_new[1:])
        pieces_new = np.fromiter(itertools.chain.from_iterable(enc), dtype=np.int32, count=int(off_new[-1]))
        lens_new = np.empty(n_new, dtype=np.int32)
        cut = C.c_int64(0)
        _lib.check(lib.smi_host_token_lengths(_ptr(off_new), n_new, len(prefix), len(suffix), limit, _ptr(lens_new),
                                              C.byref(cut)))
        if stats is not None:
            stats["n_truncated"] = stats.get("n_truncated", 0) + int(cut.value)
        # append to the carried sequences
        pieces = np.concatenate([c_pieces, pieces_new]) if len(c_pieces) else pieces_new
        off = np.concatenate([c_off, off_new[1:] + c_off[-1]]) if len(c_lens) else off_new
        lens = np.concatenate([c_lens, lens_new]) if len(c_lens) else lens_new
        n = len(lens)
        bounds = np.empty(n + 1, dtype=np.int64)
        nb, nopen = C.c_int64(0), C.c_int64(0)
        _lib.check(lib.smi_host_dynamic_bucket(_ptr(lens), n, threshold, max_num, 1, _ptr(bounds), C.byref(nb),
                                               C.byref(nopen)))
        for b in range(nb.value):
            yield emit(pieces, off, lens, int(bounds[b]), int(bounds[b + 1]))
        done = int(bounds[nb.value])
        c_pieces = pieces[off[done]:].copy()
        c_off = (off[done:] - off[done]).copy()
        c_lens = lens[done:].copy()
    if len(c_lens):  # drop_remainder=False: the trailing open bucket is emitted as it is
        yield emit(c_pieces, c_off, c_lens, 0, len(c_lens))
