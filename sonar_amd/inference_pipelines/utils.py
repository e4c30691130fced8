"""Pipeline helpers mirroring sonar/inference_pipelines/utils.py:18-46."""
from __future__ import annotations

from typing import Iterable, Optional, Sequence

import torch

from ..text_encoder import PaddingMask, SequenceBatch


def extract_sequence_batch(x: dict, device) -> SequenceBatch:
    """dict{seqs [N,S], seq_lens [N], is_ragged} -> SequenceBatch on `device`;
    the padding mask is None when the batch is not ragged (utils.py:18-21)."""
    seqs = x["seqs"].to(device, non_blocking=True)
    ready = None
    if seqs.is_cuda:  # called on the prefetch thread: let the consumer's stream wait for the copy
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(seqs.device))
    if not x["is_ragged"]:
        return SequenceBatch(seqs, None, ready)
    return SequenceBatch(seqs, PaddingMask(x["seq_lens"], seqs.shape[1]), ready)


def add_progress_bar(sequence: Iterable, inputs: Optional[Sequence] = None,
                     batch_size: Optional[int] = None, **kwargs) -> Iterable:
    """tqdm wrapper (utils.py:24-46); a no-op iterator when tqdm is unavailable."""
    try:
        from tqdm.auto import tqdm
    except ImportError:
        return sequence
    total = None
    if inputs is not None and batch_size is not None and hasattr(inputs, "__len__"):
        total = (len(inputs) + batch_size - 1) // batch_size
    return tqdm(sequence, total=total, **kwargs)
