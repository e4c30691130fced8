"""Samplers accepted by `EmbeddingToTextModelPipeline.predict(sampler=...)`.

The reference passes a fairseq2 `Sampler` to `SamplingSeq2SeqGenerator`
(sonar/inference_pipelines/text.py:315-320); fairseq2.generation ships `TopKSampler(k)` and
`TopPSampler(p=0.9)`.  These classes carry the same constructor arguments; the filtering and the draw
run on the device (`smi_text_decoder_sample`, csrc/sampling.hip).  Objects with the same class names
from fairseq2 itself are accepted too (duck-typed on `k` / `p`, see `resolve_sampler`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

from . import _lib


@dataclass(frozen=True)
class TopKSampler:
    """Sample among the `k` most probable tokens (fairseq2.generation.TopKSampler)."""

    k: int

    def __post_init__(self):
        if self.k < 1:
            raise ValueError(f"`k` must be greater than or equal to 1, but is {self.k} instead.")


@dataclass(frozen=True)
class TopPSampler:
    """Nucleus sampling: the smallest set of most probable tokens whose cumulative probability
    exceeds `p` (fairseq2.generation.TopPSampler)."""

    p: float = 0.9

    def __post_init__(self):
        if not 0.0 < self.p <= 1.0:
            raise ValueError(f"`p` must be in (0, 1], but is {self.p} instead.")


def resolve_sampler(sampler) -> Tuple[int, int, float]:
    """-> (SMI_SAMPLER_*, k, p) for one of the classes above or a fairseq2 object of the same kind."""
    name = type(sampler).__name__
    if isinstance(sampler, TopKSampler) or name == "TopKSampler":
        k = int(getattr(sampler, "k", getattr(sampler, "_k", 0)))
        if k < 1:
            raise ValueError("TopKSampler: k must be >= 1")
        return _lib.SMI_SAMPLER_TOP_K, k, 1.0
    if isinstance(sampler, TopPSampler) or name == "TopPSampler":
        p = float(getattr(sampler, "p", getattr(sampler, "_p", 0.0)))
        if not 0.0 < p <= 1.0:
            raise ValueError("TopPSampler: p must be in (0, 1]")
        return _lib.SMI_SAMPLER_TOP_P, 1, p
    raise NotImplementedError(f"sampler {name!r} is not covered by the MI355X engine (TopKSampler, TopPSampler)")
