"""NLLB-style SentencePiece tokenizer used by the SONAR text pipelines.

The reference loads it from fairseq2 (`tokenizer_family: nllb`,
sonar/cards/text_sonar_basic_encoder.yaml:11-13) and uses it through
`tokenizer.create_encoder(lang=..., device=...)`, `tokenizer.vocab_info.pad_idx`
(sonar/inference_pipelines/text.py:199-201,241) and `create_decoder()`
(text.py:322-327).  fairseq2 is not a dependency here; the id layout restated
from its published behaviour is:

    0 <pad>, 1 <unk>, 2 <s>, 3 </s>            (SentencePiece piece p -> p + 1)
    256001 + i : __lang_i__ in card order       (eng_Latn = 256047, fra_Latn = 256057)
    then <MINED_DATA>, <MMT_BT_DATA>, <SMT_BT_DATA>   => vocabulary size 256206

    source mode : [__lang__] pieces... [</s>]
    target mode : [</s>] [__lang__] pieces...          (decoder prompt prefix)

Tokenisation stays on the host (it is string processing; the reference runs it
in fairseq2n C++ threads).  `encode_batch` uses SentencePiece's own
multi-threaded batch API.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Callable, List, Optional, Sequence, Union

import torch

from .nllb_langs import NLLB_EXTRA_CONTROL, NLLB_LANGS


@dataclass
class TokenizerVocabularyInfo:
    size: int
    unk_idx: int = 1
    bos_idx: int = 2
    eos_idx: int = 3
    pad_idx: int = 0


class NllbTokenizer:
    def __init__(self, spm_model: Union[str, Path, bytes], langs: Optional[Sequence[str]] = None,
                 default_lang: str = "eng_Latn"):
        import sentencepiece as spm

        if isinstance(spm_model, (bytes, bytearray)):
            self.sp = spm.SentencePieceProcessor(model_proto=bytes(spm_model))
        else:
            self.sp = spm.SentencePieceProcessor(model_file=str(spm_model))
        self.langs = list(langs) if langs is not None else list(NLLB_LANGS)
        self.default_lang = default_lang
        self.num_pieces = self.sp.get_piece_size()
        self.lang_base = self.num_pieces + 1
        self.lang_to_idx = {l: self.lang_base + i for i, l in enumerate(self.langs)}
        size = self.lang_base + len(self.langs) + len(NLLB_EXTRA_CONTROL)
        self.vocab_info = TokenizerVocabularyInfo(size=size)

    # -- reference-shaped factory methods ---------------------------------
    def lang_idx(self, lang: str) -> int:
        try:
            return self.lang_to_idx[lang]
        except KeyError:
            raise ValueError(f"`lang` must be a supported language, but is {lang!r} instead") from None

    def create_encoder(self, *, task: Optional[str] = None, lang: Optional[str] = None,
                       mode: Optional[str] = None, device=None, pin_memory: bool = False) -> "NllbEncoder":
        if task is not None and task != "translation":
            raise ValueError(f"`task` must be 'translation', but is {task!r} instead")
        return NllbEncoder(self, lang or self.default_lang, mode or "source", device)

    def create_decoder(self) -> Callable[[torch.Tensor], str]:
        return self.decode

    # -- helpers ------------------------------------------------------------
    def decode(self, ids: Union[torch.Tensor, Sequence[int]]) -> str:
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        # control symbols (pad, bos, eos, language tokens) are dropped; <unk> (id 1 = SentencePiece
        # piece 0) is kept: SentencePiece renders it with its unk surface (" \u2047 "), as fairseq2's
        # decoder does
        unk = self.vocab_info.unk_idx
        pieces = [i - 1 for i in ids if 4 <= i < self.lang_base or i == unk]
        return self.sp.decode(pieces)


class NllbEncoder:
    """str -> int64 tensor; `prefix_indices`/`suffix_indices` as in fairseq2's token encoders."""

    def __init__(self, tok: NllbTokenizer, lang: str, mode: str, device=None):
        self.tok = tok
        lang_id = tok.lang_idx(lang)
        eos = tok.vocab_info.eos_idx
        if mode == "source":
            self.prefix, self.suffix = [lang_id], [eos]
        elif mode == "target":
            self.prefix, self.suffix = [eos, lang_id], []
        else:
            raise ValueError(f"`mode` must be 'source' or 'target', but is {mode!r} instead")
        self.device = device

    @property
    def prefix_indices(self) -> torch.Tensor:
        return torch.tensor(self.prefix, dtype=torch.int64)

    @property
    def suffix_indices(self) -> torch.Tensor:
        return torch.tensor(self.suffix, dtype=torch.int64)

    def ids(self, text: str) -> List[int]:
        return self.prefix + [p + 1 for p in self.tok.sp.encode(text)] + self.suffix

    def encode_batch(self, texts: Sequence[str], num_threads: int = -1) -> List[List[int]]:
        enc = self.tok.sp.encode(list(texts), num_threads=num_threads)
        return [self.prefix + [p + 1 for p in e] + self.suffix for e in enc]

    def __call__(self, text: str) -> torch.Tensor:
        t = torch.tensor(self.ids(text), dtype=torch.int64)
        return t.to(self.device) if self.device is not None and str(self.device) != "cpu" else t
