"""Card-name resolution for local checkpoints.

The reference pipelines accept asset-card NAMES ("text_sonar_basic_encoder", ...) and let
fairseq2's asset store download the files the card points at (sonar/cards/*.yaml,
sonar/inference_pipelines/text.py:157-167).  There is no network and no asset store here; a card
name resolves to the file the card's URL ends in, looked up in `$SONAR_CHECKPOINT_DIR`
(or `~/.cache/sonar_amd`).  Anything that is not a known card name is taken as a path.

    text_sonar_basic_encoder      -> sonar_text_encoder.pt           arch basic
    text_sonar_basic_decoder      -> sonar_text_decoder.pt           arch basic
    text_sonar_finetuned_decoder  -> finetuned_decoder.pt            arch basic
    (tokenizer of all three)      -> sentencepiece.source.256000.model
    sonar_speech_encoder_eng      -> spenc.eng.pt                    arch english
    sonar_speech_encoder_<lang>   -> spenc.v3ap.<lang>.pt | spenc.v5ap.<lang>.pt   arch non_english
    blaser_2_0_ref / blaser_2_0_qe -> blaser_2_0_ref.pt / blaser_2_0_qe.pt (the cards point at
                                      huggingface `model.pt` files; store them under these names)
    sonar_mutox                   -> mutox.pt
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Union

NLLB_SPM = "sentencepiece.source.256000.model"

_TEXT_CARDS = {
    "text_sonar_basic_encoder": ("sonar_text_encoder.pt", "basic"),
    "text_sonar_basic_decoder": ("sonar_text_decoder.pt", "basic"),
    "text_sonar_finetuned_decoder": ("finetuned_decoder.pt", "basic"),
}
_HEAD_CARDS = {
    "blaser_2_0_ref": ("blaser_2_0_ref.pt", "basic_ref"),
    "blaser_2_0_qe": ("blaser_2_0_qe.pt", "basic_qe"),
    "sonar_mutox": ("mutox.pt", "mutox"),
}
_SPEECH_RE = re.compile(r"^sonar_speech_encoder_([a-z]{3})$")


@dataclass
class ResolvedCard:
    name: str
    checkpoint: Path
    arch: str
    tokenizer: Optional[Path] = None


def asset_dirs() -> List[Path]:
    dirs = []
    env = os.environ.get("SONAR_CHECKPOINT_DIR")
    if env:
        dirs.append(Path(env))
    dirs.append(Path.home() / ".cache" / "sonar_amd")
    return dirs


def _find(basenames: List[str]) -> Optional[Path]:
    for d in asset_dirs():
        for b in basenames:
            p = d / b
            if p.is_file():
                return p
    return None


def is_card_name(name: Union[str, Path]) -> bool:
    s = str(name)
    return s in _TEXT_CARDS or s in _HEAD_CARDS or bool(_SPEECH_RE.match(s))


def resolve_card(name: Union[str, Path]) -> ResolvedCard:
    """Card name -> local files.  Raises FileNotFoundError naming the file and the directories searched."""
    s = str(name)
    if s in _TEXT_CARDS:
        base, arch = _TEXT_CARDS[s]
        cands = [base]
    elif s in _HEAD_CARDS:
        base, arch = _HEAD_CARDS[s]
        cands = [base]
    else:
        m = _SPEECH_RE.match(s)
        if not m:
            raise KeyError(f"{s!r} is not a known SONAR card name")
        lang = m.group(1)
        if lang == "eng":
            cands, arch = ["spenc.eng.pt"], "english"
        else:
            cands, arch = [f"spenc.v3ap.{lang}.pt", f"spenc.v5ap.{lang}.pt"], "non_english"
    ckpt = _find(cands)
    if ckpt is None:
        raise FileNotFoundError(
            f"card {s!r}: none of {cands} found in {[str(d) for d in asset_dirs()]} "
            "(set SONAR_CHECKPOINT_DIR to the directory holding the downloaded SONAR files)")
    tok = _find([NLLB_SPM]) if s in _TEXT_CARDS else None
    return ResolvedCard(s, ckpt, arch, tok)


def resolve_checkpoint(name_or_path: Union[str, Path], default_arch: str):
    """(path, arch) for a card name or a plain path."""
    if is_card_name(name_or_path) and not Path(str(name_or_path)).exists():
        r = resolve_card(name_or_path)
        return r.checkpoint, r.arch
    return Path(str(name_or_path)), default_arch


def resolve_tokenizer(name_or_path: Union[str, Path]) -> Path:
    """SentencePiece model path for a text card name or a plain path."""
    if is_card_name(name_or_path) and not Path(str(name_or_path)).exists():
        tok = _find([NLLB_SPM])
        if tok is None:
            raise FileNotFoundError(f"tokenizer of card {name_or_path!r}: {NLLB_SPM} not found in "
                                    f"{[str(d) for d in asset_dirs()]}")
        return tok
    return Path(str(name_or_path))
