"""SpeechToEmbeddingModelPipeline with the reference's interface
(sonar/inference_pipelines/speech.py:402-474) on the MI355X engine.

Audio decoding stays on the host (the reference uses fairseq2n's libsndfile AudioDecoder,
speech.py:111-141,298-308); here 16-bit / float PCM WAV files are read with the standard
library.  The filterbank, the conformer encoder and the attention pooler run on the GPU.
"""
from __future__ import annotations

import wave
from dataclasses import dataclass
from pathlib import Path
from typing import Iterable, List, Optional, Sequence, Union

import torch

from ..speech_encoder import (SonarSpeechEncoderModel, load_sonar_speech_encoder, waveform_to_fbank,
                              waveforms_to_fbank_batch)
from ..text_encoder import PaddingMask, SequenceBatch
from .utils import add_progress_bar

CPU = torch.device("cpu")


def read_wav(path: Union[str, Path]) -> torch.Tensor:
    """[channels, samples] float32 in [-1, 1]; 16 kHz is assumed by the pipeline (speech.py:299-304)."""
    with wave.open(str(path), "rb") as w:
        nch, width, rate, nfr = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(nfr)
    if rate != 16000:
        raise ValueError(f"{path}: sample rate {rate}, the SONAR speech encoders expect 16 kHz audio")
    if width == 2:
        x = torch.frombuffer(bytearray(raw), dtype=torch.int16).float() / 32768.0
    elif width == 4:
        x = torch.frombuffer(bytearray(raw), dtype=torch.int32).float() / 2147483648.0
    elif width == 1:
        x = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).float() - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    return x.view(-1, nch).t().contiguous()


@dataclass
class SpeechInferenceParams:
    """sonar/inference_pipelines/speech.py:42-73 (fields that apply to waveform inputs)."""

    batch_size: int = 3
    fbank_dtype: torch.dtype = torch.float32
    n_parallel: int = 1
    pad_idx: int = 0
    n_prefetched_batches: int = 2


class SpeechToEmbeddingModelPipeline(torch.nn.Module):
    model: SonarSpeechEncoderModel

    def __init__(self, encoder: Union[str, Path, SonarSpeechEncoderModel], device: torch.device = CPU,
                 fbank_dtype: torch.dtype = torch.float32) -> None:
        """
        Args:
            encoder: a checkpoint path (`english` arch) or a model object
            device: the HIP device; this engine has no CPU path, so a CPU device raises.
            fbank_dtype: kept for interface parity; features are fp32 on device.
        """
        super().__init__()
        device = torch.device(device)
        if isinstance(encoder, (str, Path)):
            if device.type != "cuda":
                raise RuntimeError("the MI355X SONAR engine needs device='cuda[:i]' (no CPU path)")
            encoder = load_sonar_speech_encoder(str(encoder), device=device)
        self.model = encoder.eval()
        self.device = getattr(encoder, "device", device)
        self.fbank_dtype = fbank_dtype

    def _decode_audio(self, inp: Union[str, Path, torch.Tensor]) -> torch.Tensor:
        """-> mono waveform [T] on the device (speech.py:298-308: tensors are [C, T])."""
        if isinstance(inp, torch.Tensor):
            wav = inp
            if wav.dim() == 1:
                wav = wav.unsqueeze(0)
            if wav.dim() != 2:
                raise ValueError("waveform tensors must be [channels, samples]")
        else:
            wav = read_wav(inp)
        # channel_last fbank of a multi-channel clip uses the first channel (kaldi takes channel 0)
        return wav[0].to(self.device, torch.float32)

    @torch.inference_mode()
    def predict(self, input: Sequence[Union[str, Path, torch.Tensor]], batch_size: int = 3, n_parallel: int = 1,
                pad_idx: int = 0, n_prefetched_batches: int = 2, progress_bar: bool = False) -> torch.Tensor:
        if batch_size <= 0:
            raise ValueError("`batch_size` should be strictly positive")
        items = list(input)
        batches: Iterable = [items[i:i + batch_size] for i in range(0, len(items), batch_size)]
        if progress_bar:
            batches = add_progress_bar(batches, inputs=items, batch_size=batch_size)
        results: List[torch.Tensor] = []
        for chunk in batches:
            # fbank of the whole batch in one launch, collated as Collater(pad_to_multiple=2) (speech.py:444)
            fb, lens = waveforms_to_fbank_batch([self._decode_audio(x) for x in chunk])
            t = fb.shape[1]
            if pad_idx != 0:
                for i, l in enumerate(lens):
                    fb[i, l:] = float(pad_idx)
            ragged = any(l != t for l in lens)
            mask = PaddingMask(torch.tensor(lens, dtype=torch.int32), t) if ragged else None
            results.append(self.model(SequenceBatch(fb, mask)).sentence_embeddings)
        if not results:
            return torch.empty((0, self.model.model_dim), dtype=self.model.dtype, device=self.device)
        return torch.cat(results, dim=0)


class SpeechToTextModelPipeline(torch.nn.Module):
    """sonar/inference_pipelines/speech.py:310-399: audio -> sentence vector (speech engine) -> text
    (decoder engine, beam search)."""

    def __init__(self, encoder, decoder, tokenizer, device: torch.device = CPU,
                 fbank_dtype: torch.dtype = torch.float32) -> None:
        super().__init__()
        from .text import EmbeddingToTextModelPipeline

        self.s2vec = SpeechToEmbeddingModelPipeline(encoder, device=device, fbank_dtype=fbank_dtype)
        self.vec2t = EmbeddingToTextModelPipeline(decoder, tokenizer, device=device)
        self.tokenizer = self.vec2t.tokenizer
        self.device = self.s2vec.device

    @torch.inference_mode()
    def predict(self, input: Sequence[Union[str, Path, torch.Tensor]], target_lang: str, batch_size: int = 3,
                n_parallel: int = 1, pad_idx: int = 0, n_prefetched_batches: int = 2, progress_bar: bool = False,
                **generator_kwargs) -> List[str]:
        items = list(input)
        out: List[str] = []
        for i in range(0, len(items), batch_size):
            chunk = items[i:i + batch_size]
            emb = self.s2vec.predict(chunk, batch_size=len(chunk), pad_idx=pad_idx)
            out.extend(self.vec2t.predict(emb, target_lang=target_lang, batch_size=len(chunk), **generator_kwargs))
        return out
