"""SpeechToEmbeddingModelPipeline / SpeechToTextModelPipeline with the reference's interface
(sonar/inference_pipelines/speech.py:277-474) on the MI355X engine.

The reference's DataPipeline is
    read_sequence -> map(decode_audio) -> map(fbank, n_parallel) -> bucket(batch_size)
    -> map(Collater(pad_idx, pad_to_multiple=2), n_parallel) -> prefetch(n_prefetched_batches)
    -> to-device -> model                                            (speech.py:431-452)
Here the host half (file read + WAV decode on `n_parallel` threads, bucketing, packing the clips
of a batch into one pinned buffer, asynchronous H2D copy on a side stream) runs on a background
thread `n_prefetched_batches` batches ahead of the consumer; the device half (ONE filterbank
launch per batch that already writes the collated, padded [n, T, 80] tensor, conformer encoder,
attention pooler) runs on the caller's thread and stream, as the model does in the reference.

Audio decoding: the reference uses fairseq2n's libsndfile AudioDecoder (speech.py:292-308);
libsndfile is not in this image, so RIFF/WAVE (PCM 8/16/24/32, float 32/64, extensible) and native FLAC
streams are decoded by the engine's own host code (`smi_host_audio_decode`: csrc/host_input.cpp,
csrc/host_audio.cpp); other containers (Ogg, MP3, ...) raise a ValueError.
"""
from __future__ import annotations

import ctypes as C
import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from pathlib import Path
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import torch

from .. import _lib
from ..speech_encoder import (SonarSpeechEncoderModel, fbank_batch_flat, load_sonar_speech_encoder,
                              waveform_to_fbank, waveforms_to_fbank_batch)  # noqa: F401
from ..text_encoder import PaddingMask, SequenceBatch
from .utils import add_progress_bar

CPU = torch.device("cpu")


def decode_audio_bytes(data: bytes, name: str = "<bytes>") -> Tuple[torch.Tensor, int]:
    """WAV or FLAC file image -> (float32 [frames, channels] in [-1, 1), sample rate), decoded by the engine's host code."""
    lib = _lib.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    ch, rate, frames = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    try:
        _lib.check(lib.smi_host_audio_info(buf, len(data), C.byref(ch), C.byref(rate), C.byref(frames)))
        out = torch.empty((frames.value, ch.value), dtype=torch.float32)
        if frames.value:
            _lib.check(lib.smi_host_audio_decode(buf, len(data), C.c_void_p(out.data_ptr()), frames.value, ch.value))
    except _lib.SmiError as e:
        raise ValueError(f"{name}: {e}") from None
    return out, int(rate.value)


decode_wav_bytes = decode_audio_bytes  # the name of rounds 1-2


def read_wav(path: Union[str, Path]) -> torch.Tensor:
    """[channels, samples] float32 in [-1, 1]; 16 kHz is what the SONAR speech encoders were trained on
    (the reference asserts it in its tests, test_sonar_speech_pipeline_models.py:25-26)."""
    with open(str(path), "rb") as fh:
        data = fh.read()
    wav, rate = decode_audio_bytes(data, str(path))
    if rate != 16000:
        raise ValueError(f"{path}: sample rate {rate}, the SONAR speech encoders expect 16 kHz audio")
    return wav.t().contiguous()


@dataclass
class SpeechInferenceParams:
    """sonar/inference_pipelines/speech.py:42-73 (fields that apply to waveform inputs)."""

    batch_size: int = 3
    fbank_dtype: torch.dtype = torch.float32
    n_parallel: int = 1
    pad_idx: int = 0
    n_prefetched_batches: int = 2


@dataclass
class _HostBatch:
    cat: torch.Tensor                  # all clips of the batch, concatenated, on the device
    offsets: List[int]                 # clip i = cat[offsets[i]:offsets[i+1]]
    ready: Optional["torch.cuda.Event"]  # H2D copy done (recorded on the producer's stream)
    stage: Optional[torch.Tensor]      # pinned staging buffer, kept alive until the copy has run


class SpeechModelPipelineInterface(torch.nn.Module):
    """speech.py:277-308: audio decoding + the fbank converter settings shared by the speech pipelines
    (num_mel_bins=80, waveform_scale=2**15, channel_last, standardize)."""

    device: torch.device

    def _load_audio(self, inp: Union[str, Path, torch.Tensor]) -> torch.Tensor:
        """-> mono waveform [T] float32 (host tensor for files; tensors stay where they are).
        speech.py:298-308: tensor inputs are [C, T] at 16 kHz."""
        if isinstance(inp, torch.Tensor):
            wav = inp
            if wav.dim() == 1:
                wav = wav.unsqueeze(0)
            if wav.dim() != 2:
                raise ValueError("waveform tensors must be [channels, samples]")
        else:
            wav = read_wav(inp)
        # channel_last fbank of a multi-channel clip uses the first channel (kaldi takes channel 0)
        return wav[0].to(torch.float32)

    # kept for callers of the round-1 name
    def _decode_audio(self, inp: Union[str, Path, torch.Tensor]) -> torch.Tensor:
        return self._load_audio(inp).to(self.device)

    def _host_batches(self, items: Sequence, batch_size: int, n_parallel: int) -> Iterator[_HostBatch]:
        """The host half of the pipeline for every bucket of `batch_size` inputs."""
        dev = torch.device(self.device)
        side = torch.cuda.Stream(dev) if dev.type == "cuda" else None
        with ThreadPoolExecutor(max_workers=max(1, int(n_parallel))) as pool:
            for i in range(0, len(items), batch_size):
                wavs = list(pool.map(self._load_audio, items[i:i + batch_size]))
                offs = [0]
                for w in wavs:
                    offs.append(offs[-1] + w.numel())
                if any(w.is_cuda for w in wavs):  # device tensors in: nothing to stage
                    cat = torch.cat([w.to(dev) for w in wavs])
                    ev = None
                    if dev.type == "cuda":
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(dev))
                    yield _HostBatch(cat, offs, ev, None)
                    continue
                stage = torch.empty(max(offs[-1], 1), dtype=torch.float32, pin_memory=dev.type == "cuda")
                for w, o in zip(wavs, offs):
                    stage[o:o + w.numel()] = w
                if side is None:
                    yield _HostBatch(stage[:offs[-1]].clone(), offs, None, None)
                    continue
                with torch.cuda.stream(side):
                    cat = stage[:offs[-1]].to(dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
                yield _HostBatch(cat, offs, ev, stage)

    def _prefetched(self, items: Sequence, batch_size: int, n_parallel: int, depth: int) -> Iterator[_HostBatch]:
        """`.prefetch(n_prefetched_batches)`: the host half runs on a background thread."""
        if depth <= 0:
            yield from self._host_batches(items, batch_size, n_parallel)
            return
        q: "queue.Queue" = queue.Queue(maxsize=depth)
        end = object()
        stop = threading.Event()

        def offer(x) -> bool:  # False once the consumer has gone away (early exit / exception on its side)
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                for hb in self._host_batches(items, batch_size, n_parallel):
                    if not offer(hb):
                        return
                offer(end)
            except BaseException as e:  # surfaces on the consumer thread
                offer(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                x = q.get()
                if x is end:
                    break
                if isinstance(x, BaseException):
                    raise x
                yield x
        finally:
            stop.set()
            th.join(timeout=5)

    def _fbank_batch(self, hb: _HostBatch, pad_idx: int) -> Tuple[SequenceBatch, List[int]]:
        """Device half up to the model input: fbank of the whole batch in one launch, collated as
        Collater(pad_value=pad_idx, pad_to_multiple=2) (speech.py:444)."""
        if hb.ready is not None:
            cur = torch.cuda.current_stream(hb.cat.device)
            cur.wait_event(hb.ready)
            # `cat` was allocated on the producer's side stream: tell the caching allocator that THIS stream reads
            # it, or the block returns to the side-stream pool when `hb` is dropped and a later batch's H2D copy
            # (issued batches ahead of the GPU) may overwrite it before the filterbank kernel has run
            hb.cat.record_stream(cur)
        fb, lens = fbank_batch_flat(hb.cat, hb.offsets)
        t = fb.shape[1]
        if pad_idx != 0:
            for i, l in enumerate(lens):
                fb[i, l:] = float(pad_idx)
        ragged = any(l != t for l in lens)
        mask = PaddingMask(torch.tensor(lens, dtype=torch.int32), t) if ragged else None
        return SequenceBatch(fb, mask), lens


class SpeechToEmbeddingModelPipeline(SpeechModelPipelineInterface):
    model: SonarSpeechEncoderModel

    def __init__(self, encoder: Union[str, Path, SonarSpeechEncoderModel], device: torch.device = CPU,
                 fbank_dtype: torch.dtype = torch.float32) -> None:
        """
        Args:
            encoder: a card name resolved under $SONAR_CHECKPOINT_DIR, a checkpoint path, or a model object
            device: the HIP device; this engine has no CPU path, so a CPU device raises.
            fbank_dtype: as in the reference (speech.py:426-429): torch.float16 on a GPU runs the model in half
                precision (fp16 residual stream and embeddings), the default float32 an fp32 model; the
                filterbank features themselves are always fp32 on the device.
        """
        super().__init__()
        device = torch.device(device)
        if isinstance(encoder, (str, Path)):
            if device.type != "cuda":
                raise RuntimeError("the MI355X SONAR engine needs device='cuda[:i]' (no CPU path)")
            encoder = load_sonar_speech_encoder(str(encoder), device=device,
                                                dtype=fbank_dtype if fbank_dtype in (torch.float16, torch.bfloat16) else torch.float32)
        self.model = encoder.eval()
        self.device = getattr(encoder, "device", device)
        self.fbank_dtype = fbank_dtype

    @torch.inference_mode()
    def predict(self, input: Sequence[Union[str, Path, torch.Tensor]], batch_size: int = 3, n_parallel: int = 1,
                pad_idx: int = 0, n_prefetched_batches: int = 2, progress_bar: bool = False) -> torch.Tensor:
        if batch_size <= 0:
            raise ValueError("`batch_size` should be strictly positive")
        items = list(input)
        batches: Iterable = self._prefetched(items, batch_size, n_parallel, n_prefetched_batches)
        if progress_bar:
            batches = add_progress_bar(batches, inputs=items, batch_size=batch_size)
        results: List[torch.Tensor] = []
        for hb in batches:
            batch, _ = self._fbank_batch(hb, pad_idx)
            results.append(self.model(batch).sentence_embeddings)
        if not results:
            return torch.empty((0, self.model.model_dim), dtype=self.model.dtype, device=self.device)
        return torch.cat(results, dim=0)


class SpeechToTextModelPipeline(SpeechModelPipelineInterface):
    """sonar/inference_pipelines/speech.py:310-399: audio -> sentence vector (speech engine) -> text
    (decoder engine, beam search).  The generator's length cap follows fairseq2's rule with the fbank
    frame count of the batch as source length (`converter.batch_convert(batch.seqs, batch.padding_mask)`,
    speech.py:369-372)."""

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
        if batch_size <= 0:
            raise ValueError("`batch_size` should be strictly positive")
        items = list(input)
        batches: Iterable = self._prefetched(items, batch_size, n_parallel, n_prefetched_batches)
        if progress_bar:
            batches = add_progress_bar(batches, inputs=items, batch_size=batch_size)
        out: List[str] = []
        for hb in batches:
            batch, lens = self.s2vec._fbank_batch(hb, pad_idx)
            emb = self.s2vec.model(batch).sentence_embeddings
            src_len = max(lens) if batch.padding_mask is not None else batch.seqs.shape[1]
            out.extend(self.vec2t.predict(emb, target_lang=target_lang, batch_size=emb.shape[0],
                                          source_len=src_len, **generator_kwargs))
        return out
