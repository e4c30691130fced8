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
from .utils import add_progress_bar, extract_sequence_batch

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
    """sonar/inference_pipelines/speech.py:42-73 -- the configuration of the TSV-driven pipelines below (same fields,
    same order, same defaults; `device` must be the HIP device of the engine)."""

    data_file: Path
    """The pathname of the test TSV data file."""

    audio_root_dir: Path
    """The pathname of the directory under which audio files are stored."""

    audio_path_index: int
    """Column index of audio path in given TSV data file."""

    batch_size: int
    """The batch size for model input."""

    fbank_dtype: torch.dtype = torch.float32

    target_lang: Optional[str] = None
    """The target translation language."""

    pad_idx: int = 0
    """Padding idx to use after applying fbank"""

    device: torch.device = CPU
    """The device on which to run inference."""

    n_parallel: int = 4
    """Number of parallel calls when running the pipeline."""

    n_prefetched_batches: int = 4
    """Number of prefetched batches"""


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


# ------------------------------------------------------------------------------------------------------------------
# TSV-driven pipelines (sonar/inference_pipelines/speech.py:76-274): the reference's own golden test drives the speech
# encoder through these (tests/integration_tests/test_sonar_speech_encoder.py:27-78).  fairseq2's DataPipelineBuilder
# is replaced by a small lazy builder with the operators these pipelines use (map with a selector, and_return); the
# stages themselves -- file read + decode on `n_parallel` host threads, bucketing, ONE filterbank launch per bucket that
# writes the Collater(pad_value, pad_to_multiple=2) layout, prefetch -- are the model pipelines' host / device halves.
class DataPipeline:
    """Re-iterable result of `DataPipelineBuilder.and_return()` (fairseq2.data.DataPipeline)."""

    def __init__(self, source, stages) -> None:
        self._source, self._stages = source, list(stages)

    def __iter__(self):
        for item in self._source():
            for fn, selector in self._stages:
                item = _apply_selected(item, fn, selector)
            yield item

    def reset(self) -> None:  # fairseq2 API; every __iter__ starts from the beginning here
        pass


def _apply_selected(item, fn, selector: Optional[str]):
    """fairseq2's `map(fn, selector="a.b.c")`: replace the element at that path of nested dicts by fn(element)."""
    if not selector:
        return fn(item)
    keys = selector.split(".")
    node = item
    for k in keys[:-1]:
        node = node[k]
    node[keys[-1]] = fn(node[keys[-1]])
    return item


class DataPipelineBuilder:
    """The subset of fairseq2.data.DataPipelineBuilder the speech pipelines compose with."""

    def __init__(self, source) -> None:
        self._source, self._stages = source, []

    def map(self, fn, selector: Optional[str] = None, num_parallel_calls: int = 1) -> "DataPipelineBuilder":
        fns = list(fn) if isinstance(fn, (list, tuple)) else [fn]
        for f in fns:
            self._stages.append((f, selector))
        return self

    def and_return(self) -> DataPipeline:
        return DataPipeline(self._source, self._stages)


class SpeechInferencePipeline:
    """speech.py:77-92."""

    def prebuild_pipeline(self, context: SpeechInferenceParams) -> DataPipelineBuilder:
        raise NotImplementedError

    def build_pipeline(self, context: SpeechInferenceParams) -> DataPipeline:
        return self.prebuild_pipeline(context).and_return()


def read_tsv_column(data_file: Union[str, Path], index: int) -> List[str]:
    """`read_text(data_file, rtrim=True).skip(1).map(StrSplitter(indices=[index]))` (speech.py:99-108): the `index`-th
    tab-separated field of every line after the header."""
    out: List[str] = []
    with open(str(data_file), "r", encoding="utf-8") as fh:
        for ln, line in enumerate(fh):
            if ln == 0:
                continue
            line = line.rstrip()
            if not line:
                continue
            fields = line.split("\t")
            if index >= len(fields):
                raise ValueError(f"{data_file}:{ln + 1}: no column {index} in a line of {len(fields)} fields")
            out.append(fields[index])
    return out


class AudioToFbankDataPipelineBuilder(SpeechInferencePipeline):
    """speech.py:95-151: TSV -> audio files under `audio_root_dir` -> decode -> fbank (80 bins, scale 2**15,
    standardised) -> buckets of `batch_size` -> Collater(pad_idx, pad_to_multiple=2) -> prefetch.  Elements are what
    the reference's pipeline yields after the collate step:
    {"audio": {"path": [...], "data": {"fbank": {"seqs" [n, T, 80], "seq_lens" [n], "is_ragged"},
                                       "sample_rate": [...]}}}."""

    def _batches(self, context: SpeechInferenceParams, paths: Sequence[Path]) -> Iterator[Tuple[SequenceBatch, List[int]]]:
        """Decode + filterbank + collate for every bucket of `batch_size` files: the model pipelines' host half
        (threads, pinned staging, side-stream H2D, prefetch) and ONE filterbank launch per bucket."""
        dev = torch.device(context.device)
        if dev.type != "cuda":
            raise RuntimeError("the MI355X SONAR engine needs device='cuda[:i]' (no CPU path)")
        host = SpeechModelPipelineInterface()
        host.device = dev
        for hb in host._prefetched(list(paths), context.batch_size, context.n_parallel, context.n_prefetched_batches):
            yield host._fbank_batch(hb, context.pad_idx)

    def prebuild_pipeline(self, context: SpeechInferenceParams) -> DataPipelineBuilder:
        if context.batch_size <= 0:
            raise ValueError("`batch_size` should be strictly positive")
        names = read_tsv_column(context.data_file, context.audio_path_index)
        root = Path(context.audio_root_dir)
        paths = [root / n for n in names]

        def source():
            bs = context.batch_size
            buckets = (paths[i:i + bs] for i in range(0, len(paths), bs))
            for (batch, lens), bucket in zip(self._batches(context, paths), buckets):
                fb = batch.seqs
                if context.fbank_dtype != torch.float32:
                    fb = fb.to(context.fbank_dtype)
                t = fb.shape[1]
                yield {"audio": {"path": [str(p) for p in bucket],
                                 "data": {"fbank": {"seqs": fb, "seq_lens": torch.tensor(lens, dtype=torch.int64),
                                                    "is_ragged": any(l != t for l in lens)},
                                          "sample_rate": [16000.0] * len(bucket)}}}

        return DataPipelineBuilder(source)


class SpeechToEmbeddingPipeline(SpeechInferencePipeline):
    """speech.py:154-203.  `build_pipeline(ctx)` yields, per bucket, {"audio": {"path", "data": SonarEncoderOutput}}."""

    audio_to_fbank_dp_builder: AudioToFbankDataPipelineBuilder = AudioToFbankDataPipelineBuilder()
    model: SonarSpeechEncoderModel

    def __init__(self, model: SonarSpeechEncoderModel) -> None:
        self.model = model.eval()

    @classmethod
    def load_model_from_name(cls, encoder_name: str, device: Union[str, torch.device] = "cuda:0",
                             dtype: torch.dtype = torch.float32) -> "SpeechToEmbeddingPipeline":
        """A card name is resolved under $SONAR_CHECKPOINT_DIR (sonar_amd/cards.py); the reference loads on the CPU and
        moves the model in prebuild_pipeline -- the engine is created on its HIP device straight away."""
        return cls(model=load_sonar_speech_encoder(encoder_name, device=torch.device(device), dtype=dtype))

    def prebuild_pipeline(self, context: SpeechInferenceParams) -> DataPipelineBuilder:
        _same_device(self.model, context.device)
        return (self.audio_to_fbank_dp_builder.prebuild_pipeline(context)
                .map(lambda fbank: extract_sequence_batch(fbank, context.device), selector="audio.data.fbank")
                .map(self.run_inference, selector="audio.data"))

    @torch.inference_mode()
    def run_inference(self, data: dict):
        return self.model(data["fbank"])


class SpeechToTextPipeline(SpeechInferencePipeline):
    """speech.py:206-274: speech -> text translation; `build_pipeline(ctx)` yields {"audio": {"path", "data": [texts]}}."""

    audio_to_fbank_dp_builder: AudioToFbankDataPipelineBuilder = AudioToFbankDataPipelineBuilder()

    def __init__(self, model, tokenizer) -> None:
        self.model = model.eval()
        self.tokenizer = tokenizer

    @classmethod
    def load_model_from_name(cls, encoder_name: str, decoder_name: str, device: Union[str, torch.device] = "cuda:0",
                             dtype: torch.dtype = torch.float32) -> "SpeechToTextPipeline":
        from ..cards import resolve_tokenizer
        from ..text_decoder import SonarEncoderDecoderModel, load_sonar_text_decoder
        from ..tokenizer import NllbTokenizer

        device = torch.device(device)
        tokenizer = NllbTokenizer(resolve_tokenizer(decoder_name))
        encoder = load_sonar_speech_encoder(encoder_name, device=device, dtype=dtype)
        decoder = load_sonar_text_decoder(decoder_name, device=device, dtype=dtype)
        return cls(model=SonarEncoderDecoderModel(encoder, decoder).eval(), tokenizer=tokenizer)

    def prebuild_pipeline(self, context: SpeechInferenceParams) -> DataPipelineBuilder:
        assert context.target_lang is not None
        _same_device(self.model.decoder, context.device)
        from .text import EmbeddingToTextModelPipeline

        vec2t = EmbeddingToTextModelPipeline(self.model.decoder, self.tokenizer, device=context.device)

        @torch.inference_mode()
        def _do_generate(data: dict) -> List[str]:
            batch: SequenceBatch = data["fbank"]
            emb = self.model.encoder(batch).sentence_embeddings
            # fairseq2 caps the output at a * source_len + b with the (padded) fbank frame count as source length
            return vec2t.predict(emb, target_lang=context.target_lang, batch_size=emb.shape[0],
                                 source_len=batch.seqs.shape[1])

        return (self.audio_to_fbank_dp_builder.prebuild_pipeline(context)
                .map(lambda fbank: extract_sequence_batch(fbank, context.device), selector="audio.data.fbank")
                .map(_do_generate, selector="audio.data"))


def _same_device(model, device) -> None:
    """`self.model.to(context.device)` of the reference: an engine cannot move, so the devices must agree."""
    have = torch.device(getattr(model, "device", device))
    want = torch.device(device)
    if want.type == "cuda" and want.index is None:
        want = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    if have.type == "cuda" and have.index is None:
        have = torch.device("cuda", 0)
    if have != want:
        raise RuntimeError(f"context.device is {device} but the engine was created on {have}")
