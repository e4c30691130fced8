"""TextToEmbeddingModelPipeline with the reference's interface
(sonar/inference_pipelines/text.py:140-269), running the model on the MI355X
engine.  Same argument names, checks, warnings and output order; the fairseq2
DataPipeline (read -> tokenize -> truncate -> dynamic_bucket -> Collater(pad) ->
to-device -> prefetch(2) -> model) is restated with a plain background thread.
"""
from __future__ import annotations

import contextlib

import queue
import threading
import warnings
from contextlib import contextmanager
from pathlib import Path
from typing import Iterable, Iterator, List, Optional, Sequence, Union

import torch

from ..text_encoder import SonarTextTransformerEncoderModel, load_sonar_text_encoder
from ..tokenizer import NllbEncoder, NllbTokenizer
from .utils import add_progress_bar, extract_sequence_batch

CPU = torch.device("cpu")


@contextmanager
def precision_context(dtype: torch.dtype) -> Iterator[None]:
    """sonar/inference_pipelines/text.py:36-54 (kept for interface parity; the
    engine's own kernels do not consult torch's matmul precision)."""
    if dtype in (torch.float16, torch.bfloat16):
        precision = "medium"
    elif dtype == torch.float32:
        precision = "high"
    elif dtype == torch.float64:
        precision = "highest"
    else:
        raise ValueError("Unsupported dtype for precision context")
    old = torch.get_float32_matmul_precision()
    torch.set_float32_matmul_precision(precision)
    try:
        yield
    finally:
        torch.set_float32_matmul_precision(old)


def dynamic_bucket(items: Iterable[torch.Tensor], threshold: int, max_num_examples: int,
                   min_num_examples: int = 1) -> Iterator[List[torch.Tensor]]:
    """fairseq2 `.dynamic_bucket(threshold, cost_fn=len, min_num_examples, max_num_examples,
    drop_remainder=False)` as used at text.py:234-240: a bucket closes once its summed
    length reaches the threshold (and it holds at least `min_num_examples`) or it holds
    `max_num_examples` sequences."""
    bucket: List[torch.Tensor] = []
    cost = 0
    for it in items:
        bucket.append(it)
        cost += len(it)
        if (cost >= threshold and len(bucket) >= min_num_examples) or len(bucket) >= max_num_examples:
            yield bucket
            bucket, cost = [], 0
    if bucket:
        yield bucket


def collate(seqs: Sequence[torch.Tensor], pad_value: int) -> dict:
    """fairseq2 Collater(pad_value): right-pad to the batch maximum."""
    lens = torch.tensor([len(s) for s in seqs], dtype=torch.int32)
    s_max = int(lens.max()) if len(seqs) else 0
    out = torch.full((len(seqs), s_max), pad_value, dtype=torch.int64)
    for i, s in enumerate(seqs):
        out[i, : len(s)] = s
    return {"seqs": out, "seq_lens": lens, "is_ragged": bool((lens != s_max).any())}


def _prefetch(it: Iterator, depth: int) -> Iterator:
    """`.prefetch(depth)`: run the upstream stages on a background thread.  If the consumer stops early (an
    exception in the model, a closed generator) the producer is told to stop instead of blocking on a full queue."""
    q: "queue.Queue" = queue.Queue(maxsize=max(depth, 1))
    end = object()
    stop = threading.Event()

    def offer(x) -> bool:
        while not stop.is_set():
            try:
                q.put(x, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for x in it:
                if not offer(x):
                    return
            offer(end)
        except BaseException as e:  # propagate to the consumer
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


class TextToEmbeddingModelPipeline(torch.nn.Module):
    model: SonarTextTransformerEncoderModel
    tokenizer: NllbTokenizer

    def __init__(self, encoder: Union[str, Path, SonarTextTransformerEncoderModel],
                 tokenizer: Union[str, Path, NllbTokenizer],
                 device: torch.device = CPU, dtype: Optional[torch.dtype] = None) -> None:
        """
        Args:
            encoder: a card name (resolved under $SONAR_CHECKPOINT_DIR), a checkpoint path (fairseq or
                fairseq2 layout, `basic` arch) or a model object
            tokenizer: a card name, a SentencePiece model path or a tokenizer object
            device: the HIP device to run on.  The reference defaults to CPU; this engine has no
                CPU path, so a CPU device raises here instead of silently running elsewhere.
            dtype: as in the reference (`hub.load(name, device=device, dtype=dtype)`, text.py:161-162):
                None loads an fp32 model -- fp32 embeddings and an fp32 residual stream (the GEMM operands
                are fp16 on this engine either way); torch.float16 is the fast path BASELINE measures.
        """
        super().__init__()
        device = torch.device(device)
        if isinstance(encoder, (str, Path)):
            if device.type != "cuda":
                raise RuntimeError("the MI355X SONAR engine needs device='cuda[:i]' (no CPU path)")
            encoder = load_sonar_text_encoder(str(encoder), device=device, dtype=dtype or torch.float32)
        if isinstance(tokenizer, (str, Path)):
            from ..cards import resolve_tokenizer

            tokenizer = NllbTokenizer(resolve_tokenizer(tokenizer))
        self.tokenizer = tokenizer
        self.model = encoder.eval()
        self.device = getattr(encoder, "device", device)
        self.dtype = dtype
        # "native": sonar_amd.host_input (C++ host threads); "python": the per-sentence restatement
        # of the reference pipeline below (any tokenizer object with create_encoder())
        self.host_input = "native"

    @torch.inference_mode()
    def predict(self, input: Union[Path, Sequence[str]], source_lang: str,
                batch_size: Optional[int] = 5, batch_max_tokens: Optional[int] = None,
                max_seq_len: Optional[int] = None, progress_bar: bool = False,
                target_device: Optional[torch.device] = None) -> torch.Tensor:
        """Transform the input texts (a list of strings or a text file) into a matrix of their
        embeddings; texts are truncated to `max_seq_len` tokens or to the model maximum."""
        if batch_max_tokens is None and batch_size is None:
            raise ValueError("at least one of `batch_size` or `batch_max_tokens` should be provided")
        if batch_max_tokens is not None and batch_max_tokens <= 0:
            raise ValueError("`batch_max_tokens` should be strictly positive")
        if batch_size is not None and batch_size <= 0:
            raise ValueError("`batch_size` should be strictly positive")

        tokenizer_encoder = self.tokenizer.create_encoder(lang=source_lang)
        model_vocab = getattr(getattr(getattr(self.model, "config", None), "vocab_info", None), "size", None)
        tok_vocab = getattr(getattr(self.tokenizer, "vocab_info", None), "size", None)
        if model_vocab is not None and tok_vocab is not None and tok_vocab > model_vocab:
            raise ValueError(f"the tokenizer's vocabulary ({tok_vocab}) is larger than the encoder's embedding "
                             f"table ({model_vocab}): tokenizer and model do not belong together")
        model_max_len = self.model.encoder_frontend.pos_encoder.max_seq_len
        if max_seq_len is None:
            max_seq_len = model_max_len
        if max_seq_len is not None and model_max_len is not None and max_seq_len > model_max_len:
            raise ValueError(f"max_seq_len cannot be larger than max_seq_len of the encoder model: {model_max_len}")

        n_truncated = 0

        def truncate(x: torch.Tensor) -> torch.Tensor:
            nonlocal n_truncated
            if max_seq_len is None:
                return x
            if x.shape[0] > max_seq_len:
                n_truncated += 1
            return x[:max_seq_len]

        if isinstance(input, (str, Path)):
            with open(Path(input), "r", encoding="utf-8") as fh:
                texts: Sequence[str] = [line.rstrip("\n") for line in fh]
            order: Iterable[int] = range(len(texts))
            sorting_index = None
        else:
            texts = input
            # sort by CHARACTER length, as the reference does (text.py:226)
            sorting_index = torch.argsort(torch.tensor(list(map(len, texts)), dtype=torch.int64))
            order = sorting_index.tolist()

        pad_idx = self.tokenizer.vocab_info.pad_idx
        dev = self.device

        stats = {"n_truncated": 0}
        native = self.host_input == "native" and isinstance(tokenizer_encoder, NllbEncoder)

        def upstream():
            if native:  # C++ host path: batch SentencePiece + threaded assembly/bucketing/collation
                from ..host_input import iter_text_batches

                yield from iter_text_batches(texts, order, tokenizer_encoder, max_seq_len=max_seq_len,
                                             batch_size=batch_size, batch_max_tokens=batch_max_tokens,
                                             pad_idx=pad_idx, device=torch.device(dev), stats=stats)
                return
            toks = (truncate(tokenizer_encoder(texts[i])) for i in order)
            for bucket in dynamic_bucket(toks, batch_max_tokens or 2**31, batch_size or 20_000):
                yield extract_sequence_batch(collate(bucket, pad_idx), dev)

        pipeline: Iterable = _prefetch(upstream(), 2)
        if progress_bar:
            pipeline = add_progress_bar(pipeline, inputs=texts,
                                        batch_size=batch_size if batch_max_tokens is None else None)
        results: List[torch.Tensor] = []
        # the engine's model object checks for out-of-vocabulary ids inside every forward() (one stream
        # synchronisation each); this loop queues its batches and checks once, when the last one has been launched
        defer = getattr(self.model, "deferring_check", None)
        with precision_context(self.model.dtype), (defer() if defer is not None else contextlib.nullcontext()):
            for batch in pipeline:
                out = self.model(batch)
                results.append(out.sentence_embeddings.to(target_device or self.device))
            # leaving the block: out-of-vocabulary ids raise IndexError here, as the reference's embedding does
        n_truncated += stats["n_truncated"]
        if n_truncated:
            warnings.warn(f"For {n_truncated} input tensors for SONAR text encoder, "
                          f"the length was truncated to {max_seq_len} elements.")
        if not results:
            return torch.empty((0, self.model.model_dim), dtype=self.model.dtype, device=target_device or self.device)
        sentence_embeddings = torch.cat(results, dim=0)
        if sorting_index is not None:
            reversed_index = torch.argsort(sorting_index)
            sentence_embeddings = sentence_embeddings[reversed_index.to(sentence_embeddings.device)]
        return sentence_embeddings


class EmbeddingToTextModelPipeline(torch.nn.Module):
    """sonar/inference_pipelines/text.py:270-346 on the MI355X engine: embeddings -> texts by
    beam search (fairseq2 BeamSearchSeq2SeqGenerator defaults; `generator_kwargs` accepts
    beam_size, min_gen_len, max_gen_len, max_seq_len, normalize_scores, len_penalty,
    unk_penalty, temperature), or -- with `sampler=TopKSampler(k) / TopPSampler(p)` (sonar_amd.generation) -- by
    fairseq2's SamplingSeq2SeqGenerator (same kwargs minus beam_size)."""

    def __init__(self, decoder, tokenizer: Union[str, Path, NllbTokenizer], device: torch.device = CPU,
                 dtype: Optional[torch.dtype] = None) -> None:
        super().__init__()
        from ..text_decoder import ConditionalTransformerDecoderModel, load_sonar_text_decoder

        device = torch.device(device)
        if isinstance(decoder, (str, Path)):
            if device.type != "cuda":
                raise RuntimeError("the MI355X SONAR engine needs device='cuda[:i]' (no CPU path)")
            decoder = load_sonar_text_decoder(str(decoder), device=device, dtype=dtype or torch.float32)
        if isinstance(tokenizer, (str, Path)):
            from ..cards import resolve_tokenizer

            tokenizer = NllbTokenizer(resolve_tokenizer(tokenizer))
        self.tokenizer = tokenizer
        self.model = decoder.eval()
        self.device = getattr(decoder, "device", device)

    @torch.inference_mode()
    def predict(self, inputs: torch.Tensor, target_lang: str, batch_size: int = 5, progress_bar: bool = False,
                sampler=None, **generator_kwargs) -> List[str]:
        if batch_size <= 0:
            raise ValueError("`batch_size` should be strictly positive")
        prompt = self.tokenizer.create_encoder(task="translation", lang=target_lang, mode="target").prefix
        decode = self.tokenizer.create_decoder()
        rows = list(inputs)
        batches: Iterable = [rows[i:i + batch_size] for i in range(0, len(rows), batch_size)]
        if progress_bar:
            batches = add_progress_bar(batches, inputs=rows, batch_size=batch_size)
        texts: List[str] = []
        for chunk in batches:
            emb = torch.stack(chunk).to(self.device)
            if sampler is None:   # text.py:315-320: beam search by default, a sampling generator otherwise
                toks, lens, _ = self.model.engine.generate(emb, prompt, **generator_kwargs)
                toks, lens = toks[:, 0].cpu(), lens[:, 0].cpu()
            else:
                toks, lens, _ = self.model.engine.sample(emb, prompt, sampler, sentence_offset=len(texts),
                                                         **generator_kwargs)
                toks, lens = toks.cpu(), lens.cpu()
            for i in range(emb.shape[0]):
                texts.append(decode(toks[i, : int(lens[i])]))
        return texts


class TextToTextModelPipeline(torch.nn.Module):
    """sonar/inference_pipelines/text.py:56-137: text -> sentence vector (encoder engine) -> text
    (decoder engine, beam search).  `max_seq_len` is clamped to the decoder's positional range as the
    reference does (text.py:104-107)."""

    def __init__(self, encoder, decoder, tokenizer: Union[str, Path, NllbTokenizer], device: torch.device = CPU,
                 dtype: Optional[torch.dtype] = None) -> None:
        super().__init__()
        self.t2vec = TextToEmbeddingModelPipeline(encoder, tokenizer, device=device, dtype=dtype)
        self.vec2t = EmbeddingToTextModelPipeline(decoder, self.t2vec.tokenizer, device=device, dtype=dtype)
        self.tokenizer = self.t2vec.tokenizer
        self.device = self.t2vec.device

    @torch.inference_mode()
    def predict(self, input: Union[Path, Sequence[str]], source_lang: str, target_lang: str, batch_size: int = 5,
                progress_bar: bool = False, **generator_kwargs) -> List[str]:
        model_max = self.vec2t.model.max_target_seq_len
        generator_kwargs["max_seq_len"] = min(model_max, generator_kwargs.get("max_seq_len", model_max))
        if isinstance(input, (str, Path)):
            with open(Path(input), "r", encoding="utf-8") as fh:
                input = [line.rstrip("\n") for line in fh]
        texts = list(input)
        out: List[str] = []
        batches: Iterable = [texts[i:i + batch_size] for i in range(0, len(texts), batch_size)]
        if progress_bar:
            batches = add_progress_bar(batches, inputs=texts, batch_size=batch_size)
        enc = self.tokenizer.create_encoder(lang=source_lang)
        for chunk in batches:
            emb = self.t2vec.predict(chunk, source_lang=source_lang, batch_size=len(chunk))
            # fairseq2's generator caps the output at a * max_source_len + b tokens, the source length
            # being the longest tokenised source of the batch the translator built (text.py:109-120)
            if hasattr(enc, "encode_batch"):
                src_len = max(len(t) for t in enc.encode_batch(chunk))
            else:
                src_len = max(len(enc(t)) for t in chunk)
            src_len = min(src_len, self.t2vec.model.encoder_frontend.pos_encoder.max_seq_len)
            out.extend(self.vec2t.predict(emb, target_lang=target_lang, batch_size=len(chunk),
                                          source_len=src_len, **generator_kwargs))
        return out
