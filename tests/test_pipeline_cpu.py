"""CPU: host-side logic of the pipeline mirror (tokenizer ids, bucketing, ordering,
errors and warnings) with a stand-in model object -- no compute kernels involved."""
import os
import warnings

import pytest
import torch

from sonar_amd.inference_pipelines.text import (TextToEmbeddingModelPipeline, collate, dynamic_bucket)
from sonar_amd.nllb_langs import NLLB_EXTRA_CONTROL, NLLB_LANGS
from sonar_amd.text_encoder import SonarEncoderOutput, convert_sonar_text_encoder_checkpoint
from sonar_amd.tokenizer import NllbTokenizer

REF_CARD = "/root/reference/sonar/cards/text_sonar_basic_encoder.yaml"


@pytest.fixture(scope="module")
def spm_model(tmp_path_factory):
    import sentencepiece as spm

    d = tmp_path_factory.mktemp("spm")
    corpus = d / "corpus.txt"
    words = ["hello", "world", "my", "name", "is", "paul", "teacher", "working", "bonjour", "monde",
             "je", "travaille", "comme", "professeur", "the", "quick", "brown", "fox", "jumps", "over"]
    g = torch.Generator().manual_seed(0)
    with open(corpus, "w") as fh:
        for _ in range(400):
            n = int(torch.randint(2, 12, (1,), generator=g))
            fh.write(" ".join(words[int(i)] for i in torch.randint(0, len(words), (n,), generator=g)) + "\n")
    prefix = str(d / "toy")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=prefix, vocab_size=48, model_type="unigram", hard_vocab_limit=False,
                                   bos_id=1, eos_id=2, unk_id=0, pad_id=-1, minloglevel=2)
    return prefix + ".model"


def test_lang_token_ids_match_reference_facts():
    # SURVEY a14 / notebook cell 44: eng_Latn -> 256047, fra_Latn -> 256057, vocab 256206
    assert 256001 + NLLB_LANGS.index("eng_Latn") == 256047
    assert 256001 + NLLB_LANGS.index("fra_Latn") == 256057
    assert 256001 + len(NLLB_LANGS) + len(NLLB_EXTRA_CONTROL) == 256206


@pytest.mark.skipif(not os.path.exists(REF_CARD), reason="reference tree not present")
def test_lang_order_matches_reference_card():
    import yaml

    card = yaml.safe_load(open(REF_CARD))
    assert card["langs"] == NLLB_LANGS


def test_lang_order_matches_hf_fairseq_language_codes():
    """The same order from an independent source: HuggingFace's FAIRSEQ_LANGUAGE_CODES
    (tests/golden/make_golden_langs.py), which travels to machines without the reference tree."""
    import json

    hf = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "nllb_language_codes.json")))
    assert hf == NLLB_LANGS and len(hf) == 202


def test_tokenizer_layout(spm_model):
    tok = NllbTokenizer(spm_model)
    assert tok.vocab_info.pad_idx == 0 and tok.vocab_info.eos_idx == 3
    enc = tok.create_encoder(lang="fra_Latn")
    ids = enc("bonjour monde")
    assert ids.dtype == torch.int64
    assert ids[0].item() == tok.lang_idx("fra_Latn") and ids[-1].item() == 3
    assert all(4 <= i < tok.lang_base for i in ids[1:-1].tolist())
    assert tok.decode(ids) == "bonjour monde"
    tgt = tok.create_encoder(lang="eng_Latn", mode="target")
    assert tgt("hello").tolist()[:2] == [3, tok.lang_idx("eng_Latn")]
    assert enc.encode_batch(["bonjour monde", "je travaille"])[0] == ids.tolist()
    with pytest.raises(ValueError):
        tok.create_encoder(lang="xx_Nope")


def test_dynamic_bucket_and_collate():
    seqs = [torch.arange(n) for n in (3, 4, 5, 1, 2, 9)]
    assert [len(b) for b in dynamic_bucket(iter(seqs), 2**31, 4)] == [4, 2]
    # threshold 7 tokens: [3,4] closes at 7, [5,1,2] closes at 8, [9]
    assert [[len(s) for s in b] for b in dynamic_bucket(iter(seqs), 7, 100)] == [[3, 4], [5, 1, 2], [9]]
    c = collate(seqs[:3], 0)
    assert c["seqs"].shape == (3, 5) and c["is_ragged"] and c["seq_lens"].tolist() == [3, 4, 5]
    assert c["seqs"][0].tolist() == [0, 1, 2, 0, 0]
    assert not collate([torch.arange(4), torch.arange(4)], 0)["is_ragged"]


class _StubEncoder:
    """Deterministic stand-in: embedding = [sum of ids, length] (host logic tests only)."""

    model_dim = 2
    dtype = torch.float32
    device = torch.device("cpu")

    class _F:
        class _P:
            max_seq_len = 16
        pos_encoder = _P()
    encoder_frontend = _F()

    def __init__(self):
        self.batches = []

    def eval(self):
        return self

    def __call__(self, batch):
        seqs = batch.seqs
        lens = batch.padding_mask.seq_lens if batch.padding_mask is not None else torch.full((seqs.shape[0],), seqs.shape[1])
        self.batches.append((tuple(seqs.shape), batch.padding_mask is None))
        mask = torch.arange(seqs.shape[1]).unsqueeze(0) < lens.unsqueeze(1)
        emb = torch.stack([(seqs * mask).sum(1).float(), lens.float()], dim=1)
        return SonarEncoderOutput(None, emb, batch.padding_mask)


def test_predict_order_bucketing_and_truncation(spm_model, tmp_path):
    tok = NllbTokenizer(spm_model)
    stub = _StubEncoder()
    pipe = TextToEmbeddingModelPipeline(stub, tok, device=torch.device("cpu"))
    texts = ["hello world my name is paul", "hello", "the quick brown fox jumps over the teacher", "bonjour monde"]
    enc = tok.create_encoder(lang="eng_Latn")
    expect = torch.tensor([[float(enc(t).sum()), float(len(enc(t)))] for t in texts])
    out = pipe.predict(texts, source_lang="eng_Latn", batch_size=2)
    assert torch.equal(out, expect)  # input order restored after the length sort
    assert len(stub.batches) == 2
    # batching invariance of the host path (reference: test_text_sonar.py:120-161)
    for kw in (dict(batch_size=1), dict(batch_size=None, batch_max_tokens=5), dict(batch_max_tokens=30)):
        assert torch.equal(pipe.predict(texts, source_lang="eng_Latn", **kw), expect)
    # file input keeps file order, no sorting
    f = tmp_path / "in.txt"
    f.write_text("\n".join(texts) + "\n")
    assert torch.equal(pipe.predict(f, source_lang="eng_Latn", batch_size=3), expect)
    # CRLF files: the line ending is stripped, nothing else (fairseq2 read_text infers the ending, rtrim=False);
    # a missing final newline and trailing spaces are kept as they are
    g = tmp_path / "crlf.txt"
    g.write_bytes("\r\n".join(texts).encode())
    assert torch.equal(pipe.predict(g, source_lang="eng_Latn", batch_size=3), expect)
    # empty input -> empty [0, d] matrix, no model call
    n_before = len(stub.batches)
    stub.model_dim, stub.dtype = 2, torch.float32
    empty = pipe.predict([], source_lang="eng_Latn")
    assert empty.shape[0] == 0 and len(stub.batches) == n_before
    # an empty STRING is a sentence: [lang, </s>]
    one = pipe.predict([""], source_lang="eng_Latn")
    assert one.shape == (1, 2) and one[0, 1].item() == 2
    # truncation warns and clips to the model maximum (reference: test_text_sonar.py:55-59)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        long_out = pipe.predict(["hello " * 100], source_lang="eng_Latn")
    assert any("truncated to 16" in str(x.message) for x in w)
    assert long_out[0, 1].item() == 16


def test_predict_argument_errors(spm_model):
    pipe = TextToEmbeddingModelPipeline(_StubEncoder(), NllbTokenizer(spm_model), device=torch.device("cpu"))
    with pytest.raises(ValueError, match="at least one of"):
        pipe.predict(["a"], "eng_Latn", batch_size=None, batch_max_tokens=None)
    with pytest.raises(ValueError, match="strictly positive"):
        pipe.predict(["a"], "eng_Latn", batch_size=0)
    with pytest.raises(ValueError, match="strictly positive"):
        pipe.predict(["a"], "eng_Latn", batch_max_tokens=-1)
    with pytest.raises(ValueError, match="max_seq_len cannot be larger"):
        pipe.predict(["a"], "eng_Latn", max_seq_len=17)
    with pytest.raises(RuntimeError, match="no CPU path"):
        TextToEmbeddingModelPipeline("/nonexistent/ckpt.pt", NllbTokenizer(spm_model), device=torch.device("cpu"))


def test_checkpoint_conversion_layouts():
    # fairseq1 layout -> renamed keys + control-token row permutation (handler.py:71-92)
    emb = torch.arange(24, dtype=torch.float32).reshape(6, 4)
    sd = {"embed_tokens.weight": emb.clone(), "layers.0.fc1.weight": torch.ones(2, 2),
          "layers.0.self_attn.out_proj.bias": torch.zeros(2), "layers.0.final_layer_norm.weight": torch.ones(2),
          "layer_norm.weight": torch.ones(2), "version": torch.tensor([1.0]),
          "embed_positions._float_tensor": torch.zeros(1)}
    out = convert_sonar_text_encoder_checkpoint({"state_dict": sd})
    assert set(out) == {"encoder_frontend.embed.weight", "encoder.layers.0.ffn.inner_proj.weight",
                        "encoder.layers.0.self_attn.output_proj.bias", "encoder.layers.0.ffn_layer_norm.weight",
                        "layer_norm.weight"}
    assert torch.equal(out["encoder_frontend.embed.weight"][:4], emb[[1, 3, 0, 2]])
    assert torch.equal(out["encoder_frontend.embed.weight"][4:], emb[4:])
    assert torch.equal(sd["embed_tokens.weight"], emb)  # caller's tensor untouched
    # fairseq2 layout passes through
    m = {"model": {"encoder_frontend.embed.weight": emb}}
    assert convert_sonar_text_encoder_checkpoint(m)["encoder_frontend.embed.weight"] is emb
    with pytest.raises(ValueError):
        convert_sonar_text_encoder_checkpoint({"weights": {}})


def test_native_host_path_equals_python_path(spm_model):
    """The C++ host input path (sonar_amd/host_input.py + smi_host_*) must produce the same batches
    -- shapes, padding masks, ids, order, truncation count -- as the per-sentence restatement."""
    tok = NllbTokenizer(spm_model)
    words = ["hello", "world", "my", "name", "is", "paul", "teacher", "bonjour", "monde", "fox", "jumps"]
    g = torch.Generator().manual_seed(3)
    texts = [" ".join(words[int(i)] for i in torch.randint(0, len(words), (int(torch.randint(1, 14, (1,), generator=g)),),
                                                           generator=g)) for _ in range(137)]
    texts[5] = ""          # empty sentence: [lang, eos] only
    for kw in (dict(batch_size=8), dict(batch_size=None, batch_max_tokens=40), dict(batch_size=16, batch_max_tokens=64),
               dict(batch_size=1000), dict(batch_size=8, max_seq_len=9)):
        got = {}
        for mode in ("native", "python"):
            stub = _StubEncoder()
            pipe = TextToEmbeddingModelPipeline(stub, tok, device=torch.device("cpu"))
            pipe.host_input = mode
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                out = pipe.predict(texts, source_lang="fra_Latn", **kw)
            got[mode] = (out, stub.batches, sorted(str(x.message) for x in w))
        assert torch.equal(got["native"][0], got["python"][0]), kw
        assert got["native"][1] == got["python"][1], kw
        assert got["native"][2] == got["python"][2], kw
    # the native path crosses its tokenisation-chunk boundary with an open bucket
    from sonar_amd.host_input import iter_text_batches

    enc = tok.create_encoder(lang="eng_Latn")
    a = list(iter_text_batches(texts, range(len(texts)), enc, max_seq_len=16, batch_size=7, batch_max_tokens=None,
                               pad_idx=0, device=torch.device("cpu"), chunk=10))
    b = list(iter_text_batches(texts, range(len(texts)), enc, max_seq_len=16, batch_size=7, batch_max_tokens=None,
                               pad_idx=0, device=torch.device("cpu"), chunk=10_000))
    assert len(a) == len(b) == (len(texts) + 6) // 7
    for x, y in zip(a, b):
        assert torch.equal(x.seqs, y.seqs) and (x.padding_mask is None) == (y.padding_mask is None)


def test_host_abi_functions():
    import ctypes as C

    import numpy as np

    from sonar_amd import _lib

    lib = _lib.load()
    off = np.array([0, 3, 3, 10], dtype=np.int64)          # piece counts 3, 0, 7
    lens = np.zeros(3, dtype=np.int32)
    cut = C.c_int64(-1)
    assert lib.smi_host_token_lengths(off.ctypes.data, 3, 1, 1, 6, lens.ctypes.data, C.byref(cut)) == 0
    assert lens.tolist() == [5, 2, 6] and cut.value == 1
    pieces = np.arange(100, 110, dtype=np.int32)
    out = np.full((3, 7), -7, dtype=np.int64)
    pre, suf = np.array([900], dtype=np.int64), np.array([3], dtype=np.int64)
    assert lib.smi_host_collate_nllb(pieces.ctypes.data, off.ctypes.data, lens.ctypes.data, 0, 3, pre.ctypes.data, 1,
                                     suf.ctypes.data, 1, 1, 0, out.ctypes.data, 7, 4) == 0
    assert out.tolist() == [[900, 101, 102, 103, 3, 0, 0], [900, 3, 0, 0, 0, 0, 0],
                            [900, 104, 105, 106, 107, 108, 0]]  # third row truncated to 6: its EOS is lost
    # row_stride smaller than a sequence is refused
    assert lib.smi_host_collate_nllb(pieces.ctypes.data, off.ctypes.data, lens.ctypes.data, 0, 3, pre.ctypes.data, 1,
                                     suf.ctypes.data, 1, 1, 0, out.ctypes.data, 5, 1) != 0
    bl = np.array([3, 4, 5, 1, 2, 9], dtype=np.int32)
    bounds = np.zeros(7, dtype=np.int64)
    nb, nopen = C.c_int64(0), C.c_int64(0)
    assert lib.smi_host_dynamic_bucket(bl.ctypes.data, 6, 7, 100, 1, bounds.ctypes.data, C.byref(nb), C.byref(nopen)) == 0
    assert bounds[:nb.value + 1].tolist() == [0, 2, 5, 6] and nopen.value == 0
    assert lib.smi_host_dynamic_bucket(bl.ctypes.data, 6, 2**31, 4, 1, bounds.ctypes.data, C.byref(nb), C.byref(nopen)) == 0
    assert bounds[:nb.value + 1].tolist() == [0, 4] and nopen.value == 2


def test_tsv_speech_pipeline_plumbing(tmp_path):
    """The host plumbing of the TSV-driven speech pipelines (sonar/inference_pipelines/speech.py:42-274) without a
    GPU: parameter object, TSV column selection (header skipped, right-trimmed), bucketing, the nested element
    structure with its `selector` maps, re-iteration, argument errors.  The device stages are stubbed."""
    import dataclasses

    from sonar_amd.inference_pipelines import (AudioToFbankDataPipelineBuilder, SpeechInferenceParams,
                                               SpeechToEmbeddingPipeline)
    from sonar_amd.inference_pipelines.speech import DataPipelineBuilder, read_tsv_column
    from sonar_amd.text_encoder import PaddingMask, SequenceBatch

    # the reference's dataclass: same fields, order and defaults (speech.py:42-73)
    fields = [(f.name, f.default) for f in dataclasses.fields(SpeechInferenceParams)]
    assert [n for n, _ in fields] == ["data_file", "audio_root_dir", "audio_path_index", "batch_size", "fbank_dtype",
                                      "target_lang", "pad_idx", "device", "n_parallel", "n_prefetched_batches"]
    assert dict(fields[4:]) == {"fbank_dtype": torch.float32, "target_lang": None, "pad_idx": 0,
                                "device": torch.device("cpu"), "n_parallel": 4, "n_prefetched_batches": 4}
    tsv = tmp_path / "ref.tsv"
    tsv.write_text("id\taudio\tnote\n1\ta.wav\tx  \n2\tb.wav\ty\n\n3\tc.wav\tz\n", encoding="utf-8")
    assert read_tsv_column(tsv, 1) == ["a.wav", "b.wav", "c.wav"]
    assert read_tsv_column(tsv, 2) == ["x", "y", "z"]
    with pytest.raises(ValueError, match="no column 5"):
        read_tsv_column(tsv, 5)

    seen = []

    class StubFbank(AudioToFbankDataPipelineBuilder):
        def _batches(self, context, paths):
            for i in range(0, len(paths), context.batch_size):
                chunk = paths[i:i + context.batch_size]
                seen.append([p.name for p in chunk])
                lens = [4 + 2 * j for j in range(len(chunk))]
                fb = torch.zeros(len(chunk), max(lens), 80)
                for j, l in enumerate(lens):
                    fb[j, :l] = float(i + j + 1)
                yield SequenceBatch(fb, PaddingMask(torch.tensor(lens), max(lens))), lens

    class StubModel:
        device = torch.device("cpu")

        def eval(self):
            return self

        def __call__(self, batch):
            from sonar_amd.text_encoder import SonarEncoderOutput

            lens = batch.padding_mask.seq_lens if batch.padding_mask is not None else None
            pooled = batch.seqs.sum(dim=1)[:, :2] / (lens.unsqueeze(1) if lens is not None else batch.seqs.shape[1])
            return SonarEncoderOutput(None, pooled, batch.padding_mask)

    ctx = SpeechInferenceParams(data_file=tsv, audio_root_dir=tmp_path / "clips", audio_path_index=1, batch_size=2)
    pipe = SpeechToEmbeddingPipeline(StubModel())
    pipe.audio_to_fbank_dp_builder = StubFbank()
    dp = pipe.build_pipeline(ctx)
    for _ in range(2):                                  # a DataPipeline can be iterated again
        seen.clear()
        items = list(dp)
        assert seen == [["a.wav", "b.wav"], ["c.wav"]]
        assert [it["audio"]["path"] for it in items] == [[str(tmp_path / "clips" / "a.wav"), str(tmp_path / "clips" / "b.wav")],
                                                        [str(tmp_path / "clips" / "c.wav")]]
        emb = torch.cat([it["audio"]["data"].sentence_embeddings for it in items])
        assert torch.allclose(emb[:, 0], torch.tensor([1.0, 2.0, 3.0]) * 80 / 80)
    # the fbank stage alone yields the collated dict of the reference (seqs / seq_lens / is_ragged)
    first = next(iter(StubFbank().build_pipeline(ctx)))
    fbd = first["audio"]["data"]["fbank"]
    assert fbd["seqs"].shape == (2, 6, 80) and fbd["seq_lens"].tolist() == [4, 6] and fbd["is_ragged"] is True
    assert first["audio"]["data"]["sample_rate"] == [16000.0, 16000.0]
    # selector maps compose on the builder, as the reference's pipelines do
    b = DataPipelineBuilder(lambda: iter([{"a": {"b": 1}}, {"a": {"b": 2}}])).map(lambda v: v + 10, selector="a.b")
    assert [x["a"]["b"] for x in b.and_return()] == [11, 12]
    with pytest.raises(ValueError, match="batch_size"):
        StubFbank().build_pipeline(SpeechInferenceParams(tsv, tmp_path, 1, 0))
    with pytest.raises(RuntimeError, match="no CPU path"):   # the real device stage refuses a CPU device
        next(iter(AudioToFbankDataPipelineBuilder().build_pipeline(ctx)))
