"""Checkpoint ingest on the CPU: the three converters against independently written
fairseq1 layouts, `torch.save` round trips, and the packed cache."""
import torch

from tests.ckpt_layouts import speech_encoder_to_fairseq1, text_decoder_to_fairseq1, text_encoder_to_fairseq1


def _same(a, b):
    assert set(a) == set(b), (sorted(set(a) - set(b)), sorted(set(b) - set(a)))
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_speech_checkpoint_conversion(tmp_path):
    """sonar_speech/handler.py:46-110: every fairseq1 key lands on the fairseq2 name the engine reads,
    mask_emb / pos_conv are dropped (:55-61), the post-conformer LayerNorm moves to the model level
    (:102-108), decoder.embed_out becomes the pooler's projection (:98)."""
    from oracle import speech_encoder as OS
    from sonar_amd.speech_encoder import convert_sonar_speech_checkpoint

    so = OS.OracleSpeechEncoderConfig(model_dim=64, num_layers=2, num_heads=1, ffn_inner_dim=128, conv_kernel=7,
                                      pooler_layers=2, pooler_heads=1, pooler_ffn_dim=96, pooler_vocab=16)
    want = OS.make_synthetic_params(so, seed=5, std=0.06)
    ck = speech_encoder_to_fairseq1(want)
    assert "encoder.w2v_model.mask_emb" in ck["model"] and "encoder.w2v_model.encoder.layer_norm.weight" in ck["model"]
    f = tmp_path / "spenc.pt"
    torch.save(ck, f)
    got = convert_sonar_speech_checkpoint(torch.load(f, weights_only=False))
    assert not any("mask_emb" in k or "pos_conv" in k for k in got)
    assert "layer_norm.weight" in got and not any(k.startswith("encoder.layer_norm") for k in got)
    assert torch.equal(got["encoder_pooler.projection_out.weight"], want["encoder_pooler.projection_out.weight"])
    extra = {k for k in got if k.endswith("num_batches_tracked")}
    _same({k: v for k, v in got.items() if k not in extra}, want)
    assert not any(k.startswith(("encoder.w2v_model", "decoder.")) for k in got), "unmapped fairseq1 keys"
    # a fairseq2 checkpoint passes through untouched (handler.py:52-54)
    _same(convert_sonar_speech_checkpoint({"model": dict(want)}), want)


def test_text_encoder_checkpoint_conversion(tmp_path):
    from oracle import text_encoder as OE
    from sonar_amd.text_encoder import convert_sonar_text_encoder_checkpoint

    cfg = OE.OracleTextEncoderConfig(model_dim=64, num_layers=2, num_heads=1, ffn_inner_dim=128, vocab_size=50)
    want = OE.make_synthetic_params(cfg, seed=1)
    f = tmp_path / "enc.pt"
    torch.save(text_encoder_to_fairseq1(want), f)
    got = convert_sonar_text_encoder_checkpoint(torch.load(f, weights_only=False))
    _same(got, want)                       # incl. the (BOS,PAD,EOS,UNK) -> (PAD,UNK,BOS,EOS) row permutation
    assert "version" not in got and "embed_positions._float_tensor" not in got
    _same(convert_sonar_text_encoder_checkpoint({"model": dict(want)}), want)


def test_text_decoder_checkpoint_conversion_and_tying(tmp_path):
    """handler.py:122-172 + the tied projection (factory.py:306-307, test_tied_weights.py:40-78): after
    loading there is ONE table -- the permuted embedding -- for both the input embedding and final_proj."""
    from oracle import text_decoder as OD
    from sonar_amd.text_decoder import convert_sonar_text_decoder_checkpoint

    cfg = OD.OracleTextDecoderConfig(model_dim=64, num_layers=2, num_heads=1, ffn_inner_dim=128, vocab_size=50, max_seq_len=16)
    want = OD.make_synthetic_params(cfg, seed=2, std=0.1)
    for tied_storage in (True, False):
        f = tmp_path / f"dec{int(tied_storage)}.pt"
        torch.save(text_decoder_to_fairseq1(want, tied_storage), f)
        got = convert_sonar_text_decoder_checkpoint(torch.load(f, weights_only=False))
        assert "final_proj.weight" not in got     # tied: the engine multiplies by the embedding table itself
        _same(got, want)
    # fairseq2 layout, as test_tied_weights.py saves it: {"model": state_dict} with both tied keys
    fs2 = dict(want)
    fs2["final_proj.weight"] = want["decoder_frontend.embed.weight"]
    f = tmp_path / "dec_fs2.pt"
    torch.save({"model": fs2}, f)
    _same(convert_sonar_text_decoder_checkpoint(torch.load(f, weights_only=False)), want)


def test_packed_cache(tmp_path, monkeypatch):
    from oracle import text_encoder as OE
    from sonar_amd import packed_cache as PC
    from sonar_amd.text_encoder import convert_sonar_text_encoder_checkpoint

    monkeypatch.setenv("SONAR_AMD_CACHE", str(tmp_path / "cache"))
    cfg = OE.OracleTextEncoderConfig(model_dim=64, num_layers=1, num_heads=1, ffn_inner_dim=128, vocab_size=50)
    want = OE.make_synthetic_params(cfg, seed=1)
    f = tmp_path / "enc.pt"
    torch.save(text_encoder_to_fairseq1(want), f)
    st = {}
    a = PC.load_converted(f, convert_sonar_text_encoder_checkpoint, "text_encoder", st)
    assert st["cache"] == "miss" and len(list((tmp_path / "cache").iterdir())) == 1
    calls = []
    b = PC.load_converted(f, lambda ck: calls.append(1) or {}, "text_encoder", st)
    assert st["cache"] == "hit" and not calls              # the second load neither unpickles nor converts
    assert set(a) == set(b) == set(want)
    for k, v in want.items():
        assert torch.equal(a[k], b[k])
        if v.dim() >= 2:
            assert b[k].dtype == torch.float16 and torch.equal(b[k], v.half())   # the engine's own rounding
        else:
            assert b[k].dtype == torch.float32 and torch.equal(b[k], v)
    # a changed file is a different cache entry
    import os
    import time
    time.sleep(0.01)
    torch.save(text_encoder_to_fairseq1(OE.make_synthetic_params(cfg, seed=2)), f)
    os.utime(f, None)
    PC.load_converted(f, convert_sonar_text_encoder_checkpoint, "text_encoder", st)
    assert st["cache"] == "miss"
    monkeypatch.setenv("SONAR_AMD_CACHE", "0")
    PC.load_converted(f, convert_sonar_text_encoder_checkpoint, "text_encoder", st)
    assert st["cache"] == "off"
    # depthwise taps and BatchNorm statistics stay fp32, integer buffers are dropped
    p = PC.pack_state_dict({"encoder.layers.0.conv.depthwise_conv.weight": torch.randn(8, 1, 7),
                            "encoder.layers.0.conv.batch_norm.num_batches_tracked": torch.tensor(3),
                            "encoder.layers.0.conv.pointwise_conv1.weight": torch.randn(16, 8, 1)})
    assert p["encoder.layers.0.conv.depthwise_conv.weight"].dtype == torch.float32
    assert p["encoder.layers.0.conv.pointwise_conv1.weight"].dtype == torch.float16
    assert len(p) == 2
    # the conformer's relative-position biases are 2-D but the engine keeps them fp32 (never a GEMM operand):
    # a file-loaded model must hold the same values as one built from the in-memory dict
    ub = torch.randn(4, 64) * 1e-3 + 1.0   # values that fp16 would visibly round
    q = PC.pack_state_dict({"encoder.layers.0.self_attn.sdpa.u_bias": ub, "encoder.layers.0.self_attn.sdpa.v_bias": ub + 1})
    assert q["encoder.layers.0.self_attn.sdpa.u_bias"].dtype == torch.float32
    assert torch.equal(q["encoder.layers.0.self_attn.sdpa.u_bias"], ub)
    assert torch.equal(q["encoder.layers.0.self_attn.sdpa.v_bias"], ub + 1)


def _load_ckpt_reference():
    import os

    return torch.load(os.path.join(os.path.dirname(__file__), "golden", "ckpt_reference.pt"), weights_only=False)


def test_converters_equal_the_reference_handlers():
    """The three converters against the outputs of the REFERENCE'S OWN converter functions -- sonar_text/handler.py:52-94,
    :122-172 and sonar_speech/handler.py:46-110, imported by path and executed in the build container by
    tests/golden/make_golden_ckpt.py (fixture: inputs + every output key with its tensor) -- on the same fairseq1-layout
    checkpoints.  Key sets and tensors must be identical; the documented differences of this package's flat format are
    spelled out below."""
    from sonar_amd.speech_encoder import convert_sonar_speech_checkpoint
    from sonar_amd.text_decoder import convert_sonar_text_decoder_checkpoint
    from sonar_amd.text_encoder import convert_sonar_text_encoder_checkpoint

    fx = _load_ckpt_reference()
    ins, ref = fx["inputs"], fx["reference"]

    # text encoder: the reference returns {"model": renamed state dict} and ALSO leaves the permuted table at the top level
    # (handler.py:92); in "model" the table is permuted too because the checkpoint's embed_tokens module shares its storage
    got = convert_sonar_text_encoder_checkpoint((ins["text_encoder"]))
    want = ref["text_encoder"]["model"]
    _same(got, want)
    assert torch.equal(got["encoder_frontend.embed.weight"], ref["text_encoder"]["top_level_embed"])
    src = ins["text_encoder"]["state_dict"]["embed_tokens.weight"]
    assert torch.equal(got["encoder_frontend.embed.weight"][[0, 1, 2, 3]], src[[1, 3, 0, 2]])   # (BOS,PAD,EOS,UNK) -> (PAD,UNK,BOS,EOS)
    assert torch.equal(got["encoder_frontend.embed.weight"][4:], src[4:])

    # text decoder: identical but for final_proj.weight, which this package drops (the engine multiplies by the embedding
    # table itself: TiedProjection, factory.py:306-307) -- in the reference's output it IS the permuted table
    got = convert_sonar_text_decoder_checkpoint((ins["text_decoder"]))
    want = dict(ref["text_decoder"]["model"])
    tied = want.pop("final_proj.weight")
    assert torch.equal(tied, want["decoder_frontend.embed.weight"])
    _same(got, want)

    # speech encoder: identical
    got = convert_sonar_speech_checkpoint((ins["speech_encoder"]))
    _same(got, ref["speech_encoder"]["model"])
