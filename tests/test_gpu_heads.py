"""GPU parity of the BLASER / MuTox heads (through the C ABI) against (a) outputs of the reference's
own modules (tests/golden/heads_reference.pt) and (b) the CPU oracle at the full widths.
Tolerance: the engine feeds fp16 features / hidden activations to fp32-accumulating MFMA GEMMs, the
reference computes in fp32 -> |delta score| <= 4e-3 + 4e-3 * |score| (scores are O(1))."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "heads_reference.pt")


def _close(got, want):
    err = (got.float().cpu() - want.float()).abs()
    tol = 4e-3 + 4e-3 * want.float().abs()
    assert bool((err <= tol).all()), (err.max().item(), want.abs().max().item())


def test_blaser_matches_reference_fixture():
    from sonar_amd.heads import BlaserConfig, BlaserModel

    gold = torch.load(GOLD)
    for g in gold["blaser"]:
        c = g["config"]
        cfg = BlaserConfig(input_form=c["input_form"], norm_emb=c["norm_emb"], embedding_dim=c["embedding_dim"],
                           hidden_dims=c["hidden_dims"], activation=c["activation"], output_act=c["output_act"])
        m = BlaserModel(cfg, {"model": g["state_dict"]}, device="cuda:0")
        y = m(src=g["src"], mt=g["mt"], ref=g["ref"])
        assert y.shape == (9, 1) and y.dtype == torch.float32
        _close(y, g["out"])
        _close(m(src=g["src"].half().cuda(), mt=g["mt"].half().cuda(), ref=g["ref"].half().cuda()), g["out"])
        if c["input_form"] == "QE":
            assert torch.equal(m(src=g["src"], mt=g["mt"]), y)     # the reference does not matter
        else:
            with pytest.raises(ValueError, match="reference embedding must be provided"):
                m(src=g["src"], mt=g["mt"])


def test_mutox_matches_reference_fixture():
    from sonar_amd.heads import MutoxClassifier, MutoxConfig

    gold = torch.load(GOLD)
    for g in gold["mutox"]:
        m = MutoxClassifier(MutoxConfig(g["input_size"]), g["state_dict"], device="cuda:0")
        _close(m(g["x"]), g["out"])
        _close(m(g["x"], output_prob=True), g["prob"])


@pytest.mark.parametrize("arch,n", [("basic_ref", 300), ("basic_qe", 129)])
def test_blaser_full_width_vs_oracle(arch, n):
    from oracle import heads as O
    from sonar_amd.heads import BlaserModel, get_blaser_config

    cfg = get_blaser_config(arch)
    g = torch.Generator().manual_seed(n)
    d = cfg.embedding_dim
    width = d * (6 if cfg.input_form == "COMET" else 4)
    dims = [width] + cfg.hidden_dims + [1]
    sd = {}
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        sd[f"mlp.{1 + 3 * i}.weight"] = (torch.randn(b, a, generator=g) * (2.0 / a ** 0.5)).half().float()
        sd[f"mlp.{1 + 3 * i}.bias"] = torch.randn(b, generator=g) * 0.2
    src, mt, ref = (torch.randn(n, d, generator=g) * s for s in (0.2, 0.3, 0.25))
    mt = 0.7 * src + 0.3 * mt   # correlated, like a translation
    m = BlaserModel(cfg, sd, device="cuda:0")
    y = m(src=src, mt=mt, ref=ref)
    want = O.blaser_forward(sd, src, mt, ref, input_form=cfg.input_form, norm_emb=True, activation="TANH")
    assert y.shape == (n, 1)
    _close(y, want)


def test_mutox_full_width_vs_oracle():
    from oracle import heads as O
    from sonar_amd.heads import MutoxClassifier, MutoxConfig

    g = torch.Generator().manual_seed(5)
    dims = [1024, 512, 128, 1]
    sd = {"unrelated.weight": torch.zeros(2, 2)}
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        sd[f"model_all.{i}.1.weight"] = (torch.randn(b, a, generator=g) * (1.5 / a ** 0.5)).half().float()
        sd[f"model_all.{i}.1.bias"] = torch.randn(b, generator=g) * 0.1
    x = torch.randn(513, 1024, generator=g) * 0.05
    m = MutoxClassifier(MutoxConfig(1024), sd, device="cuda:0")
    _close(m(x), O.mutox_forward(sd, x))
    _close(m(x.cuda().half(), output_prob=True), O.mutox_forward(sd, x.half().float(), output_prob=True))
    with pytest.raises(ValueError):
        MutoxClassifier(MutoxConfig(512), sd, device="cuda:0")


def test_mutox_speech_pipeline_end_to_end():
    """audio -> GPU fbank -> speech encoder -> MuTox head, against oracle encoder + oracle head
    (reference pipeline: sonar/inference_pipelines/mutox_speech.py:25-93)."""
    from oracle import heads as OH
    from oracle import speech_encoder as OS
    from oracle.speech_encoder import OracleSpeechEncoderConfig
    from sonar_amd.heads import MutoxClassifier, MutoxConfig
    from sonar_amd.inference_pipelines import MutoxSpeechClassifierPipeline
    from sonar_amd.speech_encoder import SonarSpeechEncoderConfig, SonarSpeechEncoderModel

    ocfg = OracleSpeechEncoderConfig(model_dim=256, num_layers=1, num_heads=4, ffn_inner_dim=512, conv_kernel=7,
                                     pooler_layers=1, pooler_heads=4, pooler_ffn_dim=384, pooler_vocab=64)
    cfg = SonarSpeechEncoderConfig(model_dim=256, num_encoder_layers=1, num_encoder_attn_heads=4, ffn_inner_dim=512,
                                   depthwise_conv_kernel_size=7, num_decoder_layers=1, num_decoder_attn_heads=4,
                                   decoder_ffn_inner_dim=384, max_frames=512)
    params = OS.make_synthetic_params(ocfg, seed=5, std=0.06)
    enc = SonarSpeechEncoderModel(cfg, params, device="cuda:0", dtype=torch.float32)
    gold = torch.load(GOLD)["mutox"][0]   # input_size 256 = the toy encoder's model_dim
    clf = MutoxClassifier(MutoxConfig(256), gold["state_dict"], device="cuda:0")
    pipe = MutoxSpeechClassifierPipeline(clf, enc, device=torch.device("cuda:0"))
    g = torch.Generator().manual_seed(11)
    wavs = [torch.rand(1, 20000, generator=g) * 2 - 1, torch.rand(1, 31000, generator=g) * 2 - 1]
    out = pipe.predict(wavs, batch_size=2)
    prob = pipe.predict(wavs, batch_size=1, output_prob=True)
    assert out.shape == (2, 1)
    for i, w in enumerate(wavs):
        f = OS.kaldi_fbank(w[0])
        t = f.shape[0] + f.shape[0] % 2
        fb = torch.zeros(1, t, 80)
        fb[0, : f.shape[0]] = f
        _, emb = OS.speech_encoder_forward(params, ocfg, fb, torch.tensor([f.shape[0]]))
        _close(out[i:i + 1], OH.mutox_forward(gold["state_dict"], emb))
        _close(prob[i:i + 1], OH.mutox_forward(gold["state_dict"], emb, output_prob=True))
    with pytest.raises(ValueError, match="Missing sentence embeddings"):
        pipe._run_classifier({})
