"""CPU: the oracle against HuggingFace on the GPU-sized twins (d = 256, 4 heads of 64; weights rebuilt by
tests/golden/twin_weights.py, outputs from tests/golden/make_golden_gpu_twin.py) -- the same fixture
the GPU tests hold the HIP engines to (tests/test_gpu_twin.py)."""
import os
import sys

import torch
from torch.testing import assert_close

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)


def test_oracle_matches_hf_on_the_gpu_sized_twins():
    import twin_weights as TW
    from oracle import text_decoder as OD
    from oracle import text_encoder as OE
    from sonar_amd.text_decoder import convert_sonar_text_decoder_checkpoint
    from sonar_amd.text_encoder import convert_sonar_text_encoder_checkpoint

    fx = torch.load(os.path.join(GOLDEN, "gpu_twin_outputs.pt"), weights_only=False)
    ecfg = OE.OracleTextEncoderConfig(model_dim=TW.D, num_layers=TW.L, num_heads=TW.H, ffn_inner_dim=TW.F,
                                      vocab_size=TW.V, max_seq_len=TW.MAXPOS - 2)
    eparams = convert_sonar_text_encoder_checkpoint(TW.fairseq_checkpoint("encoder"))
    _, emb = OE.text_encoder_forward(eparams, ecfg, fx["enc_ids"], fx["enc_lens"])
    assert_close(emb, fx["enc_pooled"], atol=2e-5, rtol=1e-4)

    dcfg = OD.OracleTextDecoderConfig(model_dim=TW.D, num_layers=TW.L, num_heads=TW.H, ffn_inner_dim=TW.F,
                                      vocab_size=TW.V, max_seq_len=TW.MAXPOS - 2)
    dparams = convert_sonar_text_decoder_checkpoint(TW.fairseq_checkpoint("decoder"))
    assert_close(OD.decoder_logits(dparams, dcfg, fx["dec_emb"], fx["dec_prev"]), fx["dec_logits"], atol=2e-4, rtol=1e-4)

    hyps = OD.beam_search(dparams, dcfg, fx["gen_emb"], fx["gen_prompt"].tolist(), beam_size=1, max_gen_len=(0, 10))
    for h, want, margin in zip(hyps, fx["gen_tokens"].tolist(), fx["gen_margin"]):
        got = h[0].seq.tolist()
        n = min(len(got), len(want)) - 1                  # the oracle's last token may be the forced EOS
        for j in range(n):
            if got[j] != want[j]:
                assert margin[j].item() < 1e-4            # fp32 vs fp32: only an exact near-tie may differ
                break
