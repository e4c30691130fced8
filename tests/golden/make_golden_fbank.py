"""Generate tests/golden/fbank_seamless_twin.pt -- pins the oracle's Kaldi filterbank
(oracle/speech_encoder.py: kaldi_fbank, the restatement of fairseq2n's WaveformToFbankConverter as the
reference configures it, sonar/inference_pipelines/speech.py:283-290) against HuggingFace
`SeamlessM4TFeatureExtractor`, an independent numpy implementation of the same Kaldi front end for the
same w2v-BERT speech-encoder family (80 mel bins, 25 ms / 10 ms povey frames, pre-emphasis 0.97, DC
removal, waveform scale 2^15, log floor 2^-23, per-utterance mean / unbiased-variance normalisation).

Stored: the waveform (seeded uniform noise plus two tones, 1.3 s) and HF's features before and after
its normalisation.  Run in the build container:  python tests/golden/make_golden_fbank.py
"""
import os

import numpy as np
import torch
from transformers import SeamlessM4TFeatureExtractor

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fbank_seamless_twin.pt")


def main():
    g = torch.Generator().manual_seed(4)
    n = 16000 * 13 // 10 + 77
    t = torch.arange(n) / 16000.0
    wav = 0.3 * (torch.rand(n, generator=g) * 2 - 1) + 0.4 * torch.sin(2 * np.pi * 440 * t) + 0.2 * torch.sin(2 * np.pi * 3100 * t)
    fe = SeamlessM4TFeatureExtractor()
    raw = fe._extract_fbank_features(wav.numpy().astype(np.float32))            # [frames, 80] log-mel
    norm = fe(wav.numpy().astype(np.float32), sampling_rate=16000, return_tensors="np", padding=False,
              do_normalize_per_mel_bins=True)["input_features"][0]              # [frames // 2, 160] stacked
    torch.save({"waveform": wav, "fbank": torch.from_numpy(np.ascontiguousarray(raw)),
                "normalized_stacked": torch.from_numpy(np.ascontiguousarray(norm))}, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; frames", raw.shape[0])


if __name__ == "__main__":
    main()
