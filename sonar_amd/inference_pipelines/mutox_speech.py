"""MutoxSpeechClassifierPipeline with the reference's shape
(sonar/inference_pipelines/mutox_speech.py:25-93): audio -> fbank -> SONAR speech encoder ->
MuTox classifier, all three stages on the MI355X engine (GPU fbank, conformer, MLP head)."""
from __future__ import annotations

from pathlib import Path
from typing import Sequence, Union

import torch

from ..heads import MutoxClassifier, load_mutox_model
from ..speech_encoder import SonarSpeechEncoderModel
from .speech import SpeechToEmbeddingModelPipeline

CPU_DEVICE = torch.device("cpu")


class MutoxSpeechClassifierPipeline(torch.nn.Module):
    def __init__(self, mutox_classifier: Union[str, Path, MutoxClassifier],
                 encoder: Union[str, Path, SonarSpeechEncoderModel], device: torch.device = CPU_DEVICE) -> None:
        """mutox_classifier / encoder: checkpoint paths or ready model objects (the reference takes
        card names; there is no model hub here)."""
        super().__init__()
        self.speech = SpeechToEmbeddingModelPipeline(encoder, device=device)
        self.model = self.speech.model
        if isinstance(mutox_classifier, (str, Path)):
            mutox_classifier = load_mutox_model(str(mutox_classifier), device=self.speech.device)
        self.mutox_classifier = mutox_classifier

    @torch.inference_mode()
    def _run_classifier(self, data: dict) -> torch.Tensor:
        sentence_embeddings = data.get("sentence_embeddings")
        if sentence_embeddings is None:
            raise ValueError("Missing sentence embeddings in the data.")
        return self.mutox_classifier(sentence_embeddings)

    @torch.inference_mode()
    def predict(self, input: Sequence[Union[str, Path, torch.Tensor]], batch_size: int = 3,
                output_prob: bool = False, **kwargs) -> torch.Tensor:
        """Toxicity logits (or probabilities) [n, 1] for audio files / waveform tensors, input order kept."""
        emb = self.speech.predict(input, batch_size=batch_size, **kwargs)
        if output_prob:
            return self.mutox_classifier(emb, output_prob=True)
        return self._run_classifier({"sentence_embeddings": emb})
