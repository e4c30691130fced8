from .text import (EmbeddingToTextModelPipeline, TextToEmbeddingModelPipeline,  # noqa: F401
                   TextToTextModelPipeline)
from .speech import SpeechToEmbeddingModelPipeline, SpeechToTextModelPipeline  # noqa: F401,E402
from .mutox_speech import MutoxSpeechClassifierPipeline  # noqa: F401,E402
