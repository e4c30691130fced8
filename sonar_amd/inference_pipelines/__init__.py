from .text import (EmbeddingToTextModelPipeline, TextToEmbeddingModelPipeline,  # noqa: F401
                   TextToTextModelPipeline)
from .speech import (AudioToFbankDataPipelineBuilder, SpeechInferenceParams, SpeechToEmbeddingModelPipeline,  # noqa: F401,E402
                     SpeechToEmbeddingPipeline, SpeechToTextModelPipeline, SpeechToTextPipeline)
from .mutox_speech import MutoxSpeechClassifierPipeline  # noqa: F401,E402
