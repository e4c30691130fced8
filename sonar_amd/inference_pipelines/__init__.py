from .text import EmbeddingToTextModelPipeline, TextToEmbeddingModelPipeline  # noqa: F401
from .speech import SpeechToEmbeddingModelPipeline  # noqa: F401,E402
