from .text import EmbeddingToTextModelPipeline, TextToEmbeddingModelPipeline  # noqa: F401
