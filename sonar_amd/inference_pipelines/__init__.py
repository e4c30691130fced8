from .text import TextToEmbeddingModelPipeline  # noqa: F401
