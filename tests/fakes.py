"""Network-free embedding models for the API tests.

`FakeTextEmbedder` restates the reference's test embedder in plain Python:
rotate-and-hash floats `(hashish(rot_i(text)) % 1961) / 1961`
(/root/reference src/typeagent/aitools/model_adapters.py:375-404), then the
production L2 normalisation (model_adapters.py:181-183), wrapped in the caching
model (embeddings.py:73-114) exactly like `create_test_embedding_model`
(model_adapters.py:441-448) does.
"""

from __future__ import annotations

import numpy as np

from typeagent_py_amd.embeddings import CachingEmbeddingModel, IEmbedder, NormalizedEmbedding, NormalizedEmbeddings  # noqa: F401


def _hashish(s: str) -> int:
    h = 0
    for ch in s:
        h = (h * 31 + ord(ch)) & 0xFFFFFFFF
    return h


def _raw_fake_embedding(text: str, size: int) -> list[float]:
    if not text:
        raise ValueError("Empty input text")
    out = []
    for i in range(size):
        cut = i % len(text)
        out.append((_hashish(text[cut:] + text[:cut]) % 1961) / 1961)
    return out


class FakeTextEmbedder:
    model_name = "test"

    def __init__(self, embedding_size: int = 3):
        self.embedding_size = embedding_size

    async def get_embedding_nocache(self, input: str):
        return (await self.get_embeddings_nocache([input]))[0]

    async def get_embeddings_nocache(self, input: list[str]):
        if not input:
            raise ValueError("Cannot embed an empty list")
        e = np.array([_raw_fake_embedding(t, self.embedding_size) for t in input], dtype=np.float32)
        norms = np.linalg.norm(e, axis=1, keepdims=True).astype(np.float32)
        norms = np.where(norms > 0, norms, np.float32(1.0))
        return (e / norms).astype(np.float32)


def create_test_embedding_model(embedding_size: int = 3) -> CachingEmbeddingModel:
    return CachingEmbeddingModel(FakeTextEmbedder(embedding_size))


class NamedModel:
    """Settings tests only need `.model_name` (reference tests/test_vectorbase.py:22-45)."""

    def __init__(self, model_name: str):
        self.model_name = model_name

    def add_embedding(self, key, embedding):
        pass

    async def get_embedding_nocache(self, input):
        return np.array([1.0], dtype=np.float32)

    async def get_embeddings_nocache(self, input):
        return np.array([[1.0]], dtype=np.float32)

    async def get_embedding(self, key):
        return np.array([1.0], dtype=np.float32)

    async def get_embeddings(self, keys):
        return np.array([[1.0]], dtype=np.float32)


class NullModel:
    model_name = "benchmark-local"

    def add_embedding(self, key, embedding):
        return None

    async def get_embedding_nocache(self, input):
        raise RuntimeError("no embedding generation in this test")

    get_embeddings_nocache = get_embedding = get_embeddings = get_embedding_nocache
