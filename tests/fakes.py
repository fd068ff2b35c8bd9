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

from typeagent_py_amd.embeddings import IEmbedder, NormalizedEmbedding, NormalizedEmbeddings


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


class CachingEmbeddingModel:
    """Test double for the provider-side cache that sits between VectorBase and an embedder (the reference keeps one in
    aitools/embeddings.py:73-114; typeagent's tests look into `_cache` to see what `add_key(..., cache=...)` did).
    One dict, filled by whatever had to be computed; everything else is a pass-through to the embedder."""

    def __init__(self, embedder: IEmbedder) -> None:
        self._embedder = embedder
        self._cache: dict[str, NormalizedEmbedding] = {}

    model_name = property(lambda self: self._embedder.model_name)

    def add_embedding(self, key: str, embedding: NormalizedEmbedding) -> None:
        self._cache[key] = embedding

    # uncached forms: straight to the embedder
    async def get_embedding_nocache(self, input: str) -> NormalizedEmbedding:
        return await self._embedder.get_embedding_nocache(input)

    async def get_embeddings_nocache(self, input: list[str]) -> NormalizedEmbeddings:
        return await self._embedder.get_embeddings_nocache(input)

    # cached forms: compute what is missing (in one embedder call), remember it, answer from the dict
    async def _fill(self, keys: list[str]) -> None:
        missing = list(dict.fromkeys(k for k in keys if k not in self._cache))
        if len(missing) == 1:
            self._cache[missing[0]] = await self._embedder.get_embedding_nocache(missing[0])
        elif missing:
            rows = await self._embedder.get_embeddings_nocache(missing)
            self._cache.update(zip(missing, rows))

    async def get_embedding(self, key: str) -> NormalizedEmbedding:
        await self._fill([key])
        return self._cache[key]

    async def get_embeddings(self, keys: list[str]) -> NormalizedEmbeddings:
        if not keys:
            raise ValueError("Cannot embed an empty list")
        await self._fill(keys)
        return np.stack([self._cache[k] for k in keys]).astype(np.float32, copy=False)


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
