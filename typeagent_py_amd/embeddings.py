"""Embedding type aliases, model protocols and the caching wrapper VectorBase
relies on -- host-side mirror of the reference's
`src/typeagent/aitools/embeddings.py` (aliases :9-10, IEmbedder :13-36,
IEmbeddingModel :39-70, CachingEmbeddingModel :73-114).  Nothing numeric lives
here; it exists so that the drop-in VectorBase can be exercised with the same
cache side effects the reference's tests pin (tests/test_vectorbase.py:82-145).
"""

from __future__ import annotations

from typing import Protocol, runtime_checkable

import numpy as np
from numpy.typing import NDArray

NormalizedEmbedding = NDArray[np.float32]  # one embedding, shape [D]
NormalizedEmbeddings = NDArray[np.float32]  # row-major [N, D]


@runtime_checkable
class IEmbedder(Protocol):
    """Provider interface: raw embedding computation only."""

    @property
    def model_name(self) -> str: ...

    async def get_embedding_nocache(self, input: str) -> NormalizedEmbedding: ...

    async def get_embeddings_nocache(self, input: list[str]) -> NormalizedEmbeddings: ...


@runtime_checkable
class IEmbeddingModel(IEmbedder, Protocol):
    """Consumer interface: the provider's three members plus a key -> embedding cache."""

    def add_embedding(self, key: str, embedding: NormalizedEmbedding) -> None: ...

    async def get_embedding(self, key: str) -> NormalizedEmbedding: ...

    async def get_embeddings(self, keys: list[str]) -> NormalizedEmbeddings: ...


class CachingEmbeddingModel:
    """In-memory key -> embedding cache in front of an IEmbedder."""

    def __init__(self, embedder: IEmbedder) -> None:
        self._embedder = embedder
        self._cache: dict[str, NormalizedEmbedding] = {}

    @property
    def model_name(self) -> str:
        return self._embedder.model_name

    def add_embedding(self, key: str, embedding: NormalizedEmbedding) -> None:
        self._cache[key] = embedding

    async def get_embedding_nocache(self, input: str) -> NormalizedEmbedding:
        return await self._embedder.get_embedding_nocache(input)

    async def get_embeddings_nocache(self, input: list[str]) -> NormalizedEmbeddings:
        return await self._embedder.get_embeddings_nocache(input)

    async def get_embedding(self, key: str) -> NormalizedEmbedding:
        hit = self._cache.get(key)
        if hit is not None:
            return hit
        fresh = await self._embedder.get_embedding_nocache(key)
        self._cache[key] = fresh
        return fresh

    async def get_embeddings(self, keys: list[str]) -> NormalizedEmbeddings:
        if not keys:
            raise ValueError("Cannot embed an empty list")
        todo = [k for k in keys if k not in self._cache]
        if todo:
            fresh = await self._embedder.get_embeddings_nocache(todo)
            for row, k in zip(fresh, todo):
                self._cache[k] = row
        return np.array([self._cache[k] for k in keys], dtype=np.float32)
