"""Embedding type aliases and model protocols VectorBase relies on -- host-side mirror of the reference's
`src/typeagent/aitools/embeddings.py` (aliases :9-10, IEmbedder :13-36, IEmbeddingModel :39-70, CachingEmbeddingModel :73-114:
`install()` registers this module as `typeagent.aitools.embeddings` when typeagent is absent, and the reference's own
`model_adapters` imports the caching wrapper from there).  Nothing numeric lives here: VectorBase only calls these methods on
whatever model the caller supplies.
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
    """The provider-side cache that turns an `IEmbedder` into an `IEmbeddingModel` (the reference's aitools/embeddings.py:73-114;
    its `model_adapters` imports the class from this module, and typeagent's tests look into `_cache` to see what
    `add_key(..., cache=...)` did).  One dict, filled by whatever had to be computed; everything else is a pass-through."""

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

    # cached forms.  The embedder sees exactly the calls the reference's class makes (aitools/embeddings.py:101-114): one
    # get_embedding_nocache(key) for a single miss, ONE get_embeddings_nocache(missing keys, in order, duplicates included) for a batch --
    # embedders that count, log or bill their calls, and batch-only ones, cannot tell the two classes apart.
    async def get_embedding(self, key: str) -> NormalizedEmbedding:
        hit = self._cache.get(key)
        if hit is not None:
            return hit
        row = await self._embedder.get_embedding_nocache(key)
        self._cache[key] = row
        return row

    async def get_embeddings(self, keys: list[str]) -> NormalizedEmbeddings:
        if not keys:
            raise ValueError("Cannot embed an empty list")
        absent = [k for k in keys if k not in self._cache]
        if absent:
            rows = await self._embedder.get_embeddings_nocache(absent)
            for k, row in zip(absent, rows):
                self._cache[k] = row
        return np.array([self._cache[k] for k in keys], dtype=np.float32)
