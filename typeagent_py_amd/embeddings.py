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
