"""Embedding type aliases and model protocols VectorBase relies on -- host-side mirror of the reference's
`src/typeagent/aitools/embeddings.py` (aliases :9-10, IEmbedder :13-36,
IEmbeddingModel :39-70; the caching wrapper :73-114 is provider plumbing and lives with the test fakes).  Nothing numeric lives
here: VectorBase only calls these methods on whatever model the caller supplies.
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
