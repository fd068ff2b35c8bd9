"""typeagent_py_amd -- MI355X (gfx950) engine behind typeagent-py's VectorBase.

Scope: exactly the reference's vector nearest-neighbour hot path
(`src/typeagent/aitools/vectorbase.py`): L2-normalise, query x corpus dot
products, score map, threshold and per-query top-k, single GPU or row-sharded
over the GPUs of a node.  Host code is Python over a ctypes C ABI
(`include/tavb.h`, `libtavb.so`); all arithmetic is hand-written HIP.

    from typeagent_py_amd import VectorBase, TextEmbeddingIndexSettings, ScoredInt
    import typeagent_py_amd; typeagent_py_amd.install()   # rebind typeagent's own names

(The directory is also reachable as `typeagent-py_amd/`, the name the build
contract uses; Python cannot import a hyphenated name, hence the alias.)
"""

from .embeddings import CachingEmbeddingModel, IEmbedder, IEmbeddingModel, NormalizedEmbedding, NormalizedEmbeddings
from .vectorbase import (
    DEFAULT_MIN_SCORE,
    MODEL_DEFAULT_MIN_SCORES,
    ScoredInt,
    TextEmbeddingIndexSettings,
    VectorBase,
    cosine_to_score,
    get_default_min_score,
)
from .install import install, uninstall

__all__ = [
    "CachingEmbeddingModel",
    "DEFAULT_MIN_SCORE",
    "IEmbedder",
    "IEmbeddingModel",
    "MODEL_DEFAULT_MIN_SCORES",
    "NormalizedEmbedding",
    "NormalizedEmbeddings",
    "ScoredInt",
    "TextEmbeddingIndexSettings",
    "VectorBase",
    "cosine_to_score",
    "get_default_min_score",
    "install",
    "uninstall",
]
__version__ = "0.1.0"
