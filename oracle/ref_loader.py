"""TEST INFRASTRUCTURE ONLY -- loader for the *verbatim* reference VectorBase.

Executes /root/reference/src/typeagent/aitools/vectorbase.py unmodified under
stubbed sibling modules, so that its `VectorBase`, `TextEmbeddingIndexSettings`
and `ScoredInt` are the reference's own bytes.  This only works in the build
container (the GPU box has no /root/reference); it is used to

  * pin `oracle/vectorbase_oracle.py` (the numpy restatement that *does* travel)
    against the real reference, and
  * generate the golden vectors under tests/golden/ (tests/golden/make_golden.py).

Nothing in the product package imports this module.

Why stubs are needed: the package as a whole needs
Python >= 3.12 (PEP 695 `type X = ...` in aitools/embeddings.py:9-10) and
`model_adapters.py` imports pydantic_ai / stamina / openai / typechat, none of
which are installed.  vectorbase.py itself is valid 3.10 and needs only numpy
plus three names from `.embeddings` and one from `.model_adapters`
(vectorbase.py:9-14).
"""

from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("TAVB_REFERENCE_ROOT", "/root/reference")
_REF_FILE = os.path.join(REFERENCE_ROOT, "src", "typeagent", "aitools", "vectorbase.py")
_MODNAME = "typeagent.aitools.vectorbase"


def reference_available() -> bool:
    return os.path.isfile(_REF_FILE)


class NullEmbeddingModel:
    """Embedding model that refuses to embed (same idea as
    tools/benchmark_vectorbase.py:28-46); lets us build settings without
    triggering vectorbase.py:74's create_embedding_model()."""

    model_name = "oracle-null"

    def add_embedding(self, key, embedding):
        return None

    async def get_embedding_nocache(self, input):
        raise RuntimeError("oracle does not generate embeddings")

    async def get_embeddings_nocache(self, input):
        raise RuntimeError("oracle does not generate embeddings")

    async def get_embedding(self, key):
        raise RuntimeError("oracle does not generate embeddings")

    async def get_embeddings(self, keys):
        raise RuntimeError("oracle does not generate embeddings")


def _install_stubs() -> list[str]:
    """Register the stand-in sibling modules; returns the names this call added (the caller removes them again)."""
    added: list[str] = []

    def _mod(name: str, is_pkg: bool) -> types.ModuleType:
        added.append(name)
        m = types.ModuleType(name)
        if is_pkg:
            m.__path__ = []  # type: ignore[attr-defined]
        return m

    if "typeagent" not in sys.modules:
        sys.modules["typeagent"] = _mod("typeagent", True)
    if "typeagent.aitools" not in sys.modules:
        sys.modules["typeagent.aitools"] = _mod("typeagent.aitools", True)
    if "typeagent.aitools.embeddings" not in sys.modules:
        emb = _mod("typeagent.aitools.embeddings", False)
        emb.IEmbeddingModel = object
        emb.NormalizedEmbedding = np.ndarray
        emb.NormalizedEmbeddings = np.ndarray
        sys.modules["typeagent.aitools.embeddings"] = emb
    if "typeagent.aitools.model_adapters" not in sys.modules:
        ma = _mod("typeagent.aitools.model_adapters", False)

        def create_embedding_model(*a, **k):
            raise RuntimeError("stub: pass an explicit embedding_model")

        ma.create_embedding_model = create_embedding_model
        sys.modules["typeagent.aitools.model_adapters"] = ma
    return added


_cached = None


def load_reference_vectorbase():
    """Return the module object of the reference's vectorbase.py, executed verbatim."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise FileNotFoundError(
            f"{_REF_FILE} not found: the verbatim reference is only available in the build container"
        )
    added = _install_stubs()
    spec = importlib.util.spec_from_file_location(_MODNAME, _REF_FILE)
    assert spec is not None and spec.loader is not None
    mod = importlib.util.module_from_spec(spec)
    saved = sys.modules.get(_MODNAME)
    sys.modules[_MODNAME] = mod
    try:
        spec.loader.exec_module(mod)
    finally:
        # do not leave the reference bound under the public name: the product
        # package may want to install itself there (typeagent_py_amd.install).
        if saved is not None:
            sys.modules[_MODNAME] = saved
        else:
            del sys.modules[_MODNAME]
        # ... and no stub `typeagent` package either: a later `typeagent_py_amd.install()` in this process must see the
        # interpreter as it was (the loaded module keeps its own references to the stubs it imported)
        for name in added:
            sys.modules.pop(name, None)
    _cached = mod
    return mod


def make_reference_vectorbase(vectors: np.ndarray | None = None, **settings_kw):
    """Build a reference VectorBase (NullEmbeddingModel) optionally holding `vectors`."""
    mod = load_reference_vectorbase()
    settings = mod.TextEmbeddingIndexSettings(embedding_model=NullEmbeddingModel(), **settings_kw)
    vb = mod.VectorBase(settings)
    if vectors is not None:
        vb.add_embeddings(None, vectors)
    return vb
