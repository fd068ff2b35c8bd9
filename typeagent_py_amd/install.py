"""Make typeagent use this engine: rebind the public names of
`typeagent.aitools.vectorbase` in that module and in every consumer that bound
them at import time with `from ...aitools.vectorbase import VectorBase`
(SURVEY.md section 8b): knowpro/fuzzyindex.py:9, knowpro/textlocindex.py:12,
storage/memory/reltermsindex.py:10-14, storage/memory/convthreads.py:4,
storage/sqlite/messageindex.py:12, storage/sqlite/reltermsindex.py:11.

Works in two situations:
  * typeagent is importable (Python >= 3.12 with its dependencies): the real
    modules are patched in place;
  * typeagent is NOT importable (this build container): `install()` registers this
    package's module as `sys.modules['typeagent.aitools.vectorbase']`, which is enough
    for anything that imports only that module -- e.g. the reference's own
    tools/benchmark_vectorbase.py runs unmodified on top of it (tests do this).
"""

from __future__ import annotations

import importlib
import sys
import types

from . import vectorbase as _vb

_PUBLIC = [
    "DEFAULT_MIN_SCORE",
    "MODEL_DEFAULT_MIN_SCORES",
    "ScoredInt",
    "TextEmbeddingIndexSettings",
    "VectorBase",
    "cosine_to_score",
    "get_default_min_score",
]

_CONSUMERS = [
    "typeagent.knowpro.fuzzyindex",
    "typeagent.knowpro.textlocindex",
    "typeagent.storage.memory.reltermsindex",
    "typeagent.storage.memory.convthreads",
    "typeagent.storage.memory.messageindex",
    "typeagent.storage.sqlite.messageindex",
    "typeagent.storage.sqlite.reltermsindex",
]

_saved: list[tuple[types.ModuleType, str, object]] = []
_registered: list[str] = []


def install(patch_consumers: bool = True) -> list[str]:
    """Returns the list of module names that were patched / registered."""
    touched: list[str] = []
    target = None
    try:
        target = importlib.import_module("typeagent.aitools.vectorbase")
    except Exception:
        target = None
    if target is None or target is _vb:
        # typeagent itself is not importable here: stand in for the one module.
        for name in ("typeagent", "typeagent.aitools"):
            if name not in sys.modules:
                pkg = types.ModuleType(name)
                pkg.__path__ = []  # type: ignore[attr-defined]
                sys.modules[name] = pkg
                _registered.append(name)
        if "typeagent.aitools.embeddings" not in sys.modules:
            from . import embeddings as _emb

            sys.modules["typeagent.aitools.embeddings"] = _emb
            _registered.append("typeagent.aitools.embeddings")
        if sys.modules.get("typeagent.aitools.vectorbase") is not _vb:
            sys.modules["typeagent.aitools.vectorbase"] = _vb
            _registered.append("typeagent.aitools.vectorbase")
        touched.append("typeagent.aitools.vectorbase")
        return touched
    for name in _PUBLIC:
        if hasattr(target, name):
            _saved.append((target, name, getattr(target, name)))
        setattr(target, name, getattr(_vb, name))
    touched.append(target.__name__)
    if patch_consumers:
        for modname in _CONSUMERS:
            mod = sys.modules.get(modname)
            if mod is None:
                try:
                    mod = importlib.import_module(modname)
                except Exception:
                    continue
            hit = False
            for name in _PUBLIC:
                if name in vars(mod):
                    _saved.append((mod, name, getattr(mod, name)))
                    setattr(mod, name, getattr(_vb, name))
                    hit = True
            if hit:
                touched.append(modname)
    return touched


def uninstall() -> None:
    while _saved:
        mod, name, value = _saved.pop()
        setattr(mod, name, value)
    while _registered:
        name = _registered.pop()
        sys.modules.pop(name, None)
