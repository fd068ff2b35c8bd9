"""TEST INFRASTRUCTURE ONLY -- run the reference's own VectorBase CONSUMERS, verbatim, over a VectorBase class of our
choice (build container only: needs /root/reference).

The reference package cannot be imported here (Python >= 3.12 syntax in `knowpro/interfaces_*.py`, provider
dependencies), but these consumer files are valid 3.10 and only need a handful of names from their siblings
(SURVEY.md section 8c):

    knowpro/fuzzyindex.py            EmbeddingIndex                 (reaches into `_vector_base._embedding_size`, :140-141)
    knowpro/textlocindex.py          TextToTextLocationIndex        (`lookup_text` threshold 0.85, :108)
    storage/memory/convthreads.py    ConversationThreads            (`add_key(cache=False)`, positional `fuzzy_lookup`, `add_embedding(list)`)
    storage/sqlite/messageindex.py   SqliteMessageTextIndex         (top-k THEN filter THEN per-message max, :182-257, 296-326)
    storage/sqlite/reltermsindex.py  SqliteRelatedTermsFuzzy        (sequential `lookup_terms`, :259-271)

and the two memory-provider consumers, which are valid 3.10 except for PEP 695 type-parameter lists on a few `def`s
(`async def build_message_index[TMessage: IMessage, ...](...)`, messageindex.py:22-25, 73, 96, 116; reltermsindex.py:98-101):

    storage/memory/messageindex.py   MessageTextIndex               (MESSAGE ordinals handed to the subset search as row ordinals, :173-183)
    storage/memory/reltermsindex.py  TermEmbeddingIndex             (sequential `lookup_terms`, :320-332), RelatedTermsIndex

Those two are executed through `pep695_to_310()`: the type-parameter lists are deleted and `from __future__ import annotations`
is prepended (so the annotations that name the deleted parameters are never evaluated); every other byte is the reference's.

`load_consumers(vectorbase_module)` executes those files unmodified inside a throw-away `typeagent` package whose
`typeagent.aitools.vectorbase` IS the given module (the verbatim reference's, or `typeagent_py_amd.vectorbase`) and whose
other siblings are the minimal stand-ins below.  It is the in-container proof of "drop-in": the same consumer bytes run over
both classes and must behave the same.  Nothing in the product package imports this module.
"""

from __future__ import annotations

import importlib.util
import os
import re
import sys
import types
import typing
from dataclasses import dataclass

import numpy as np

from .ref_loader import REFERENCE_ROOT

_SRC = os.path.join(REFERENCE_ROOT, "src", "typeagent")

CONSUMER_FILES = {
    "typeagent.knowpro.fuzzyindex": "knowpro/fuzzyindex.py",
    "typeagent.knowpro.textlocindex": "knowpro/textlocindex.py",
    "typeagent.storage.memory.convthreads": "storage/memory/convthreads.py",
    "typeagent.storage.sqlite.messageindex": "storage/sqlite/messageindex.py",
    "typeagent.storage.sqlite.reltermsindex": "storage/sqlite/reltermsindex.py",
}


# executed after pep695_to_310(); loaded in this order, memory/messageindex BEFORE sqlite/messageindex (which imports its base class from it)
TRANSFORMED_FILES = {
    "typeagent.storage.memory.messageindex": "storage/memory/messageindex.py",
    "typeagent.storage.memory.reltermsindex": "storage/memory/reltermsindex.py",
}

_TYPE_PARAMS = re.compile(r"((?:def|class)\s+\w+)\s*\[[^\]]*\](?=\s*[\(:])")


def pep695_to_310(source: str) -> str:
    """The whole source transform: drop PEP 695 type-parameter lists (`def f[T: X](...)` -> `def f(...)`) and make every
    annotation lazy.  Nothing else changes -- asserted by the caller (the number of edited lines is checked)."""
    return "from __future__ import annotations\n" + _TYPE_PARAMS.sub(r"\1", source)


def consumers_available() -> bool:
    return all(os.path.isfile(os.path.join(_SRC, rel)) for rel in list(CONSUMER_FILES.values()) + list(TRANSFORMED_FILES.values()))


# ---- stand-ins for `typeagent.knowpro.interfaces` (only what the five files touch) ---------------------------------
class _Subscriptable(dict):
    def __class_getitem__(cls, item):
        return cls


@dataclass
class TextLocation:
    message_ordinal: int = 0
    chunk_ordinal: int = 0

    def serialize(self) -> dict:
        return {"messageOrdinal": self.message_ordinal, "chunkOrdinal": self.chunk_ordinal}

    @staticmethod
    def deserialize(data: dict) -> "TextLocation":
        return TextLocation(data.get("messageOrdinal", 0), data.get("chunkOrdinal", 0))


@dataclass
class ScoredMessageOrdinal:
    message_ordinal: int
    score: float


@dataclass
class ScoredThreadOrdinal:
    thread_ordinal: int
    score: float


@dataclass
class Term:
    text: str
    weight: float | None = None

    def serialize(self) -> dict:  # knowpro/interfaces_core.py:399-402 (pydantic, by_alias, exclude_none)
        return {"text": self.text} if self.weight is None else {"text": self.text, "weight": self.weight}


@dataclass
class Thread:
    description: str
    ranges: list

    def serialize(self) -> dict:
        return {"description": self.description, "ranges": list(self.ranges)}

    @staticmethod
    def deserialize(data: dict) -> "Thread":
        return Thread(data["description"], list(data.get("ranges", [])))


def _interfaces_module() -> types.ModuleType:
    m = types.ModuleType("typeagent.knowpro.interfaces")
    for name, obj in dict(
        TextLocation=TextLocation, ScoredMessageOrdinal=ScoredMessageOrdinal, ScoredThreadOrdinal=ScoredThreadOrdinal, Term=Term, Thread=Thread,
        TextLocationData=dict, TextToTextLocationIndexData=_Subscriptable, MessageTextIndexData=_Subscriptable, TextEmbeddingIndexData=_Subscriptable,
        TermToRelatedTermsData=_Subscriptable, TermsToRelatedTermsIndexData=_Subscriptable, TermsToRelatedTermsDataItem=_Subscriptable, TermData=_Subscriptable,
        ConversationThreadData=_Subscriptable, ThreadDataItem=_Subscriptable, MessageOrdinal=int,
    ).items():
        setattr(m, name, obj)
    # (typing.Protocol bases: memory/reltermsindex.py:253 declares `class ITermEmbeddingIndex(ITermToRelatedTermsFuzzy, Protocol)`)
    for proto in ("IMessage", "IMessageCollection", "IConversationThreads", "ITermToRelatedTerms", "ITermToRelatedTermsFuzzy", "ITermToRelatedTermsIndex",
                  "IKnowledgeExtractor", "IStorageProvider", "IConversation", "IMessageTextIndex", "ITermToSemanticRefIndex"):
        setattr(m, proto, types.new_class(proto, (typing.Protocol,)))
    m.SearchTerm = SearchTerm
    return m


@dataclass
class SearchTerm:  # knowpro/interfaces_search.py (only named by memory/reltermsindex.py's resolve_related_terms, off the lookup path)
    term: Term
    related_terms: list | None = None


def _pkg(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = []  # type: ignore[attr-defined]
    return m


def load_consumers(vectorbase_module: types.ModuleType, keep: bool = False) -> types.SimpleNamespace:
    """Execute the five consumer files verbatim over `vectorbase_module` (must define VectorBase, ScoredInt,
    TextEmbeddingIndexSettings).  Returns a namespace with their modules and the stand-in interface types; sys.modules is
    left exactly as it was found -- unless `keep`, which leaves the throw-away `typeagent` package registered (to test
    `typeagent_py_amd.install()` against it) until `ns.cleanup()` is called."""
    if not consumers_available():
        raise FileNotFoundError(f"{_SRC}: the reference consumers are only available in the build container")
    saved = {k: v for k, v in sys.modules.items() if k == "typeagent" or k.startswith("typeagent.")}
    for k in saved:
        del sys.modules[k]
    try:
        mods: dict[str, types.ModuleType] = {}
        for name in ("typeagent", "typeagent.aitools", "typeagent.knowpro", "typeagent.storage", "typeagent.storage.memory", "typeagent.storage.sqlite"):
            mods[name] = _pkg(name)
        emb = types.ModuleType("typeagent.aitools.embeddings")
        emb.NormalizedEmbedding = np.ndarray
        emb.NormalizedEmbeddings = np.ndarray
        emb.IEmbeddingModel = object
        mods["typeagent.aitools.embeddings"] = emb
        mods["typeagent.aitools.vectorbase"] = vectorbase_module
        mods["typeagent.knowpro.interfaces"] = _interfaces_module()
        conv = types.ModuleType("typeagent.knowpro.convsettings")  # knowpro/convsettings.py:19-32 (the file itself imports stamina)

        class MessageTextIndexSettings:
            def __init__(self, embedding_index_settings):
                self.embedding_index_settings = embedding_index_settings

        class RelatedTermIndexSettings(MessageTextIndexSettings):
            pass

        conv.MessageTextIndexSettings = MessageTextIndexSettings
        conv.RelatedTermIndexSettings = RelatedTermIndexSettings
        mods["typeagent.knowpro.convsettings"] = conv
        # siblings memory/reltermsindex.py imports at module level but only uses in resolve_related_terms / dedupe_related_terms
        # (the query compiler's side of the file: out of scope, never called here)
        coll = types.ModuleType("typeagent.knowpro.collections")
        coll.TermSet = type("TermSet", (), {})
        mods["typeagent.knowpro.collections"] = coll
        common = types.ModuleType("typeagent.knowpro.common")
        common.is_search_term_wildcard = lambda search_term: search_term.term.text == "*"  # knowpro/common.py
        mods["typeagent.knowpro.common"] = common
        schema = types.ModuleType("typeagent.storage.sqlite.schema")  # storage/sqlite/schema.py:193-212 (PEP 695 elsewhere in the file)
        schema.serialize_embedding = lambda e: None if e is None else e.tobytes()
        schema.deserialize_embedding = lambda b: None if b is None else np.frombuffer(b, dtype=np.float32)
        mods["typeagent.storage.sqlite.schema"] = schema
        sys.modules.update(mods)
        ns = types.SimpleNamespace(interfaces=mods["typeagent.knowpro.interfaces"], convsettings=conv)
        def _exec(modname: str, rel: str, transform) -> types.ModuleType:
            path = os.path.join(_SRC, rel)
            mod = types.ModuleType(modname)
            mod.__file__ = path
            mod.__package__ = modname.rsplit(".", 1)[0]
            sys.modules[modname] = mod
            with open(path, encoding="utf-8") as f:
                source = f.read()
            if transform is not None:
                edited = transform(source)
                # the transform may only shorten `def name[...](` headers: every other line of the reference survives byte for byte
                kept_lines = set(edited.splitlines())
                lost = [ln for ln in source.splitlines() if ln not in kept_lines]
                assert 0 < len(lost) <= 12 and all("[" in ln or ln.strip().endswith(",") or ln.strip() in ("](", "]") for ln in lost), lost
                source = edited
            exec(compile(source, path, "exec"), mod.__dict__)  # the reference's bytes
            return mod

        def _attr(modname: str) -> str:
            leaf = modname.rsplit(".", 1)[1]
            if ".sqlite." in modname:
                return "sqlite_" + leaf
            if modname in TRANSFORMED_FILES:
                return "memory_" + leaf
            return leaf

        order = list(CONSUMER_FILES.items())
        first_sqlite = next(i for i, (name, _) in enumerate(order) if ".sqlite." in name)
        order[first_sqlite:first_sqlite] = list(TRANSFORMED_FILES.items())
        for modname, rel in order:
            setattr(ns, _attr(modname), _exec(modname, rel, pep695_to_310 if modname in TRANSFORMED_FILES else None))
        done = True
        return ns
    finally:
        def cleanup() -> None:
            for k in [k for k in sys.modules if k == "typeagent" or k.startswith("typeagent.")]:
                del sys.modules[k]
            sys.modules.update(saved)

        if keep and "done" in locals():
            ns.cleanup = cleanup
        else:
            cleanup()


MESSAGE_TEXT_INDEX_DDL = """
CREATE TABLE MessageTextIndex (
    msg_id INTEGER NOT NULL,
    chunk_ordinal INTEGER NOT NULL,
    embedding BLOB NOT NULL,
    index_position INTEGER
)
"""  # the columns of storage/sqlite/schema.py:71-81 the index touches
