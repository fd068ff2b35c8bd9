"""CPU suite for the host-side adapters around the hot path (SURVEY 8f "next" rows); the device
half (`lookup_texts_batched`) is in the GPU suite."""

import numpy as np
import pytest

from tests.fakes import NullModel
from typeagent_py_amd import ScoredInt, TextEmbeddingIndexSettings, VectorBase
from typeagent_py_amd.adapters import best_score_per_message, load_embeddings_bin


def test_best_score_per_message_matches_reference_aggregation():
    # rows 0,1 -> message 7; row 2 -> message 3; row 3 -> message 9
    hits = [ScoredInt(1, 0.9), ScoredInt(2, 0.8), ScoredInt(0, 0.95), ScoredInt(3, 0.8)]
    out = best_score_per_message(hits, [7, 7, 3, 9])
    assert [(h.item, h.score) for h in out] == [(7, 0.95), (3, 0.8), (9, 0.8)]  # stable among equal scores
    assert [(h.item, h.score) for h in best_score_per_message(hits, [7, 7, 3, 9], max_matches=1)] == [(7, 0.95)]
    only = best_score_per_message(hits, lambda r: [7, 7, 3, 9][r], accept=lambda m: m in {3, 9})
    assert [h.item for h in only] == [3, 9]


def test_load_embeddings_bin_streams_both_blocks(tmp_path):
    rng = np.random.default_rng(0)
    related = rng.standard_normal((37, 16)).astype("<f4")
    messages = rng.standard_normal((11, 16)).astype("<f4")
    path = tmp_path / "x_embeddings.bin"
    with open(path, "wb") as f:  # layout of knowpro/serialization.py:84-98
        f.write(related.tobytes())
        f.write(messages.tobytes())
    rv = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    mv = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    assert load_embeddings_bin(str(path), 16, 37, 11, rv, mv, chunk_rows=8) == (37, 11)
    np.testing.assert_array_equal(rv.serialize(), related)
    np.testing.assert_array_equal(mv.serialize(), messages)
    with pytest.raises(ValueError):
        load_embeddings_bin(str(path), 16, 37, 12, rv, mv)
