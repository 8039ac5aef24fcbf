"""Independent numpy emulation of the reference's strict-f32 arithmetic.

TEST INFRASTRUCTURE ONLY.  A second, independent restatement (vectorised over
rows, sequential over the dimension) used to cross-check oracle/vdb_oracle.c and
to generate tests/golden/*.json (tests/golden/make_golden.py).  numpy float32
arithmetic rounds every operation to f32, which is exactly the semantics of
`a.iter().zip(b).map(..).sum()` in /root/reference/src/distance/mod.rs:72-77.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def dot_rows(rows: np.ndarray, q: np.ndarray) -> np.ndarray:
    """distance/mod.rs:72-74 for every row: strict left fold in f32."""
    rows = np.asarray(rows, dtype=F)
    q = np.asarray(q, dtype=F)
    acc = np.zeros(rows.shape[0], dtype=F)
    for j in range(rows.shape[1]):
        acc = acc + rows[:, j] * q[j]
    return acc


def l2_rows(rows: np.ndarray, q: np.ndarray) -> np.ndarray:
    """distance/mod.rs:75-77 for every row."""
    rows = np.asarray(rows, dtype=F)
    q = np.asarray(q, dtype=F)
    acc = np.zeros(rows.shape[0], dtype=F)
    for j in range(rows.shape[1]):
        df = rows[:, j] - q[j]
        acc = acc + df * df
    return acc


def selfdot_rows(rows: np.ndarray) -> np.ndarray:
    rows = np.asarray(rows, dtype=F)
    acc = np.zeros(rows.shape[0], dtype=F)
    for j in range(rows.shape[1]):
        acc = acc + rows[:, j] * rows[:, j]
    return acc


def cosine_rows(rows: np.ndarray, q: np.ndarray) -> np.ndarray:
    """distance/mod.rs:60-69: 1 - dot / max(|a||b|, 1e-10)."""
    q = np.asarray(q, dtype=F)
    nq = np.sqrt(selfdot_rows(q[None, :]))[0]
    nr = np.sqrt(selfdot_rows(rows))
    den = np.maximum(nq * nr, F(1e-10))
    return F(1.0) - dot_rows(rows, q) / den


def l2_cached_rows(rows: np.ndarray, q: np.ndarray) -> np.ndarray:
    """distance/mod.rs:54-57 with a = row, b = query (hnsw_index.rs:351-355)."""
    q = np.asarray(q, dtype=F)
    ca = selfdot_rows(rows)
    cb = selfdot_rows(q[None, :])[0]
    return (ca + cb) - F(2.0) * dot_rows(rows, q)


def topk_lex(dist: np.ndarray, k: int):
    """k smallest by (distance, index) -- candidate_pair.rs:36-41 + flat_index.rs:48-57."""
    order = np.lexsort((np.arange(dist.size), dist))[:k]
    return order.astype(np.uint64), dist[order]


def flat_knn(base: np.ndarray, q: np.ndarray, k: int, cosine: bool = False):
    d = cosine_rows(base, q) if cosine else l2_rows(base, q)
    return topk_lex(d, k)
