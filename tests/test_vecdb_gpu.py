"""GPU: the VecDB-shaped surface reproduces the reference's Python / database tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pyo3_example_restated():
    """examples/test_pyo3.py:13-32."""
    from lab_1806_vec_db_amd.vecdb import VecDB

    db = VecDB("./tmp/vec_db")
    for key in db.get_all_keys():
        db.delete_table(key)
    assert len(db.get_all_keys()) == 0
    db.create_table_if_not_exists("table_1", 4)
    db.add("table_1", [1.0, 0.0, 0.0, 0.0], {"content": "a"})
    db.add("table_1", [0.0, 1.0, 0.0, 0.0], {"content": "b"})
    db.build_hnsw_index("table_1")
    db.add("table_1", [0.0, 0.0, 1.0, 0.0], {"content": "c"})
    db.add("table_1", [0.0, 0.0, 1.0, 1.0], {"content": "d", "type": "oops"})
    assert db.has_hnsw_index("table_1"), "Add operation should not clear HNSW index"
    db.delete("table_1", {"type": "oops"})
    assert db.get_len("table_1") == 3
    assert not db.has_hnsw_index("table_1"), "HNSW index should be cleared when a vector is deleted"
    db.build_hnsw_index("table_1")
    db.build_pq_table("table_1")
    result = db.search("table_1", [1.0, 0.0, 0.0, 0.0], 3, None, 0.5)
    assert len(result) == 1
    assert result[0][0]["content"] == "a"


def test_database_mod_rs_restated():
    """database/mod.rs:551-607: cosine, dim 4, PQ with defaults, search k=ef=len, upper_bound 0.5 -> ["c"]."""
    from lab_1806_vec_db_amd.vecdb import VecDB

    db = VecDB()
    for key in ("table_1", "table_中文"):  # one key is non-ASCII in the reference test
        assert db.create_table_if_not_exists(key, 4, "cosine")
        assert not db.create_table_if_not_exists(key, 4, "cosine")
        db.batch_add(key, [[1, 0, 0, 0], [0, 1, 0, 0]], [{"content": "a"}, {"content": "b"}])
        db.add(key, [0, 0, 1, 0], {"content": "c"})
        db.build_pq_table(key)
        assert db.has_pq_table(key)
        n = db.get_len(key)
        res = db.search(key, [0.0, 0.0, 1.0, 0.0], n, n, 0.5)
        assert [m["content"] for m, _ in res] == ["c"]
        assert db.get_dist(key) == "cosine" and db.get_dim(key) == 4
    assert db.delete_table("table_1") and not db.delete_table("table_1")


def test_invalidation_rules_and_errors():
    from lab_1806_vec_db_amd.vecdb import VecDB

    db = VecDB()
    with pytest.raises(ValueError):
        db.create_table_if_not_exists("t", 4, "manhattan")  # pyo3/mod.rs:15-22
    db.create_table_if_not_exists("t", 8, "l2sqr")
    with pytest.raises(RuntimeError):
        db.build_pq_table("t")  # empty table (metadata_vec_table.rs:120-122)
    rng = np.random.default_rng(0)
    rows = rng.standard_normal((50, 8)).astype(np.float32)
    db.batch_add("t", rows, [{"i": str(i), "par": str(i % 2)} for i in range(50)])
    with pytest.raises(RuntimeError):
        db.add("t", [1.0, 2.0], {})  # dimension mismatch (database/mod.rs:427-429)
    with pytest.raises(RuntimeError):
        db.build_pq_table("t", n_bits=5)
    db.build_pq_table("t", n_bits=8)  # validated, but the table is 4-bit (metadata_vec_table.rs:140)
    assert db._t("t").index.pq_export()["n_bits"] == 4
    db.add("t", rows[0], {"i": "50", "par": "0"})
    assert not db.has_pq_table("t")  # add clears PQ (:65)
    db.build_hnsw_index("t", 50)
    db.build_pq_table("t", 0.5, None, 4)
    # delete by pattern: swap_remove in descending match order (:176-186)
    meta_before = [dict(m) for _, m in db.extract_data("t")]
    n_del = db.delete("t", {"par": "1"})
    assert n_del == 25 and db.get_len("t") == 26
    assert not db.has_hnsw_index("t") and not db.has_pq_table("t")
    exp = list(range(51))
    for i in reversed([i for i, m in enumerate(meta_before) if m["par"] == "1"]):
        exp[i] = exp[-1]
        exp.pop()
    got = [int(m["i"]) for _, m in db.extract_data("t")]
    assert got == exp
    # rows moved with their metadata
    data = db.extract_data("t")
    for vec, m in data:
        src = rows[0] if m["i"] == "50" else rows[int(m["i"])]
        assert np.array_equal(np.asarray(vec, np.float32), src)
    # Flat ignores ef (dynamic_index.rs:77); upper_bound filter
    r1 = db.search("t", rows[2], 5)
    r2 = db.search("t", rows[2], 5, ef=1)
    assert r1 == r2 and r1[0][1] == 0.0 and r1[0][0]["i"] == "2"
    assert db.search("t", rows[2], 5, upper_bound=-1.0) == []
