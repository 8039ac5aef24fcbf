"""GPU: the bincode reader end to end.  A table file in the reference's layout (MetadataVecTable:
metadata, DynamicIndex::HNSW with its VecSet, Option<PQTable>; Appendix B of SURVEY.md) is written from one index,
read back, and its arrays are attached to a FRESH index through vdb_index_add / vdb_pq_attach / vdb_hnsw_attach: every
search of the re-attached index must equal the source index's and the oracle's.

The format itself stays UNPINNED: no reference-written file exists in /root/reference (data/ holds raw vector files
only) and the reference cannot be built here, so the bytes are checked against the layout rules and against this
package's own writer -- not against bytes the Rust `bincode::serialize_into` produced."""
import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_table_file_round_trip_into_fresh_index(dist, kind, tmp_path):
    import lab_1806_vec_db_amd as vdb
    from lab_1806_vec_db_amd import bincode_io as B
    from oracle import oracle as O

    n, dim = 3000, 64
    base = gist_like(n, dim=dim, seed=81)
    src = vdb.GpuIndex(dim, dist)
    src.batch_add(base)
    src.pq_build(n_bits=4, m=16, train_n=500, max_iter=4, seed=3)
    src.hnsw_build(M=8, ef_construction=40, seed=5, batch=8, nthreads=4)
    pq, g = src.pq_export(), src.hnsw_export()
    meta = [{"id": str(i)} for i in range(n)]
    path = tmp_path / "table.db"
    path.write_bytes(B.dumps_table(meta, kind, base, hnsw_graph=g, ef_construction=40,
                                   pq={"n_bits": 4, "m": 16, "centroids": pq["centroids"], "codes": pq["codes"]}))

    t = B.load(str(path), "table")
    assert t["inner"]["kind"] == "hnsw" and t["inner"]["dist"] == kind and t["metadata"][17] == {"id": "17"}
    rows, h, p = t["inner"]["rows"], t["inner"]["hnsw"], t["pq_table"]
    assert np.array_equal(rows, base)
    dst = vdb.GpuIndex(dim, dist)
    dst.batch_add(rows)
    dst.pq_attach(p["n_bits"], p["m"], p["centroids"], p["codes"])
    dst.hnsw_attach(h["config"]["m"], h["config"]["ef_construction"], h["graph"])
    assert np.array_equal(dst.pq_export()["codes"], pq["codes"])

    qs = gist_like(12, dim=dim, seed=82)
    opq = O.PQ.from_centroids(dim, 16, 4, kind, p["centroids"])
    opq.set_codes(p["codes"])
    oh = O.HNSW.from_graph(rows, kind, 8, 40, h["graph"])
    for fn, ofn in ((lambda ix: ix.flat_knn(qs, 10), lambda q: O.flat_knn(rows, q, 10, kind)),
                    (lambda ix: ix.knn_with_ef(qs, 10, 50), lambda q: oh.knn(q, 10, 50)),
                    (lambda ix: ix.knn_pq(qs, 10, 60), lambda q: oh.knn_pq(opq, q, 10, 60)),
                    (lambda ix: ix._search(ix._lib.vdb_flat_knn_pq, qs, 10, 60), lambda q: O.flat_knn_pq(rows, opq, q, 10, 60, kind))):
        a, b = fn(src), fn(dst)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        for q in range(len(qs)):
            oi, od = ofn(qs[q])
            assert b[0][q, :len(oi)].tolist() == oi.tolist() and np.array_equal(b[1][q, :len(od)], od)

    # a PQ table file and an HNSW file "without vec_set" (hnsw_index.rs:645-656) attach the same way
    pq_blob = B.dumps_pq_table(dim, 4, 16, kind, pq["centroids"], pq["codes"])
    hn_blob = B.dumps_hnsw_index(dim, kind, None, g, 40)
    p2, h2 = B.loads_pq_table(pq_blob), B.loads_hnsw_index(hn_blob)
    assert h2["rows"].shape[0] == 0
    d2 = vdb.GpuIndex(dim, dist)
    d2.batch_add(base)
    d2.pq_attach(p2["n_bits"], p2["m"], p2["centroids"], p2["codes"])
    d2.hnsw_attach(h2["config"]["m"], h2["config"]["ef_construction"], h2["graph"])
    a, b = src.knn_pq(qs, 10, 60), d2.knn_pq(qs, 10, 60)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # a truncated file is rejected by the reader, and mismatched arrays by the attach wrappers (no short reads in C)
    with pytest.raises(ValueError):
        B.loads_table(path.read_bytes()[:-5])
    d3 = vdb.GpuIndex(dim, dist)
    d3.batch_add(base[:-1])
    with pytest.raises(vdb.VdbError):
        d3.pq_attach(p2["n_bits"], p2["m"], p2["centroids"], p2["codes"])
    with pytest.raises(vdb.VdbError):
        d3.hnsw_attach(h2["config"]["m"], h2["config"]["ef_construction"], h2["graph"])
