"""CPU: reader/writer for the reference's bincode files.  No reference-written file exists in
the repository (the README's dataset link is remote), so the layout is pinned by hand-assembled byte strings that
follow bincode 1.3.3's rules for the reference's struct definitions, plus round trips."""
import struct

import numpy as np
import pytest

from lab_1806_vec_db_amd import bincode_io as B
from oracle import oracle as O


def test_ground_truth_bytes():
    # GroundTruth{rows: Vec<GroundTruthRow{knn_indices: Vec<usize>}>} (candidate_pair.rs:111-149)
    raw = struct.pack("<Q", 2) + struct.pack("<Q3Q", 3, 7, 1, 4) + struct.pack("<Q1Q", 1, 9)
    rows = B.loads_ground_truth(raw)
    assert [r.tolist() for r in rows] == [[7, 1, 4], [9]]
    assert B.dumps_ground_truth(rows) == raw
    with pytest.raises(ValueError):
        B.loads_ground_truth(raw + b"\x00")
    with pytest.raises(ValueError):
        B.loads_ground_truth(raw[:-1])


def test_flat_table_bytes():
    # MetadataVecTable{metadata, inner: DynamicIndex::Flat(FlatIndex{dist, vec_set}), pq_table: None}
    raw = (struct.pack("<Q", 1) + struct.pack("<Q", 1) + struct.pack("<Q", 7) + b"content" + struct.pack("<Q", 1) + b"a"
           + struct.pack("<I", 0) + struct.pack("<I", 1) + struct.pack("<Q", 2) + struct.pack("<Q", 2)
           + struct.pack("<2f", 1.0, 0.5) + b"\x00")
    t = B.loads_table(raw)
    assert t["metadata"] == [{"content": "a"}] and t["inner"]["kind"] == "flat" and t["inner"]["dist"] == B.COSINE
    assert t["inner"]["rows"].tolist() == [[1.0, 0.5]] and t["pq_table"] is None
    assert B.dumps_table(t["metadata"], B.COSINE, t["inner"]["rows"]) == raw


def test_pq_and_hnsw_round_trip(gist_base):
    base = np.ascontiguousarray(gist_base[:200, :24])
    pq = O.PQ.train(base, m=8, n_bits=4, kind=O.COSINE, k_means_size=50, max_iter=3, seed=1)
    blob = B.dumps_pq_table(24, 4, 8, B.COSINE, pq.centroids, pq.codes, k_means_size=50)
    got = B.loads_pq_table(blob)
    assert got["n_bits"] == 4 and got["m"] == 8 and got["dim"] == 24 and got["k"] == 16 and got["encoded_dim"] == 4
    assert np.array_equal(got["centroids"], pq.centroids) and np.array_equal(got["codes"], pq.codes)
    assert np.array_equal(got["dist_cache"], pq.cent_cache)  # dot(c,c) in reference order (pq_table.rs:160-165)
    assert got["selected"] == [(3 * g, 3 * g + 3) for g in range(8)]
    assert B.dumps_pq_table(24, 4, 8, B.COSINE, got["centroids"], got["codes"], k_means_size=50) == blob

    h = O.HNSW.build(base, kind=O.L2SQR, M=6, ef_construction=30, seed=3)
    g = h.graph()
    for rows in (base, None):  # with and "without vec_set" (hnsw_index.rs:645-656)
        blob = B.dumps_hnsw_index(24, B.L2SQR, rows, g, 30)
        back = B.loads_hnsw_index(blob)
        assert back["config"]["m"] == 6 and back["config"]["max_m0"] == 12 and back["config"]["default_ef"] == 15
        assert back["rows"].shape[0] == (200 if rows is not None else 0)
        for key in ("level0", "len0", "vec_level", "upper", "upper_len"):
            assert np.array_equal(np.asarray(back["graph"][key]), np.asarray(g[key])), key
        assert (back["graph"]["enter_point"], back["graph"]["enter_level"]) == (g["enter_point"], g["enter_level"])
        # the graph read back drives the oracle to the same answers
        h2 = O.HNSW.from_graph(base, O.L2SQR, 6, 30, back["graph"])
        assert h2.knn(base[5], 4, 20)[0].tolist() == h.knn(base[5], 4, 20)[0].tolist()

    table = B.dumps_table([{"i": str(i), "k": "v"} for i in range(200)], B.L2SQR, base, hnsw_graph=g, ef_construction=30,
                          pq={"n_bits": 4, "m": 8, "centroids": pq.centroids, "codes": pq.codes})
    t = B.loads_table(table)
    assert t["inner"]["kind"] == "hnsw" and len(t["metadata"]) == 200 and t["metadata"][7] == {"i": "7", "k": "v"}
    assert np.array_equal(t["inner"]["rows"], base) and np.array_equal(t["pq_table"]["codes"], pq.codes)


def test_raw_vector_file(gist_base, tmp_path):
    p = tmp_path / "v.bin"
    gist_base[:10].tofile(p)
    assert np.array_equal(B.read_raw_vectors(p, 960), gist_base[:10])
    assert B.read_raw_vectors(p, 960, limit=3).shape == (3, 960)
