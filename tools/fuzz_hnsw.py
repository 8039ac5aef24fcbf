"""Randomised parity run of the HNSW walk against the oracle on the same graph (longer than the suite's cases; not part of it):
row-staging variants, the certified half-precision pre-pass in its three modes, small and large query batches, result
lists of one lane up to the heap walk.  usage: python tools/fuzz_hnsw.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lab_1806_vec_db_amd as vdb
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1806)
t_end = time.time() + budget
it = bad = 0
while time.time() < t_end:
    it += 1
    dim = int(rng.choice([32, 64, 96, 128, 192, 256, 320, 960]))
    n = int(rng.integers(2000, 25000 if dim < 960 else 9000))
    nq = int(rng.choice([1, 7, 100, 800, 1500]))
    k = int(rng.choice([1, 5, 10, 40]))
    ef = int(rng.choice([k, 16, 64, 128, 300, 1100]))
    M = int(rng.choice([4, 8, 16, 24]))
    efc = int(rng.choice([20, 60, 100]))
    dist = str(rng.choice(["l2sqr", "cosine"]))
    kind = 0 if dist == "l2sqr" else 1
    style = int(rng.integers(0, 4))
    if style == 0:
        base = rng.standard_normal((n, dim)).astype(np.float32)
    elif style == 1:  # positive, quantised like gist
        base = np.round(np.abs(rng.normal(0.07, 0.045, (n, dim))), 4).astype(np.float32)
    elif style == 2:  # clusters of near-duplicates
        c = rng.standard_normal((n // 40 + 1, dim)).astype(np.float32)
        base = (np.repeat(c, 40, axis=0)[:n] + 1e-3 * rng.standard_normal((n, dim))).astype(np.float32)
    else:  # wildly different row norms
        base = (rng.standard_normal((n, dim)) * np.exp(rng.normal(0, 1.0, (n, 1)))).astype(np.float32)
    base[n - 3:] = base[:3]
    qs = (base[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * np.float32(0.05) * np.abs(base).mean()).astype(np.float32)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.hnsw_build(M=M, ef_construction=efc, seed=it, batch=int(rng.choice([1, 64, 512])), nthreads=16)
    oh = O.HNSW.from_graph(base, kind, M, efc, ix.hnsw_export())
    half = int(rng.choice([0, 1, 2])); dma = int(rng.choice([0, 1, 1, 2]))
    ix.set_param("hnsw_half", half); ix.set_param("hnsw_dma", dma)
    try:
        idx, d, cnt = ix.knn_with_ef(qs, k, ef)
        st = ix.hnsw_last_stats()
    finally:
        ix.set_param("hnsw_half", 1); ix.set_param("hnsw_dma", 1)
    oi, od, oc, nd, ne = oh.knn_batch(qs, k, ef, nthreads=16)
    ok = cnt.tolist() == oc.tolist() and st == (nd, ne)
    if ok:
        for q in range(nq):
            c = int(cnt[q])
            if idx[q, :c].tolist() != oi[q, :c].tolist() or not np.array_equal(d[q, :c], od[q, :c], equal_nan=True):
                ok = False
                break
    if not ok:
        bad += 1
        print(f"MISMATCH it={it} dim={dim} n={n} nq={nq} k={k} ef={ef} M={M} efc={efc} dist={dist} style={style} half={half} dma={dma} stats={st} oracle={(nd, ne)}", flush=True)
    if rng.random() < 0.35:  # HNSWIndex::knn_pq on the same graph (hnsw_index.rs:672-697): ADC walk + cached-form re-sort
        n_bits = int(rng.choice([4, 4, 8]))
        m = int(rng.integers(1, min(dim, 48) + 1))
        ix.pq_build(n_bits=n_bits, m=m, train_n=min(n, 1200 if n_bits == 4 else 3000), max_iter=2, seed=it)
        pq = ix.pq_export()
        opq = O.PQ.from_centroids(dim, m, n_bits, kind, pq["centroids"])
        opq.set_codes(pq["codes"])
        nqp = min(nq, 24)
        idx, d, cnt = ix.knn_pq(qs[:nqp], k, ef)
        okp = True
        for q in range(nqp):
            oi, od = oh.knn_pq(opq, qs[q], k, ef)
            c = int(cnt[q])
            if c != len(oi) or idx[q, :c].tolist() != oi.tolist() or not np.array_equal(d[q, :c], od, equal_nan=True):
                okp = False
                break
        if not okp:
            bad += 1
            print(f"MISMATCH (knn_pq) it={it} dim={dim} n={n} nq={nqp} k={k} ef={ef} M={M} dist={dist} style={style} n_bits={n_bits} m={m}", flush=True)
    ix.close()
    if it % 10 == 0:
        print(f"{it} configurations, {bad} mismatches", flush=True)
print(f"done: {it} configurations, {bad} mismatches")
sys.exit(1 if bad else 0)
