"""Randomised parity run of the IVF probe-list scan (with and without its certified half-precision pre-pass) against the oracle
on the same centroids and clusters (longer than the suite's cases; not part of it).
usage: python tools/fuzz_ivf.py [seconds] [seed]"""
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
    dim = int(rng.choice([64, 100, 128, 192, 256, 320, 960]))
    n = int(rng.integers(3000, 40000 if dim < 960 else 12000))
    nq = int(rng.choice([1, 9, 70, 300]))
    k = int(rng.choice([1, 5, 10, 40, 90]))
    kc = int(rng.integers(2, 80))
    npb = int(rng.choice([1, 2, 4, 9, 30]))
    dist = str(rng.choice(["l2sqr", "cosine"]))
    kind = 0 if dist == "l2sqr" else 1
    style = int(rng.integers(0, 5))
    if style == 0:
        base = rng.standard_normal((n, dim)).astype(np.float32)
    elif style == 1:  # positive, quantised like gist
        base = np.round(np.abs(rng.normal(0.07, 0.045, (n, dim))), 4).astype(np.float32)
    elif style == 2:  # clusters of near-duplicates: many offers within the bound of the k-th distance
        c = rng.standard_normal((n // 40 + 1, dim)).astype(np.float32)
        base = (np.repeat(c, 40, axis=0)[:n] + 1e-3 * rng.standard_normal((n, dim))).astype(np.float32)
    elif style == 3:  # wildly different row norms
        base = (rng.standard_normal((n, dim)) * np.exp(rng.normal(0, 1.0, (n, 1)))).astype(np.float32)
    else:  # every row several times: exact ties at every cut
        u = rng.standard_normal((n // 5 + 1, dim)).astype(np.float32)
        base = np.concatenate([u] * 5)[:n]
    qs = (base[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * np.float32(0.05) * np.abs(base).mean()).astype(np.float32)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.ivf_build(kc, train_n=min(n, 1500), max_iter=4, seed=it)
    ex = ix.ivf_export()
    iv = O.IVF(base, ex["centroids"], kind, assign=ex["assign"])
    half = int(rng.choice([0, 1, 1])); q8 = int(rng.choice([0, 1, 1]))
    ix.set_param("ivf_half", half); ix.set_param("ivf_q8", q8)
    try:
        idx, d, cnt = ix.ivf_knn(qs, k, npb)
    finally:
        ix.set_param("ivf_half", 1); ix.set_param("ivf_q8", 1)
    ok = True
    for q in range(nq):
        oi, od = iv.knn(qs[q], k, npb)
        c = int(cnt[q])
        if c != len(oi) or idx[q, :c].tolist() != oi.tolist() or not np.array_equal(d[q, :c], od, equal_nan=True):
            ok = False
            break
    if not ok:
        bad += 1
        print(f"MISMATCH it={it} dim={dim} n={n} nq={nq} k={k} kc={kc} probes={npb} dist={dist} style={style} half={half} q8={q8} q={q}", flush=True)
    ix.close()
    if it % 10 == 0:
        print(f"{it} configurations, {bad} mismatches", flush=True)
print(f"done: {it} configurations, {bad} mismatches")
sys.exit(1 if bad else 0)
