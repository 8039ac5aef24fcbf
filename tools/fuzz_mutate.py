"""Randomised WRITE / READ sequences on one index against the oracle: batch_add of random sizes, swap_remove (vec_set.rs:131-137),
searches in between (Flat with a random number of queries, so that the one-launch kernel, the many-queries path, the exact scan and
the MFMA pipeline with its mirrors -- refreshed by add, patched by swap_remove -- all see tables that grew and shrank across their
thresholds), an IVF or HNSW search on the current rows now and then (their lazily built images must follow the rows too).
usage: python tools/fuzz_mutate.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lab_1806_vec_db_amd as vdb
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1806)
t_end = time.time() + budget
runs = ops = bad = 0
while time.time() < t_end:
    runs += 1
    dim = int(rng.choice([16, 64, 96, 128, 192, 320, 960]))
    dist, kind = (("l2sqr", 0), ("cosine", 1))[int(rng.integers(0, 2))]
    scale = float(rng.choice([0.05, 1.0, 30.0]))
    mk = lambda m: (rng.standard_normal((m, dim)) * scale).astype(np.float32)  # noqa: E731
    ix = vdb.GpuIndex(dim, dist)
    host = np.zeros((0, dim), dtype=np.float32)
    n0 = int(rng.choice([50, 3000, 9000, 15000, 17000, 40000]))
    first = mk(n0)
    ix.batch_add(first)
    host = first.copy()
    log = [f"dim {dim} {dist} scale {scale} start {n0}"]
    for step in range(int(rng.integers(6, 16))):
        op = rng.random()
        if op < 0.3:  # add: one row, a handful, or a block that crosses a threshold
            m = int(rng.choice([1, 3, 200, 2500, 9000]))
            rows = mk(m)
            if rng.random() < 0.2:
                rows[0] = host[int(rng.integers(0, len(host)))]  # an exact duplicate: ties
            ix.batch_add(rows)
            host = np.concatenate([host, rows])
            log.append(f"add {m}")
        elif op < 0.55 and len(host) > 2:
            for _ in range(int(rng.choice([1, 1, 5, 300]))):
                if len(host) <= 2:
                    break
                i = int(rng.integers(0, len(host)))
                ix.swap_remove(i)
                host[i] = host[-1]
                host = host[:-1]
            log.append(f"swap_remove -> {len(host)}")
        else:
            nq = int(rng.choice([1, 2, 8, 31, 40, 130]))
            k = int(rng.choice([1, 10, 33]))
            qs = (host[rng.integers(0, len(host), nq)] + 0.05 * scale * rng.standard_normal((nq, dim))).astype(np.float32)
            which = rng.random()
            ops += 1
            if which < 0.75 or len(host) < 600:
                mode = int(rng.choice([0, 0, 0, 2]))
                ix.set_flat_mode(mode)
                idx, d, cnt = ix.flat_knn(qs, k)
                ix.set_flat_mode(0)
                oi, od, oc = O.flat_knn_batch(host, qs, k, kind, nthreads=16)
                ok = cnt.tolist() == oc.tolist() and all(idx[q, :int(cnt[q])].tolist() == oi[q][:int(cnt[q])].tolist() and
                                                         np.array_equal(d[q, :int(cnt[q])], od[q][:int(cnt[q])]) for q in range(nq))
                log.append(f"flat nq {nq} k {k} mode {mode} n {len(host)}: {'ok' if ok else 'MISMATCH'}")
            elif which < 0.9:
                kc = int(rng.integers(2, 30))
                ix.ivf_build(kc, train_n=min(len(host), 500), max_iter=3, seed=runs)
                ex = ix.ivf_export()
                iv = O.IVF(host, ex["centroids"], kind, assign=ex["assign"])
                npb = int(rng.choice([1, 4, 30]))
                idx, d, cnt = ix.ivf_knn(qs[:8], k, npb)
                ok = True
                for q in range(min(nq, 8)):
                    oi, od = iv.knn(qs[q], k, npb)
                    c = int(cnt[q])
                    ok = ok and c == len(oi) and idx[q, :c].tolist() == oi.tolist() and np.array_equal(d[q, :c], od)
                ix.ivf_clear()  # (swap_remove refuses to run under IVF clusters; add would leave them stale)
                log.append(f"ivf {kc} clusters {npb} probes n {len(host)}: {'ok' if ok else 'MISMATCH'}")
            else:
                sub = min(len(host), 3000)  # (the host builder on a prefix would not match the index: build on all rows when small)
                if len(host) <= 3000:
                    ix.hnsw_build(M=8, ef_construction=40, seed=runs, batch=1, nthreads=4)
                    oh = O.HNSW.from_graph(host, kind, 8, 40, ix.hnsw_export())
                    ix.set_param("hnsw_half", int(rng.choice([1, 2, 0])))
                    idx, d, cnt = ix.knn_with_ef(qs[:8], k, 50)
                    ix.set_param("hnsw_half", 1)
                    ok = True
                    for q in range(min(nq, 8)):
                        oi, od = oh.knn(qs[q], k, 50)
                        c = int(cnt[q])
                        ok = ok and c == len(oi) and idx[q, :c].tolist() == oi.tolist() and np.array_equal(d[q, :c], od)
                    ix.hnsw_clear()
                    log.append(f"hnsw n {len(host)}: {'ok' if ok else 'MISMATCH'}")
                else:
                    ok = True
            if not ok:
                bad += 1
                print("MISMATCH in run", runs, "|", " ; ".join(log), flush=True)
    print(f"run {runs}: {log[0]} -> {len(host)} rows after {len(log) - 1} operations", flush=True)
    ix.close()
print(f"done: {runs} runs, {ops} checked searches, {bad} mismatches")
sys.exit(1 if bad else 0)
