"""Randomised parity run of the Flat pipeline against the oracle (longer than tests/test_fuzz_gpu.py; not part of the suite).
usage: python tools/fuzz_flat.py [seconds] [seed] [data style 0-3 or -1 = random] [smallest table, default 17000 rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lab_1806_vec_db_amd as vdb
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1806)
rng2 = np.random.default_rng(7 + (int(sys.argv[2]) if len(sys.argv) > 2 else 1806))
min_n = int(sys.argv[4]) if len(sys.argv) > 4 else 17000  # (below 16 384 rows calls of < 32 queries take the one-launch kernel)
t_end = time.time() + budget
it = bad = 0
while time.time() < t_end:
    it += 1
    dim = int(rng.choice([64, 96, 100, 128, 192, 256, 320, 384, 512, 768, 960, 1000, 1024, 1536]))
    if min_n >= 60000: dim = int(rng.choice([64, 128, 192, 320, 960]))  # (keeps the oracle's share of a configuration in seconds)
    big = min_n >= 60000  # (tables of >= 98 304 rows and calls of an even number of 128-query groups: the filter's cooperative sets)
    n = int(rng.integers(min_n, int(min_n * 1.6))) if big else int(rng.integers(min_n, 60000))
    nq = int(rng.choice([130, 256, 300, 512, 513, 700, 1024])) if big else int(rng.choice([1, 3, 17, 64, 65, 100, 128, 129, 200, 257]))
    k = int(rng.choice([1, 2, 5, 10, 16, 17, 33, 64, 70]))
    dist = str(rng.choice(["l2sqr", "cosine"]))
    kind = 0 if dist == "l2sqr" else 1
    style = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) >= 0 else int(rng.integers(0, 4))
    if style == 0:
        base = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    elif style == 1:  # positive, quantised like gist
        base = np.round(np.abs(rng.normal(0.07, 0.045, (n, dim))), 4).astype(np.float32)
        qs = np.round(np.abs(rng.normal(0.07, 0.045, (nq, dim))), 4).astype(np.float32)
    elif style == 2:  # clusters of near-duplicates (tiny margins -> redo tiers)
        c = rng.standard_normal((n // 50 + 1, dim)).astype(np.float32)
        base = (np.repeat(c, 50, axis=0)[:n] + 1e-4 * rng.standard_normal((n, dim))).astype(np.float32)
        qs = (c[rng.integers(0, len(c), nq)] + 1e-4 * rng.standard_normal((nq, dim))).astype(np.float32)
    else:  # wildly different row norms (hub rows under L2Sqr)
        base = (rng.standard_normal((n, dim)) * np.exp(rng.normal(0, 1.0, (n, 1)))).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    ix = vdb.GpuIndex(dim, dist)
    if rng.random() < 0.5:
        ix.batch_add(base)
    else:
        cut = int(rng.integers(1, n - 1))
        ix.batch_add(base[:cut]); ix.batch_add(base[cut:])
    ix.set_flat_mode(int(rng.choice([0, 2])))
    ix.set_param("flat_half", int(rng.choice([0, 0, 1, 2])))
    ix.set_param("flat_tail", int(rng.choice([0, 0, 1])))
    # (a generator of its own: the draws above stay those of the logged soaks, tools/replay_fuzz_flat.py)
    nw = int(rng2.choice([0, 0, 0, 40, 41, 8, 4, 2, 1]))
    ix.set_param("flat_tail_lb_nw", nw)
    sec, umin = int(rng2.choice([0, 0, 0, 1])), int(rng2.choice([0, 0, 1]))  # round 4: second 8-bit attempt off / unit-minima sample off
    ix.set_param("flat_i8_second", sec)
    ix.set_param("flat_i8_unit_min", umin)
    ref = int(rng2.choice([0, 0, 2, 2, 1]))  # hit keys refined from the fp16 image: auto / always / off (k_flat_refine_half)
    ix.set_param("flat_i8_refine", ref)
    idx, d, cnt = ix.flat_knn(qs, k)
    oi, od, oc = O.flat_knn_batch(base, qs, k, kind, nthreads=16)
    ok = cnt.tolist() == oc.tolist() and all(idx[q, :int(cnt[q])].tolist() == oi[q][:int(cnt[q])].tolist() and
                                             np.array_equal(d[q, :int(cnt[q])], od[q][:int(cnt[q])]) for q in range(nq))
    print(f"#{it} dim {dim} n {n} nq {nq} k {k} {dist} style {style}: {'ok' if ok else 'MISMATCH'} "
          f"i8 {ix.get_stat('flat_i8_queries')} second {ix.get_stat('flat_i8_second_queries')} passed on {ix.get_stat('flat_i8_redo')} tail {nw} refine {ref}:{ix.get_stat('flat_i8_refine_queries')} sets {ix.get_stat('flat_gemm8_coop_sets')} half {ix.get_stat('flat_half_queries')} redo {ix.get_stat('flat_half_redo')} fallback {ix.flat_fallback_count()}", flush=True)
    bad += 0 if ok else 1
    del ix
print(f"done: {it} configurations, {bad} mismatches")
sys.exit(1 if bad else 0)
