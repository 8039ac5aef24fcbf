"""Debug probe: fp16 first pass, shortlist length vs certification / redo on small and medium corpora."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lab_1806_vec_db_amd as vdb
from oracle import oracle as O

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 96
for n in (777, 5000, 40000):
    rng = np.random.default_rng(dim)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((9, dim)).astype(np.float32)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.set_flat_mode(2); ix.set_param("flat_gemm", 2)
    for half in (0, 1):
        ix.set_param("flat_half", half)
        for k, kmul in ((10, 4), (10, 7), (17, 1), (17, 4), (40, 4)):
            ix.set_param("flat_half_kmul", kmul)
            h0, r0, f0 = ix.get_stat("flat_half_queries"), ix.get_stat("flat_half_redo"), ix.flat_fallback_count()
            idx, dd, cnt = ix.flat_knn(qs, k)
            bad = []
            for q in range(9):
                oi, od = O.flat_knn(base, qs[q], k, 0)
                if idx[q, :len(oi)].tolist() != oi.tolist():
                    bad.append(q)
            print(f"n {n} half-mode {half} k {k} kmul {kmul}: half {ix.get_stat('flat_half_queries') - h0} redo {ix.get_stat('flat_half_redo') - r0} "
                  f"fallback {ix.flat_fallback_count() - f0} bad queries {bad}", flush=True)
