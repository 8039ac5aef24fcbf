"""ef sweep of HNSW and HNSW+PQ on one graph (the grid of the reference's published table, data/t_bench.toml: ef = 120..360
for HNSW, 180..600 for HNSW+PQ): QPS and recall@10 per ef, 1000 queries per call (tooling; `sweep_hnsw_ef.py [rows]`)."""
import json, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_lowrank_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim, nq, k = 960, 1000, 10
dev = torch.device('cuda', 0)
base = gist_lowrank_gpu(torch, n, dim, 1806, dev); qt = gist_lowrank_gpu(torch, nq, dim, 1807, dev)
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n)
tr = vdb.GpuIndex(dim, 'l2sqr'); tr.add_device(base.data_ptr(), 10000); tr.pq_build(n_bits=4, m=dim // 3, train_n=0, max_iter=20, seed=42)
cent = tr.pq_export()['centroids']; tr.close()
t = time.time(); ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=1024, nthreads=16); build_s = time.time() - t
ix.pq_attach(4, dim // 3, cent, None)
oi = torch.zeros((nq, k), dtype=torch.int64, device=dev); od = torch.zeros((nq, k), dtype=torch.float32, device=dev); oc = torch.zeros((nq,), dtype=torch.int64, device=dev)
ix.flat_knn_device(qt.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr()); gt = oi.cpu().numpy().copy()
out = {"rows": n, "dim": dim, "queries_per_call": nq, "k": k, "data": "synthetic (low-rank gist-like)", "M": 16, "ef_construction": 200,
       "build_s": round(build_s, 1), "hnsw": [], "hnsw_pq": []}
for name, use_pq, efs in (("hnsw", False, (64, 120, 128, 160, 200, 240, 280, 320, 360)), ("hnsw_pq", True, (128, 180, 240, 300, 360, 480, 600))):
    for ef in efs:
        f = lambda: ix.hnsw_knn_device(qt.data_ptr(), nq, k, ef, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), use_pq=use_pq)
        f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        got = oi.cpu().numpy()
        rec = float(np.mean([len(set(got[q].tolist()) & set(gt[q].tolist())) / k for q in range(nq)]))
        nd, ne = ix.hnsw_last_stats()
        out[name].append({"ef": ef, "ms_per_query": round(dt / nq * 1e3, 5), "qps": round(nq / dt, 1), "recall_at_10": round(rec, 4),
                          "n_dist_per_query": round(nd / nq, 1), "n_expanded_per_query": round(ne / nq, 1)})
        print(name, out[name][-1], flush=True)
print(json.dumps(out))
