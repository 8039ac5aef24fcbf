"""Small HNSW calls (1, 32, 256, 1000 queries) at 1M rows: wall time per call with the half-precision pre-pass auto / off / forced (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_lowrank_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = torch.device('cuda', 0)
base = gist_lowrank_gpu(torch, n, 960, 1806, dev)
ix = vdb.GpuIndex(960, 'l2sqr'); ix.add_device(base.data_ptr(), n)
t = time.time(); ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=1024, nthreads=16); print(f"build {time.time()-t:.1f} s", flush=True)
tr = vdb.GpuIndex(960, 'l2sqr'); tr.add_device(base.data_ptr(), 10000)
tr.pq_build(n_bits=4, m=320, train_n=0, max_iter=20, tol=1e-6, seed=42)
ix.pq_attach(4, 320, tr.pq_export()["centroids"], None); tr.close()
qs = gist_lowrank_gpu(torch, 1024, 960, 1807, dev)
o_idx = torch.zeros((1024, 10), dtype=torch.int64, device=dev); o_dist = torch.zeros((1024, 10), dtype=torch.float32, device=dev); o_cnt = torch.zeros((1024,), dtype=torch.int64, device=dev)
extra = [a.split("=") for a in sys.argv[2:]]
for name, val in extra:
    ix.set_param(name, int(val))
import os
quick = os.environ.get("HNSW_PROBE_QUICK") == "1"  # exact walk, default pre-pass rule only
for use_pq in ((False,) if quick else (False, True)):
    for half in ((1,) if quick or use_pq else (1, 0, 2)):
        ix.set_param('hnsw_half', half)
        for nq in (1, 32, 256, 1000):
            ts = []
            for it in range(12):
                torch.cuda.synchronize(); t = time.perf_counter()
                ix.hnsw_knn_device(qs.data_ptr(), nq, 10, 128, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr(), use_pq=use_pq)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            st = ix.hnsw_last_stats()
            el = sorted(ts)[len(ts) // 2]
            print(f"{'ADC' if use_pq else 'exact'} half={half} nq={nq}: median {el*1e3:.3f} ms (min {min(ts)*1e3:.3f}); n_exp/query {st[1]/nq:.1f} -> {el*1e6/(st[1]/nq):.2f} us per expansion", flush=True)
