"""A/B of the HNSW walk's latency mode (vdb_set_param "hnsw_latency": 1 off, 2 + bits forced; bit 0 = pre-pass over all listed
neighbours, bit 1 = next candidate's link row fetched ahead) inside one process: calls of 1 / 256 / 1000 / 2048 queries on a
low-rank gist-like graph; results and work counters must not move.
usage: python tools/probe_hnsw_latency_ab.py [rows=300000] [pq=0]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_lowrank_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
use_pq = len(sys.argv) > 2 and sys.argv[2] == '1'
dim, k, ef = 960, 10, 128
dev = torch.device('cuda', 0)
base = gist_lowrank_gpu(torch, n, dim, 1806, dev); dq = gist_lowrank_gpu(torch, 2048, dim, 1807, dev)
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n)
t = time.time(); ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=1024, nthreads=16); print(f"build {time.time()-t:.1f} s", flush=True)
if use_pq:
    tr = vdb.GpuIndex(dim, 'l2sqr'); tr.add_device(base.data_ptr(), 10000); tr.pq_build(n_bits=4, m=320, train_n=0, max_iter=10, seed=42)
    ix.pq_attach(4, 320, tr.pq_export()['centroids'], None); del tr
o_i = torch.zeros((2048, k), dtype=torch.int64, device=dev); o_d = torch.zeros((2048, k), dtype=torch.float32, device=dev); o_c = torch.zeros((2048,), dtype=torch.int64, device=dev)
for nq in (1, 256, 1000, 2048):
    ref = None
    for rnd in range(2):
        for v in (1, 3, 4, 5):
            ix.set_param('hnsw_latency', v)
            call = lambda: ix.hnsw_knn_device(dq.data_ptr(), nq, k, ef, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr(), use_pq=use_pq)
            for _ in range(3): call()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): call()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
            cur = (o_i[:nq].cpu().numpy().copy(), o_d[:nq].cpu().numpy().copy(), ix.hnsw_last_stats())
            same = True if ref is None else bool((ref[0] == cur[0]).all() and (ref[1] == cur[1]).all() and ref[2] == cur[2])
            if ref is None: ref = cur
            print(f"nq {nq} hnsw_latency={v} (mode bits {0 if v == 1 else v - 2}) rnd {rnd}: {dt*1e3:.3f} ms  same={same}", flush=True)
ix.set_param('hnsw_latency', 0)
