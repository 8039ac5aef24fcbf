"""HNSW+PQ walk timing on a 100k-row graph (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_lowrank_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device('cuda', 0)
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
base = gist_lowrank_gpu(torch, n, 960, 1806, dev); qs = gist_lowrank_gpu(torch, nq, 960, 1807, dev).cpu().numpy()
ix = vdb.GpuIndex(960, 'l2sqr'); ix.add_device(base.data_ptr(), n)
ix.pq_build(n_bits=4, m=320, train_n=5000, max_iter=5, seed=42)
ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=1024, nthreads=16)
for it in range(3):
    t = time.perf_counter(); idx, d, c = ix.knn_pq(qs, 10, 128); dt = time.perf_counter() - t
print(f"hnsw_pq: {dt*1e3:.2f} ms -> {nq/dt:.0f} QPS stats={ix.hnsw_last_stats()} checksum={int(idx.sum())} {float(d.sum()):.6f}")
for it in range(3):
    t = time.perf_counter(); idx, d, c = ix.knn_with_ef(qs, 10, 128); dt = time.perf_counter() - t
print(f"hnsw   : {dt*1e3:.2f} ms -> {nq/dt:.0f} QPS")
