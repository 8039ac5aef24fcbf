"""Timing probe for BASELINE configs 3 (HNSW ef=128) and 4 (PQ-Flat ADC) on gist-like synthetic data (tooling)."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_h = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
dim, nq, k = 960, 1000, 10
dev = torch.device('cuda', 0)
base = gist_like_gpu(torch, n, dim, 1806, dev)
qs = gist_like_gpu(torch, nq, dim, 1807, dev).cpu().numpy()
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n)
t = time.time(); gt, _, _ = ix.flat_knn(qs, k); print(f'flat gt {time.time()-t:.3f}s')
t = time.time(); ix.pq_build(n_bits=4, m=320, train_n=10000, max_iter=20, tol=1e-6, seed=42); print(f'pq_build(train 10000) {time.time()-t:.2f}s')
ix.prof_enable(True)
for ef in (100, 200):
    for it in range(2):
        ix.prof_reset(); t = time.time(); idx, d, c = ix.knn_pq(qs, k, ef); dt = time.time() - t
    rec = np.mean([len(set(idx[q].tolist()) & set(gt[q].tolist())) / k for q in range(nq)])
    p = ix.prof_get('pq_adc')
    print(f'PQ-Flat n={n} ef={ef}: {dt*1e3:.1f} ms -> {nq/dt:.0f} QPS recall@10={rec:.4f}; adc kernel {p["ms"]/max(p["launches"],1):.3f} ms/launch x{p["launches"]} ({p["bytes"]/max(p["ms"],1e-9)/1e6:.0f} GB/s codes)')
# HNSW on a smaller corpus (host build is the slow part)
ix2 = vdb.GpuIndex(dim, 'l2sqr'); ix2.add_device(base.data_ptr(), n_h)
gt2, _, _ = ix2.flat_knn(qs, k)
t = time.time(); ix2.hnsw_build(M=16, ef_construction=200, seed=42, batch=64, nthreads=16); print(f'hnsw_build n={n_h} {time.time()-t:.1f}s')
for ef in (128, 200):
    for it in range(2):
        t = time.time(); idx, d, c = ix2.knn_with_ef(qs, k, ef); dt = time.time() - t
    rec = np.mean([len(set(idx[q].tolist()) & set(gt2[q].tolist())) / k for q in range(nq)])
    nd, ne = ix2.hnsw_last_stats()
    print(f'HNSW n={n_h} ef={ef}: {dt*1e3:.1f} ms -> {nq/dt:.0f} QPS recall@10={rec:.4f}; n_dist/q={nd/nq:.0f} n_exp/q={ne/nq:.0f} bytes/q={(nd*(dim*4+4)+ne*128)/nq/1e6:.2f} MB')
