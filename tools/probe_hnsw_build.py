"""Build-time probe of the HNSW builder: `probe_hnsw_build.py rows batch [gpu]` (VDB_HNSW_PROF=1 prints the phases)."""
import os, sys, time, numpy as np, torch
os.environ.setdefault("VDB_HNSW_PROF", "1")
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_lowrank_gpu
n = int(sys.argv[1]); batch = int(sys.argv[2]); gpu = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
dev = torch.device('cuda', 0)
base = gist_lowrank_gpu(torch, n, 960, 1806, dev); qs = gist_lowrank_gpu(torch, 1000, 960, 1807, dev).cpu().numpy()
ix = vdb.GpuIndex(960, 'l2sqr'); ix.add_device(base.data_ptr(), n)
ix.set_param('hnsw_build_gpu', 0 if gpu else 1)
t = time.time(); ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=batch, nthreads=16); dt = time.time() - t
gt, _, _ = ix.flat_knn(qs, 10)
idx, d, c = ix.knn_with_ef(qs, 10, 128)
rec = np.mean([len(set(idx[q].tolist()) & set(gt[q].tolist())) / 10 for q in range(1000)])
t = time.perf_counter(); ix.knn_with_ef(qs, 10, 128); st = time.perf_counter() - t
print(f"n={n} batch={batch} gpu_assist={gpu}: build {dt:.1f} s; search ef=128: {1000/st:.0f} QPS recall@10={rec:.4f} stats={ix.hnsw_last_stats()}", flush=True)
