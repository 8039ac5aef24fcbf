"""Host HNSW build timing on gist-like rows (tooling): VDB_HNSW_PROF=1 prints the phase split."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device('cuda', 0)
base = gist_like_gpu(torch, n, 960, 1806, dev)
ix = vdb.GpuIndex(960, 'l2sqr'); ix.add_device(base.data_ptr(), n)
t = time.time(); ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=64, nthreads=16); print(f"build n={n}: {time.time()-t:.1f} s", flush=True)
