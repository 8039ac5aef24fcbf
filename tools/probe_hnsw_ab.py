"""A/B probe of the HNSW walk's distance evaluation (register loads vs LDS-DMA staging) inside one process (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 960
dev = torch.device('cuda', 0)
base = gist_like_gpu(torch, n, dim, 1806, dev); qs = gist_like_gpu(torch, 1000, dim, 1807, dev).cpu().numpy()
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n)
t = time.time(); ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=64, nthreads=16); print(f"build {time.time()-t:.1f} s", flush=True)
ref = None
for rnd in range(2):
    for v in (0, 2, 1):
        ix.set_param('hnsw_dma', v)
        ix.knn_with_ef(qs, 10, 128)
        t = time.perf_counter(); idx, d, c = ix.knn_with_ef(qs, 10, 128); dt = time.perf_counter() - t
        st = ix.hnsw_last_stats()
        same = True if ref is None else bool((ref[0] == idx).all() and (ref[1] == d).all() and ref[2] == st)
        if ref is None: ref = (idx.copy(), d.copy(), st)
        print(f"dma={v} rnd {rnd}: {dt*1e3:.2f} ms -> {1000/dt:.0f} QPS stats={st} same={same}", flush=True)
