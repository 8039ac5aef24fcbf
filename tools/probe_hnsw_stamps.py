"""Phase stamps of the HNSW walk (library built with -DHNSW_STAMP: the kernel sums wall-clock ticks per phase of the level-0 loop and the host
prints microseconds per expansion to stderr) for calls of 1, 32 and 1000 queries, exact and ADC walks (tooling; tools/hnsw_stamps.sh)."""
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
qs = gist_lowrank_gpu(torch, 1000, 960, 1807, dev)
o_idx = torch.zeros((1000, 10), dtype=torch.int64, device=dev); o_dist = torch.zeros((1000, 10), dtype=torch.float32, device=dev); o_cnt = torch.zeros((1000,), dtype=torch.int64, device=dev)
import os
exact_only = os.environ.get('HNSW_STAMPS_EXACT_ONLY') == '1'
for use_pq in ((False,) if exact_only else (False, True)):
    for nq in ((1, 1000) if exact_only else (1, 32, 1000)):
        for it in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            ix.hnsw_knn_device(qs.data_ptr(), nq, 10, 128, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr(), use_pq=use_pq)
            torch.cuda.synchronize(); el = time.perf_counter() - t
        st = ix.hnsw_last_stats()
        print(f"== {'ADC' if use_pq else 'exact'} walk, one call of {nq}: {el*1e3:.3f} ms; per query n_dist {st[0]/nq:.0f}, n_expanded {st[1]/nq:.1f} -> {el*1e6/(st[1]/nq):.2f} us per expansion", flush=True)
        sys.stderr.write(f"^^ {'ADC' if use_pq else 'exact'} nq={nq}\n"); sys.stderr.flush()
