"""HNSW walk throughput against the query batch size: one wave per query, 4 waves per CU resident, so a 1024-query batch
is one round whose wall time is its longest walk (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_lowrank_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
dev = torch.device('cuda', 0)
base = gist_lowrank_gpu(torch, n, 960, 1806, dev)
ix = vdb.GpuIndex(960, 'l2sqr'); ix.add_device(base.data_ptr(), n)
t = time.time(); ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=1024, nthreads=16); print(f"build {time.time()-t:.1f} s", flush=True)
ref = {}
for half, nq in [(h, q) for h in (0, 2) for q in (256, 1024, 2048, 4096, 8192, 16384)]:
    ix.set_param('hnsw_half', half)
    qs = gist_lowrank_gpu(torch, nq, 960, 1807, dev).cpu().numpy()
    best = 1e9
    for it in range(4):
        t = time.perf_counter(); idx, d, c = ix.knn_with_ef(qs, 10, 128); best = min(best, time.perf_counter() - t)
    st = ix.hnsw_last_stats()
    same = True
    if half == 0: ref[nq] = (idx.copy(), d.copy(), st)
    else: same = bool((ref[nq][0] == idx).all() and (ref[nq][1] == d).all() and ref[nq][2] == st)
    print(f"half={half} same={same} nq={nq}: {best*1e3:.2f} ms -> {nq/best:.0f} QPS, {st[0]*3840/best/1e12:.2f} TB/s of row gathers, stats/query=({st[0]/nq:.0f}, {st[1]/nq:.1f})", flush=True)
