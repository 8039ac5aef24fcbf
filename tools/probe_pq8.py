"""PQ-Flat with 8-bit codes (n_bits = 8: 256 centroids per group, pq_table.rs:142-145) on 1M low-rank gist-like rows: where the scan
stands (tooling).  usage: python tools/probe_pq8.py [m=320] [nq=200]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_lowrank_gpu
m = int(sys.argv[1]) if len(sys.argv) > 1 else 320
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, dim, k, ef = 1_000_000, 960, 10, 100
dev = torch.device('cuda', 0)
base = gist_lowrank_gpu(torch, n, dim, 1806, dev); dq = gist_lowrank_gpu(torch, nq, dim, 1807, dev)
tr = vdb.GpuIndex(dim, 'l2sqr'); tr.add_device(base.data_ptr(), 20000)
t = time.time(); tr.pq_build(n_bits=8, m=m, train_n=0, max_iter=5, seed=42); print(f"train {time.time()-t:.1f} s", flush=True)
cent = tr.pq_export()['centroids']; del tr
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n)
t = time.time(); ix.pq_attach(8, m, cent, None); print(f"encode {time.time()-t:.1f} s", flush=True)
o_i = torch.zeros((nq, k), dtype=torch.int64, device=dev); o_d = torch.zeros((nq, k), dtype=torch.float32, device=dev); o_c = torch.zeros((nq,), dtype=torch.int64, device=dev)
ix.prof_enable(True)
for _ in range(2): ix.knn_pq_device(dq.data_ptr(), nq, k, ef, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
ix.prof_reset(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): ix.knn_pq_device(dq.data_ptr(), nq, k, ef, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
p = ix.prof_get('pq_adc')
print(f"n_bits 8, m {m}: {dt*1e3:.2f} ms per {nq} queries -> {nq/dt:.0f} QPS; pq_adc {p['ms']/max(p['launches'],1):.2f} ms x {p['launches']} launches, adc16 queries {ix.get_stat('pq_adc16_queries')}", flush=True)
