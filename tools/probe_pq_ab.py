"""A/B probe of a PQ-Flat tuning switch inside one process (tooling): knn_pq on 1M low-rank gist-like rows, 4-bit m = 320.
usage: python tools/probe_pq_ab.py [param=pq_sample16] [values=0,1] [ef=100]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_lowrank_gpu
param = sys.argv[1] if len(sys.argv) > 1 else 'pq_sample16'
vals = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '0,1').split(',')]
ef = int(sys.argv[3]) if len(sys.argv) > 3 else 100
n, dim, nq, k = 1_000_000, 960, 1000, 10
dev = torch.device('cuda', 0)
base = gist_lowrank_gpu(torch, n, dim, 1806, dev); dq = gist_lowrank_gpu(torch, nq, dim, 1807, dev)
tr = vdb.GpuIndex(dim, 'l2sqr'); tr.add_device(base.data_ptr(), 10000); tr.pq_build(n_bits=4, m=320, train_n=0, max_iter=20, seed=42)
cent = tr.pq_export()['centroids']; del tr
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n); ix.pq_attach(4, 320, cent, None)
o_i = torch.zeros((nq, k), dtype=torch.int64, device=dev); o_d = torch.zeros((nq, k), dtype=torch.float32, device=dev); o_c = torch.zeros((nq,), dtype=torch.int64, device=dev)
ix.prof_enable(True)
ref = None
for rnd in range(3):
    for v in vals:
        ix.set_param(param, v)
        for _ in range(2): ix.knn_pq_device(dq.data_ptr(), nq, k, ef, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
        ix.prof_reset(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): ix.knn_pq_device(dq.data_ptr(), nq, k, ef, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        p = ix.prof_get('pq_adc')
        cur = (o_i.cpu().numpy().copy(), o_d.cpu().numpy().copy())
        same = True if ref is None else bool((ref[0] == cur[0]).all() and (ref[1] == cur[1]).all())
        if ref is None: ref = cur
        print(f"{param}={v} rnd {rnd}: {dt*1e3:.3f} ms per step -> {nq/dt:.0f} QPS; adc16 {p['ms']/max(p['launches'],1):.3f} ms; same={same}", flush=True)
