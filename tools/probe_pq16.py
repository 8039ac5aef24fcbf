"""A/B probe of the quantised ADC first pass (pq_adc16 0 = on, 1 = off) inside one process (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu, gist_lowrank_gpu
n, dim, nq, k = 1_000_000, 960, 1000, 10
gen = gist_lowrank_gpu if (len(sys.argv) > 1 and sys.argv[1] == 'lowrank') else gist_like_gpu
metric = sys.argv[2] if len(sys.argv) > 2 else 'l2sqr'
dev = torch.device('cuda', 0)
base = gen(torch, n, dim, 1806, dev); qt = gen(torch, nq, dim, 1807, dev); qs = qt.cpu().numpy()
tr = vdb.GpuIndex(dim, metric); tr.add_device(base.data_ptr(), 10000); tr.pq_build(n_bits=4, m=320, train_n=0, max_iter=20, seed=42)
cent = tr.pq_export()['centroids']; del tr
ix = vdb.GpuIndex(dim, metric); ix.add_device(base.data_ptr(), n); ix.pq_attach(4, 320, cent, None)
oi = torch.zeros((nq, k), dtype=torch.int64, device=dev); od = torch.zeros((nq, k), dtype=torch.float32, device=dev); oc = torch.zeros((nq,), dtype=torch.int64, device=dev)
ix.prof_enable(True)
for ef in (100, 200):
    ref = None
    for rnd in range(2):
        for v in (1, 0):
            ix.set_param('pq_adc16', v)
            ix.knn_pq_device(qt.data_ptr(), nq, k, ef, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
            torch.cuda.synchronize(); ix.prof_reset(); t = time.perf_counter()
            for _ in range(5):
                ix.knn_pq_device(qt.data_ptr(), nq, k, ef, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
            p = ix.prof_get('pq_adc')
            idx, d = oi.cpu().numpy(), od.cpu().numpy()
            same = True if ref is None else bool((ref[0] == idx).all() and (ref[1] == d).all())
            if ref is None: ref = (idx.copy(), d.copy())
            print(f"ef={ef} adc16={'off' if v else 'on'} rnd {rnd}: {dt*1e3:.2f} ms/step -> {nq/dt:.0f} QPS; adc {p['ms']/p['launches']:.3f} ms x{p['launches']/5:.0f} = {p['ms']/5:.2f} ms/step; same={same}", flush=True)
