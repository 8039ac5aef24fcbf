"""A/B probe of the ADC inner loop inside one process (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu
n, dim, nq, k, ef = 1_000_000, 960, 1000, 10, 100
dev = torch.device('cuda', 0)
base = gist_like_gpu(torch, n, dim, 1806, dev); qs = gist_like_gpu(torch, nq, dim, 1807, dev).cpu().numpy()
tr = vdb.GpuIndex(dim, 'l2sqr'); tr.add_device(base.data_ptr(), 10000); tr.pq_build(n_bits=4, m=320, train_n=0, max_iter=5, seed=42)
cent = tr.pq_export()['centroids']; del tr
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n); ix.pq_attach(4, 320, cent, None)
ix.prof_enable(True)
ref = None
for rnd in range(2):
    for v in (0, 1):
        ix.set_param('pq_adc_fast', v)
        ix.knn_pq(qs, k, ef)
        ix.prof_reset(); t = time.perf_counter(); idx, d, c = ix.knn_pq(qs, k, ef); dt = time.perf_counter() - t
        p = ix.prof_get('pq_adc')
        same = True if ref is None else bool((ref[0] == idx).all() and (ref[1] == d).all())
        if ref is None: ref = (idx.copy(), d.copy())
        print(f"fast={v} rnd {rnd}: {dt*1e3:.1f} ms -> {nq/dt:.0f} QPS; adc {p['ms']/p['launches']:.3f} ms x{p['launches']}; same={same}", flush=True)
