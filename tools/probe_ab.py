"""A/B probe inside one process (box-to-box variance is ~5 %): alternates a parameter and times whole steps + the filter kernel."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu
name = sys.argv[1]; vals = [int(x) for x in sys.argv[2].split(',')]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dim, nq, k = 960, 1000, 10
dev = torch.device('cuda', 0)
base = gist_like_gpu(torch, n, dim, 1806, dev); qs = gist_like_gpu(torch, nq, dim, 1807, dev)
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n); del base
oi = torch.zeros(nq, k, dtype=torch.int64, device=dev); od = torch.zeros(nq, k, device=dev); oc = torch.zeros(nq, dtype=torch.int64, device=dev)
ix.prof_enable(True)
ix.set_param('flat_half', 2)  # keep the fp16 first pass on even when a debug setting makes every query redo
for rnd in range(3):
    for v in vals:
        ix.set_param(name, v)
        for _ in range(2): ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        ix.prof_reset(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        p = ix.prof_get('flat_half')
        if not p['launches']: p = ix.prof_get('flat_mfma')
        print(f"{name}={v} rnd {rnd}: step {dt*1e3:.3f} ms, filter kernel {p['ms']/p['launches']:.3f} ms, other {dt*1e3 - p['ms']/p['launches']:.3f} ms, fb={ix.flat_fallback_count()}", flush=True)
