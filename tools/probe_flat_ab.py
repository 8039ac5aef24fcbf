"""A/B probe of parameter COMBINATIONS inside one process: `probe_flat_ab.py "a=1,b=2;a=2,b=2" [rows] [nq]` alternates the
settings, times whole Flat steps + the filter kernel and checks that the results never change (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu
combos = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in c.split(',')) for c in sys.argv[1].split(';')]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
dim, k = 960, 10
dev = torch.device('cuda', 0)
base = gist_like_gpu(torch, n, dim, 1806, dev); qs = gist_like_gpu(torch, nq, dim, 1807, dev)
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n); del base
oi = torch.zeros(nq, k, dtype=torch.int64, device=dev); od = torch.zeros(nq, k, device=dev); oc = torch.zeros(nq, dtype=torch.int64, device=dev)
ix.prof_enable(True)
ref = None
for rnd in range(3):
    for c in combos:
        for name, v in c.items(): ix.set_param(name, v)
        for _ in range(2): ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        ix.prof_reset(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        p = ix.prof_get('flat_half')
        if not p['launches']: p = ix.prof_get('flat_mfma')
        cur = (oi.cpu().numpy().copy(), od.cpu().numpy().copy())
        if ref is None: ref = cur
        same = bool((ref[0] == cur[0]).all() and (ref[1] == cur[1]).all())
        print(f"n={n} nq={nq} {c} rnd {rnd}: step {dt*1e3:.3f} ms, filter {p['ms']/p['launches']:.3f} ms, other {dt*1e3 - p['ms']/p['launches']:.3f} ms, same={same} fb={ix.flat_fallback_count()}", flush=True)
