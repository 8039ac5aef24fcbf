"""Probe of the 8-bit first pass (k_gemm8.hip / k_i8.hip): (1) the lower-bound property of its keys against float64 distances on a
small table, (2) parity of whole searches with the pass on / off, (3) step and kernel times per parameter combination.
`probe_i8.py [rows] [nq] ["a=1,b=2;..."]` (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu, gist_lowrank_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
combos = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in c.split(',')) for c in (sys.argv[3] if len(sys.argv) > 3 else "flat_i8=1;flat_i8=2").split(';')]
gen = gist_lowrank_gpu if len(sys.argv) > 4 and sys.argv[4] == 'lowrank' else gist_like_gpu
dim, k = 960, 10
dev = torch.device('cuda', 0)

# (1) bound check on a small table
ns, nqs = 20000, 64
xb = gen(torch, ns, dim, 11, dev); xq = gen(torch, nqs, dim, 12, dev)
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(xb.data_ptr(), ns)
keys, qsq, qoff, dx = ix.flat_shortlist_keys(xq.cpu().numpy(), 2)
D = torch.cdist(xq.double(), xb.double()).pow(2).cpu().numpy()
lb = keys.astype(np.float64) + qoff.astype(np.float64)[:, None]
viol = int((lb > D * (1 + 1e-6) + 1e-6).sum())
print(f"bound check {ns}x{dim}, {nqs} queries: l1 {dx['dx_abs']:.5f} l2 {dx['dx_rel']:.2f} |mu| {dx['xsq_max']:.3f}; violations {viol}; "
      f"gap D - LB: mean {float((D - lb).mean()):.4f} min {float((D - lb).min()):.5f} (mean D {float(D.mean()):.3f})", flush=True)
assert viol == 0
del ix, xb, xq

base = gen(torch, n, dim, 1806, dev); qs = gen(torch, nq, dim, 1807, dev)
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n); del base
oi = torch.zeros(nq, k, dtype=torch.int64, device=dev); od = torch.zeros(nq, k, device=dev); oc = torch.zeros(nq, dtype=torch.int64, device=dev)
ix.prof_enable(True)
ref = None
for rnd in range(2):
    for c in combos:
        for name, v in c.items(): ix.set_param(name, v)
        for _ in range(2): ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        q0, r0 = ix.get_stat('flat_i8_queries'), ix.get_stat('flat_i8_redo')
        ix.prof_reset(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        name = 'flat_i8'
        p = ix.prof_get(name)
        if not p['launches']: name = 'flat_half'; p = ix.prof_get(name)
        if not p['launches']: name = 'flat_mfma'; p = ix.prof_get(name)
        cur = (oi.cpu().numpy().copy(), od.cpu().numpy().copy())
        if ref is None: ref = cur
        same = bool((ref[0] == cur[0]).all() and (ref[1] == cur[1]).all())
        print(f"n={n} nq={nq} {c} rnd {rnd}: step {dt*1e3:.3f} ms, {name} {p['ms']/p['launches']:.3f} ms ({p['bytes']/p['launches']/(p['ms']/p['launches'])/1e6:.0f} GB/s), "
              f"other {dt*1e3 - p['ms']/p['launches']:.3f} ms, same={same} fb={ix.flat_fallback_count()} i8 q/redo +{ix.get_stat('flat_i8_queries')-q0}/+{ix.get_stat('flat_i8_redo')-r0} "
              f"B/row {ix.get_stat('hbm_bytes_per_row')}", flush=True)
