import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
n, dim, nq, k = 1_000_000, 960, 1024, 10
variants = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0]
shares = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2]
g = torch.Generator(device='cuda'); g.manual_seed(1806)
base = (torch.randn(n, dim, device='cuda', generator=g) * 0.045 + 0.07).abs_().clamp_(0, 0.8)
qs = (torch.randn(nq, dim, device='cuda', generator=g) * 0.045 + 0.07).abs_().clamp_(0, 0.8)
torch.cuda.synchronize()
ix = vdb.GpuIndex(dim, 'l2sqr')
ix.add_device(base.data_ptr(), n)
oi = torch.zeros(nq, k, dtype=torch.int64, device='cuda'); od = torch.zeros(nq, k, device='cuda'); oc = torch.zeros(nq, dtype=torch.int64, device='cuda')
ix.prof_enable(True)
ix.set_flat_mode(2)
ref = None
for rnd in range(2):
 for sh in shares:
  ix.set_param('flat_share', sh)
  for v in variants:
    ix.set_param('mfma_variant', v)
    ix.prof_reset()
    torch.cuda.synchronize(); t = time.time()
    ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t
    p = ix.prof_get('flat_mfma')
    gbs = p['bytes'] / (p['ms'] * 1e-3) / 1e9
    ok = True
    if ref is None: ref = (oi.clone(), od.clone())
    else: ok = bool((ref[0] == oi).all() and (ref[1] == od).all())
    print(f"share {sh} variant {v} rnd {rnd}: total {dt*1e3:.2f} ms -> {nq/dt:.0f} QPS; kernel {p['ms']/p['launches']:.3f} ms, {gbs:.0f} GB/s; same={ok} fb={ix.flat_fallback_count()}")
