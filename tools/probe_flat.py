import sys, time, numpy as np, torch, ctypes as C
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
n, dim, nq, k = 1_000_000, 960, 1024, 10
g = torch.Generator(device='cuda'); g.manual_seed(1806)
base = (torch.randn(n, dim, device='cuda', generator=g) * 0.045 + 0.07).abs_().clamp_(0, 0.8)
qs = (torch.randn(nq, dim, device='cuda', generator=g) * 0.045 + 0.07).abs_().clamp_(0, 0.8)
torch.cuda.synchronize()
ix = vdb.GpuIndex(dim, 'l2sqr')
t = time.time(); ix.add_device(base.data_ptr(), n); print('add_device', time.time() - t)
oi = torch.zeros(nq, k, dtype=torch.int64, device='cuda'); od = torch.zeros(nq, k, device='cuda'); oc = torch.zeros(nq, dtype=torch.int64, device='cuda')
ix.prof_enable(True)
for mode in (2, 1):
    ix.set_flat_mode(mode)
    nqq = nq if mode == 2 else 64
    for it in range(3):
        ix.prof_reset()
        torch.cuda.synchronize(); t = time.time()
        ix.flat_knn_device(qs.data_ptr(), nqq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t
        p = ix.prof_get('flat_mfma' if mode == 2 else 'flat_exact')
        gbs = p['bytes'] / (p['ms'] * 1e-3) / 1e9 if p['ms'] else 0
        print(f"mode {mode} it {it}: {dt*1e3:.2f} ms for {nqq} q -> {nqq/dt:.0f} QPS; kernel {p['ms']:.2f} ms / {p['launches']} launches = {p['ms']/max(p['launches'],1):.3f} ms each, {gbs:.0f} GB/s; fallbacks {ix.flat_fallback_count()}")
    if mode == 2:
        ref = (oi.clone(), od.clone())
# exact vs mfma agreement on first 64
print('agree idx', bool((ref[0][:64] == oi[:64]).all()), 'dist', bool((ref[1][:64] == od[:64]).all()))
