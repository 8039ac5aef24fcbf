"""Timing/equality probe: k_flat_gemm (128 queries per pass) against k_flat_mfma on the bench workload (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tws = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3, 2]
dbg = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0]
dim, nq, k = 960, 1024, 10
g = torch.Generator(device='cuda'); g.manual_seed(1806)
base = (torch.randn(n, dim, device='cuda', generator=g) * 0.045 + 0.07).abs_().clamp_(0, 0.8)
qs = (torch.randn(nq, dim, device='cuda', generator=g) * 0.045 + 0.07).abs_().clamp_(0, 0.8)
torch.cuda.synchronize()
ix = vdb.GpuIndex(dim, 'l2sqr')
ix.add_device(base.data_ptr(), n)
oi = torch.zeros(nq, k, dtype=torch.int64, device='cuda'); od = torch.zeros(nq, k, device='cuda'); oc = torch.zeros(nq, dtype=torch.int64, device='cuda')
ix.prof_enable(True)
ix.set_flat_mode(2)
def run(tag):
    for rnd in range(2):
        ix.prof_reset(); fb0 = ix.flat_fallback_count()
        torch.cuda.synchronize(); t = time.time()
        ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t
        p = ix.prof_get('flat_mfma')
        print(f"{tag} rnd {rnd}: total {dt*1e3:.2f} ms -> {nq/dt:.0f} QPS; kernel {p['ms']/p['launches']:.3f} ms, {p['bytes']/(p['ms']*1e-3)/1e9:.0f} GB/s; fb={ix.flat_fallback_count()-fb0}", flush=True)
ix.set_param('flat_gemm', 1)
run('mfma share2')
ref = (oi.clone(), od.clone())
ix.set_param('flat_gemm', 2)
for tw in tws:
    ix.set_param('flat_gemm_tw', tw)
    for d in dbg:
        ix.set_param('flat_gemm_debug', d)
        run(f'gemm tw{tw} dbg{d}')
        if d == 0:
            print('   same as mfma path:', bool((ref[0] == oi).all() and (ref[1] == od).all()), flush=True)
