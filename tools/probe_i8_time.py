"""times the 8-bit filter pass alone (results are not checked: for ablation builds of k_gemm8.hip, VDBHIP_LIB=...; tooling)"""
import sys, time, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu
n, nq, dim, k = 1_000_000, 1000, 960, 10
dev = torch.device('cuda', 0)
base = gist_like_gpu(torch, n, dim, 1806, dev); qs = gist_like_gpu(torch, nq, dim, 1807, dev)
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n); del base
ix.set_param('flat_i8', 2)
oi = torch.zeros(nq, k, dtype=torch.int64, device=dev); od = torch.zeros(nq, k, device=dev); oc = torch.zeros(nq, dtype=torch.int64, device=dev)
ix.prof_enable(True)
for combo in sys.argv[1:] or ['flat_gemm8_kc=0']:
    for kv in combo.split(','):
        ix.set_param(kv.split('=')[0], int(kv.split('=')[1]))
    for _ in range(3): ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
    ix.prof_reset()
    for _ in range(15): ix.flat_knn_device(qs.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
    p = ix.prof_get('flat_i8')
    print(combo, f"flat_i8 {p['ms']/p['launches']:.4f} ms", flush=True)
