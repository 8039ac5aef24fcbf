import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim, nq, k = 960, 1000, 10
dev = torch.device('cuda', 0)
base = gist_like_gpu(torch, n, dim, 1806, dev)
qs = gist_like_gpu(torch, nq, dim, 1807, dev).cpu().numpy()
ix = vdb.GpuIndex(dim, 'l2sqr'); ix.add_device(base.data_ptr(), n)
gt, _, _ = ix.flat_knn(qs, k)
t = time.time(); ix.pq_build(n_bits=4, m=320, train_n=10000, max_iter=20, tol=1e-6, seed=42); print(f'pq_build {time.time()-t:.2f}s')
ix.prof_enable(True)
for ef in (100, 200):
    for it in range(2):
        ix.prof_reset(); t = time.time(); idx, d, c = ix.knn_pq(qs, k, ef); dt = time.time() - t
    rec = np.mean([len(set(idx[q].tolist()) & set(gt[q].tolist())) / k for q in range(nq)])
    p = ix.prof_get('pq_adc')
    print(f'PQ-Flat n={n} ef={ef}: {dt*1e3:.1f} ms -> {nq/dt:.0f} QPS recall@10={rec:.4f}; adc kernel {p["ms"]/max(p["launches"],1):.3f} ms/launch x{p["launches"]} = {p["ms"]:.1f} ms')
