"""IVF build / search timing on gist-like rows (tooling)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu, gist_lowrank_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
kc = int(sys.argv[2]) if len(sys.argv) > 2 else 256
gen = gist_lowrank_gpu if (len(sys.argv) > 3 and sys.argv[3] == 'lowrank') else gist_like_gpu
dev = torch.device('cuda', 0)
base = gen(torch, n, 960, 1806, dev); qs = gen(torch, 1000, 960, 1807, dev).cpu().numpy()
ix = vdb.GpuIndex(960, 'l2sqr'); ix.add_device(base.data_ptr(), n)
gt, _, _ = ix.flat_knn(qs, 10)
t = time.time(); ix.ivf_build(kc, train_n=10000, max_iter=10, seed=42); print(f"ivf_build n={n} k={kc}: {time.time()-t:.1f} s", flush=True)
for npb in (4, 16):
    ref = None
    for half, q8 in ((0, 0), (1, 0), (1, 2), (1, 1)):  # plain scan / fp16 tier / + 8-bit tier query-major / cluster-major
        ix.set_param('ivf_half', half); ix.set_param('ivf_q8', q8)
        for it in range(3):
            t = time.time(); idx, d, c = ix.ivf_knn(qs, 10, npb); dt = time.time() - t
        rec = np.mean([len(set(idx[q].tolist()) & set(gt[q].tolist())) / 10 for q in range(1000)])
        same = True if ref is None else bool((ref[0] == idx).all() and (ref[1] == d).all())
        ref = (idx.copy(), d.copy())
        print(f"IVF n_probes={npb} ivf_half={half} ivf_q8={q8}: {dt*1e3:.1f} ms -> {1000/dt:.0f} QPS recall@10={rec:.4f} same={same}", flush=True)
ix.set_param('ivf_half', 1); ix.set_param('ivf_q8', 1)
