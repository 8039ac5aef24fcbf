"""Where the one-launch exact kernel (k_flat_small) stops paying against the MFMA shortlist pipeline: calls of 1 / 4 / 16 queries
with device pointers on tables of 16k .. 256k rows x 960 (tooling; sets "flat_small_max_rows")."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from bench import gist_like_gpu
dev = torch.device('cuda', 0)
k = 10
for n in (16384, 32768, 65536, 131072, 262144):
    base = gist_like_gpu(torch, n, 960, 1806, dev); dq = gist_like_gpu(torch, 16, 960, 1807, dev)
    ix = vdb.GpuIndex(960, 'l2sqr'); ix.add_device(base.data_ptr(), n)
    o_i = torch.zeros((16, k), dtype=torch.int64, device=dev); o_d = torch.zeros((16, k), dtype=torch.float32, device=dev); o_c = torch.zeros((16,), dtype=torch.int64, device=dev)
    for nq in (1, 4, 16):
        res = {}
        for mode in (1, 2):  # 1 = off (MFMA pipeline for these sizes), 2 = forced
            ix.set_param('flat_small', mode)
            for _ in range(5): ix.flat_knn_device(dq.data_ptr(), nq, k, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(100): ix.flat_knn_device(dq.data_ptr(), nq, k, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
            torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t) / 100, o_i[:nq].cpu().numpy().copy(), o_d[:nq].cpu().numpy().copy())
        same = bool((res[1][1] == res[2][1]).all() and (res[1][2] == res[2][2]).all())
        print(f"rows {n} nq {nq}: MFMA pipeline {res[1][0]*1e6:.1f} us, one-launch exact {res[2][0]*1e6:.1f} us, same={same}", flush=True)
    ix.close(); del base
