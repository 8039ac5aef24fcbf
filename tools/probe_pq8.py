"""8-bit PQ-Flat at the bench's shape (1M low-rank gist-like rows, m = 320, ef = 100, 1000 queries): step time of the three scan variants
and what the quantised pass handed on (candidates per query, lists that overflowed, lists that came out short) (tooling)."""
import sys, time
sys.path.insert(0, '.')
import torch
import bench as B
import lab_1806_vec_db_amd as vdb

n, dim, nq, k, ef = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 960, 1000, 10, 100
dev = torch.device("cuda:0")
base = B.gist_lowrank_gpu(torch, n, dim, 1806, dev)
qs = B.gist_lowrank_gpu(torch, nq, dim, 1807, dev)
torch.cuda.synchronize()
tr = vdb.GpuIndex(dim, "l2sqr"); tr.add_device(base.data_ptr(), 20000)
tr.pq_build(n_bits=8, m=dim // 3, train_n=0, max_iter=5, tol=1e-6, seed=42)
cent = tr.pq_export()["centroids"]; tr.close()
ix = vdb.GpuIndex(dim, "l2sqr"); ix.add_device(base.data_ptr(), n); ix.pq_attach(8, dim // 3, cent, None)
o = (torch.zeros((nq, k), dtype=torch.int64, device=dev), torch.zeros((nq, k), dtype=torch.float32, device=dev), torch.zeros(nq, dtype=torch.int64, device=dev))
ref = None
for v in (0, 2, 1, 0):
    ix.set_param("pq_adc8_sliced", v)
    for _ in range(2):
        ix.knn_pq_device(qs.data_ptr(), nq, k, ef, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    s0 = {s: ix.get_stat(s) for s in ("pq_q8_overflow", "pq_q8_short", "pq_q8_hits_sum")}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ix.knn_pq_device(qs.data_ptr(), nq, k, ef, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    s1 = {s: ix.get_stat(s) for s in s0}
    res = (o[0].clone(), o[1].clone())
    same = ref is None or (torch.equal(ref[0], res[0]) and torch.equal(ref[1], res[1]))
    ref = ref or res
    print(f"variant {v}: {dt * 1e3:.2f} ms per 1000 queries; per call: overflowed {(s1['pq_q8_overflow'] - s0['pq_q8_overflow']) / 5:.1f}, "
          f"short {(s1['pq_q8_short'] - s0['pq_q8_short']) / 5:.1f}, candidates per query {(s1['pq_q8_hits_sum'] - s0['pq_q8_hits_sum']) / 5 / nq:.0f} "
          f"(largest list so far {ix.get_stat('pq_q8_hits_max')}); same answers as variant 0: {same}", flush=True)
