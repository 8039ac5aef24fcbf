"""Where a db.search()-shaped call's time goes (gist_1000 table, one query per call): the Python wrapper, the bare C call through
ctypes with preallocated arrays, and the kernel's HIP-event time (tooling)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import lab_1806_vec_db_amd as vdb
from lab_1806_vec_db_amd import _lib as L
g = os.path.join('tests', 'golden')
base = np.fromfile(os.path.join(g, 'gist_1000.bin'), dtype=np.float32).reshape(1000, 960)
test = np.fromfile(os.path.join(g, 'gist_test.bin'), dtype=np.float32).reshape(1000, 960)
for n in (1000, 16000):
    b = np.tile(base, (n // 1000, 1))
    ix = vdb.GpuIndex(960, 'l2sqr'); ix.batch_add(b)
    k = 10
    ix.flat_knn(test[0], k)
    t = time.perf_counter()
    for q in range(1000): ix.flat_knn(test[q], k)
    t_py = (time.perf_counter() - t) / 1000
    lib = L.load()
    oi = np.zeros((1, k), dtype=np.uint64); od = np.zeros((1, k), dtype=np.float32); oc = np.zeros(1, dtype=np.uint64)
    pi, pd, pc = oi.ctypes.data_as(L.u64p), od.ctypes.data_as(L.f32p), oc.ctypes.data_as(L.u64p)
    qp = [test[q].ctypes.data_as(L.f32p) for q in range(1000)]
    t = time.perf_counter()
    for q in range(1000): lib.vdb_flat_knn(ix._h, qp[q], 1, 960, k, pi, pd, pc)
    t_c = (time.perf_counter() - t) / 1000
    ix.prof_enable(True); ix.prof_reset()
    for q in range(200): lib.vdb_flat_knn(ix._h, qp[q], 1, 960, k, pi, pd, pc)
    p = ix.prof_get('flat_small')
    print(f"rows {n}: python wrapper {t_py*1e6:.1f} us per call, bare C call {t_c*1e6:.1f} us, kernel (HIP events) {p['ms']/max(p['launches'],1)*1e3:.1f} us", flush=True)
    ix.close()

# the same kernel with the query already in HBM (device-pointer entry): isolates what reading the query over PCIe costs
import torch
for n in (1000, 16000):
    b = np.tile(base, (n // 1000, 1))
    ix = vdb.GpuIndex(960, 'l2sqr'); ix.batch_add(b)
    dq = torch.from_numpy(test[:8].copy()).cuda()
    o_i = torch.zeros((8, 10), dtype=torch.int64, device='cuda'); o_d = torch.zeros((8, 10), dtype=torch.float32, device='cuda'); o_c = torch.zeros((8,), dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    ix.prof_enable(True)
    for nq in (1, 8):
        for _ in range(20): ix.flat_knn_device(dq.data_ptr(), nq, 10, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
        ix.prof_reset()
        t = time.perf_counter()
        for _ in range(300): ix.flat_knn_device(dq.data_ptr(), nq, 10, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
        dt = (time.perf_counter() - t) / 300
        p = ix.prof_get('flat_small')
        print(f"rows {n}, {nq} queries in HBM: call {dt*1e6:.1f} us, kernel (HIP events) {p['ms']/max(p['launches'],1)*1e3:.1f} us", flush=True)
    ix.close()
