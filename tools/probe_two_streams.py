"""Do two 1000-query Flat calls in flight on their own streams (two host threads, one workspace each) finish sooner than the same
calls one after another?  1M x 960 gist-like rows; aggregate queries per second for 1, 2, 3 threads (tooling)."""
import sys, time, threading
sys.path.insert(0, '.')
import torch
import bench as B
import lab_1806_vec_db_amd as vdb

n, dim, nq, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 960, 1000, 10
dev = torch.device("cuda:0")
base = B.gist_like_gpu(torch, n, dim, 1806, dev)
ix = vdb.GpuIndex(dim, "l2sqr", device=0)
torch.cuda.synchronize()
ix.add_device(base.data_ptr(), n)
del base
qs = [B.gist_like_gpu(torch, nq, dim, 1900 + t, dev) for t in range(3)]
outs = [(torch.zeros((nq, k), dtype=torch.int64, device=dev), torch.zeros((nq, k), dtype=torch.float32, device=dev),
         torch.zeros(nq, dtype=torch.int64, device=dev)) for _ in range(3)]
torch.cuda.synchronize()

def run(t, steps):
    o = outs[t]
    for _ in range(steps):
        ix.flat_knn_device(qs[t].data_ptr(), nq, k, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())

for nt in (1, 2, 3, 1, 2):
    th = [threading.Thread(target=run, args=(t, 5)) for t in range(nt)]
    [x.start() for x in th]; [x.join() for x in th]
    steps = 40
    th = [threading.Thread(target=run, args=(t, steps)) for t in range(nt)]
    t0 = time.perf_counter()
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t0
    print(f"threads {nt}: {nt * steps * nq / dt / 1e3:.1f}k queries/s, {dt / steps * 1e3:.3f} ms per round of {nt} calls", flush=True)
